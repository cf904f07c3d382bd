#!/bin/bash
# round 5, first GPU pass: baseline line of this box, the multi-rank bench path executed on ONE GPU (DDMI_BENCH_SHARE_GPU: all
# ranks on cuda:0 over gloo; readiness only, no scaling claim), the producer / consumer micro-kernel (tools/micro/pc_ring.hip),
# in-kernel phase clocks of the shipped k_conv_fused (var_prof2 = -DDDMI_PROFILING=2), the bf16x4 GPU tests.
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/r05_p1_base.json 2> $out/r05_p1_base.err
for n in 2 8; do
  DDMI_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) \
    bench.py --gpus $n --steps 2 --warmup 1 --verify-shards > $out/r05_bench_share$n.log 2>&1
done
( cd tools/micro && timeout 300 ./pc_ring 6 ) > $out/r05_micro_pc_ring.txt 2>&1
DDMI_STREAMS=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serialised-pass --lib diffdock_amd/csrc/build/var_prof2.so > $out/r05_p1_prof2.json 2> $out/r05_p1_prof2.err
grep FCPROF $out/r05_p1_prof2.err > $out/r05_p1_phase_clocks.txt
( time timeout 1200 python -m pytest tests/test_gpu_bf16x4.py -q -s ) > $out/r05_p1_pytest_bf16x4.log 2>&1
tail -3 $out/r05_p1_pytest_bf16x4.log
tail -c 600 $out/r05_p1_base.json | head -c 300; echo
grep -o '"value": [0-9.]*' $out/r05_p1_base.json | head -1
tail -2 $out/r05_bench_share2.log | cut -c 1-400
tail -2 $out/r05_bench_share8.log | cut -c 1-400
cat $out/r05_micro_pc_ring.txt
