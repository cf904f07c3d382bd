#!/bin/bash
# round 5, GPU pass 2: instruction-cache counters of k_conv_fused (2-MB kernel image: cold code per granule?), bit-exact shard
# test (tile_per_pose), pose shards on one GPU (--pose-shards), the multi-rank bench path re-run with the per-block shard check.
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tile_per_pose or sharded_sampling or full_size_properties" ) > $out/r05_p2_pytest.log 2>&1
tail -3 $out/r05_p2_pytest.log
timeout 1500 tools/ab.sh r05_e1 "A=0" "A=1 -- --pose-shards 2" "A=2 -- --pose-shards 4" "A=3 -- --tile-per-pose" "A=4 -- --tile-per-pose --pose-shards 2" \
  "A=0" "A=1 -- --pose-shards 2" "A=5 -- --pose-shards 2 --config configs1" "A=6 -- --config configs1" "A=7 -- --pose-shards 8"
for n in 2 8; do
  DDMI_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29520 + n)) \
    bench.py --gpus $n --steps 1 --warmup 1 --verify-shards --fixed-center-conv --tile-per-pose > $out/r05_bench_share$n.log 2>&1
  grep -o '"shard_check": {[^}]*}' $out/r05_bench_share$n.log
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_[A-Z_]*ICACHE[A-Za-z_]*\|SQ_IFETCH[A-Za-z_]*\|SQC_[A-Z_]*INST[A-Za-z_]*\|SQ_INST_LEVEL[A-Za-z_]*\|SQ_WAIT_INST[A-Za-z_]*\|SQ_INSTS_[A-Z_]*" | sort | uniq > $out/r05_p2_counters.txt
cat $out/r05_p2_counters.txt | tr '\n' ' '
cd $GRAFT_REPO_ROOT
bash tools/pmc_kernel.sh k_conv_fused "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" > $out/r05_p2_icache.txt 2>&1
tail -3 /tmp/pk1.log >> $out/r05_p2_icache.txt
head -50 $out/r05_p2_icache.txt
