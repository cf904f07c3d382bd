#!/bin/bash
# Round-end evidence, run ON THE GPU BOX through gpurun:  tools/round_profile.sh <tag>
# Writes gpurun_out/<tag>_{pytest_gpu.log,smoke.log,bench*.json,rocprof_kernel_stats.txt,pmc_summary.txt,traffic.json}; copy them to profiles/.
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
( time python -m pytest tests -m gpu -q -s ) > $out/${tag}_pytest_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/${tag}_smoke.log 2>&1
cd /tmp && export TMPDIR=/tmp
# counters first (the default bench line then picks up profiles/traffic_latest.json of THIS build)
# (one launch regime in the kernel stats: the extra one-stream pass behind roofline.serialised is switched off here)
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serialised-pass --no-small-batch > /tmp/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/prof_kt -name "*.db" | head -1) > $out/${tag}_rocprof_kernel_stats.txt 2>&1
tail -1 /tmp/kt.log >> $out/${tag}_rocprof_kernel_stats.txt
: > $out/${tag}_pmc_summary.txt
i=0
for p in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  DDMI_STREAMS=1 rocprofv3 --kernel-trace --pmc $p -d /tmp/pmc$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-serialised-pass --no-small-batch > /tmp/pmc$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pmc$i -name "*.db" | head -1) >> $out/${tag}_pmc_summary.txt 2>&1
done
python $GRAFT_REPO_ROOT/tools/traffic_json.py $(find /tmp/pmc3 -name "*.db" | head -1) $(find /tmp/pmc4 -name "*.db" | head -1) k_conv_ configs2 \
  "profiles/${tag}_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT+TCC_MISS passes, kernels serialised on one stream)" \
  $(find /tmp/pmc5 -name "*.db" | head -1) > $out/${tag}_traffic.json 2>> $out/${tag}_pmc_summary.txt
cp $out/${tag}_traffic.json $GRAFT_REPO_ROOT/profiles/traffic_latest.json
cd $GRAFT_REPO_ROOT
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python bench.py --config configs1 > $out/${tag}_bench_configs1.json 2>> $out/${tag}_bench.err
python bench.py --config mix --steps 2 > $out/${tag}_bench_mix.json 2>> $out/${tag}_bench.err
python bench.py --config configs4 --steps 2 --warmup 1 > $out/${tag}_bench_configs4.json 2>> $out/${tag}_bench.err
python bench.py --samples 5 --no-cpu-baseline > $out/${tag}_bench_b5.json 2>> $out/${tag}_bench.err
python bench.py --all-atoms > $out/${tag}_bench_all_atoms.json 2>> $out/${tag}_bench.err
python bench.py --tile-per-pose --no-cpu-baseline > $out/${tag}_bench_tile_per_pose.json 2>> $out/${tag}_bench.err
# secondary line: split-bf16 edge product (its own dtype), with the whole GPU suite under that route
python bench.py --edge-product bf16x4 --no-cpu-baseline > $out/${tag}_bench_bf16x4.json 2>> $out/${tag}_bench.err
( time DDMI_EDGE_PRODUCT=bf16x4 python -m pytest tests -m gpu -q ) > $out/${tag}_bf16x4_pytest_gpu.log 2>&1
# where the wall clock of a forward goes (tools/timeline.py)
python $GRAFT_REPO_ROOT/tools/timeline.py $(find /tmp/prof_kt -name "*.db" | head -1) k_perturb 10 22 > $out/${tag}_timeline.txt 2>&1   # (22: the trailing HIP-event pass of bench.py left out)
# timelines of the small batches (one GPU's share of configs[3] at 8 / 4 GPUs)
cd /tmp
for n in 5 10; do
  rm -rf /tmp/prof_b$n
  rocprofv3 --kernel-trace --stats -d /tmp/prof_b$n -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --samples $n --no-cpu-baseline --no-serialised-pass > /tmp/kt_$n.log 2>&1
  python $GRAFT_REPO_ROOT/tools/timeline.py $(find /tmp/prof_b$n -name "*.db" | head -1) k_perturb 10 22 > $out/${tag}_timeline_b$n.txt 2>&1
done
cd $GRAFT_REPO_ROOT
# the optional routes of round 6 on the headline workload (own lines, not the default)
DDMI_GROUPED=2 python bench.py --no-cpu-baseline --no-small-batch > $out/${tag}_bench_grouped.json 2>> $out/${tag}_bench.err
DDMI_NODE_UPDATE=1 python bench.py --no-cpu-baseline --no-small-batch > $out/${tag}_bench_node_update.json 2>> $out/${tag}_bench.err
for n in 10 16 20 30; do
  python bench.py --samples $n --no-cpu-baseline > $out/${tag}_bench_b$n.json 2>> $out/${tag}_bench.err
done
# the multi-rank bench path as the driver launches it (no extra flags), 2 and 8 ranks time-slicing this ONE GPU over gloo: readiness only
for n in 2 8; do
  DDMI_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus $n --steps 3 --warmup 1 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|OMP_NUM_THREADS\|^\*\*\*" | cut -c1-20000 > $out/${tag}_bench_share$n.log
done
