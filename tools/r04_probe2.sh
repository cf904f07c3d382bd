#!/bin/bash
# round 4, split-bf16 edge product: the whole GPU suite under DDMI_EDGE_PRODUCT=bf16x4, then f32 vs bf16x4 bench lines (same box)
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
( time DDMI_EDGE_PRODUCT=bf16x4 python -m pytest tests -m gpu -q -s -x ) > $out/r04_bf16x4_pytest_gpu.log 2>&1
tail -3 $out/r04_bf16x4_pytest_gpu.log
DDMI_TIME_GROUPS=1 tools/ab.sh r04_e2 "A=0" "A=1 -- --edge-product bf16x4" "A=0" "A=1 -- --edge-product bf16x4"
python bench.py --edge-product bf16x4 --no-cpu-baseline > $out/r04_p2_bench_bf16x4.json 2> $out/r04_p2_bench.err
cut -c1-400 $out/r04_p2_bench_bf16x4.json
