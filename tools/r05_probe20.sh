#!/bin/bash
# round 5, GPU pass 29: the split-bf16 line with the NODE CONTRACTION on split operands too (classic / dense loop, BC) against the
# same line with the f32 contraction (var_nobc.so); the bf16x4 GPU tests and the whole parity file under the route first
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
B=diffdock_amd/csrc/build
( timeout 900 python -m pytest tests/test_gpu_bf16x4.py -q -x -s ) > $out/r05_p23_pytest_bf16x4.log 2>&1
tail -4 $out/r05_p23_pytest_bf16x4.log
( DDMI_EDGE_PRODUCT=bf16x4 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x ) > $out/r05_p23_pytest_route.log 2>&1
tail -2 $out/r05_p23_pytest_route.log
DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r05_e18 "A=1 -- --edge-product bf16x4" "A=0 -- --edge-product bf16x4 --lib $B/var_nobc.so" \
  "A=1 -- --edge-product bf16x4" "A=0 -- --edge-product bf16x4 --lib $B/var_nobc.so" "A=1" "A=0 -- --lib $B/var_nobc.so" \
  "DDMI_STREAMS=1 -- --edge-product bf16x4 --no-serialised-pass" "DDMI_STREAMS=1 -- --edge-product bf16x4 --no-serialised-pass --lib $B/var_nobc.so" \
  "A=1 -- --edge-product bf16x4 --samples 5" "A=0 -- --edge-product bf16x4 --samples 5 --lib $B/var_nobc.so"
