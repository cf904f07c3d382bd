#!/bin/bash
# round 5, GPU pass 39: k_edge_hidden_mm with the requests of both row tiles of a node issued before the first MFMA (three waves per SIMD)
# against the committed kernel (var_nostag.so); parity tests first
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
B=diffdock_amd/csrc/build
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x ) > $out/r05_p30_pytest.log 2>&1
tail -2 $out/r05_p30_pytest.log
DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r05_e23 "A=1" "A=0 -- --lib $B/var_nostag.so" "A=1" "A=0 -- --lib $B/var_nostag.so" \
  "DDMI_STREAMS=1 -- --no-serialised-pass" "DDMI_STREAMS=1 -- --no-serialised-pass --lib $B/var_nostag.so" \
  "A=1 -- --samples 5" "A=0 -- --samples 5 --lib $B/var_nostag.so" "A=1 -- --config configs1" "A=0 -- --config configs1 --lib $B/var_nostag.so"
