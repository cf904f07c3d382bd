#!/bin/bash
# round 5, GPU pass 7: branch-free masked puts in the coupling epilogue (+ two row tiles per phase, one-wait tile prologue) against the
# round-4 kernel (build/var_old.so), same box; GPU parity first.
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
B=diffdock_amd/csrc/build
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_bf16x4.py -q -x ) > $out/r05_p7_pytest.log 2>&1
tail -3 $out/r05_p7_pytest.log
DDMI_TIME_GROUPS=1 timeout 1500 tools/ab.sh r05_e4 "A=1" "A=0 -- --lib $B/var_old.so" "A=1" "A=0 -- --lib $B/var_old.so" \
  "DDMI_STREAMS=1 A=1" "DDMI_STREAMS=1 A=0 -- --lib $B/var_old.so" "A=1 -- --config configs1" "A=0 -- --config configs1 --lib $B/var_old.so" \
  "A=1 -- --samples 5" "A=0 -- --samples 5 --lib $B/var_old.so" "A=1 -- --all-atoms" "A=0 -- --all-atoms --lib $B/var_old.so" \
  "A=1 -- --edge-product bf16x4" "A=0 -- --edge-product bf16x4 --lib $B/var_old.so"
