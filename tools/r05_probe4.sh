#!/bin/bash
# round 5, GPU pass 4: k_conv_fused with the one-wait tile prologue and the two-row-tile coupling epilogue (batched read-modify-writes of
# the pre-reduction) against the round-4 kernel (build/var_old.so = HEAD's k_conv.hip), same box; GPU parity first.
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
B=diffdock_amd/csrc/build
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x ) > $out/r05_p4_pytest.log 2>&1
tail -3 $out/r05_p4_pytest.log
DDMI_TIME_GROUPS=1 timeout 1500 tools/ab.sh r05_e3 "A=1" "A=0 -- --lib $B/var_old.so" "A=1" "A=0 -- --lib $B/var_old.so" \
  "DDMI_STREAMS=1 A=1" "DDMI_STREAMS=1 A=0 -- --lib $B/var_old.so" "A=1 -- --config configs1" "A=0 -- --config configs1 --lib $B/var_old.so" \
  "A=1 -- --samples 5" "A=0 -- --samples 5 --lib $B/var_old.so" "A=1 -- --all-atoms" "A=0 -- --all-atoms --lib $B/var_old.so"
