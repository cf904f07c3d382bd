#!/bin/bash
# round 4, session 3: epilogue hand-offs of k_conv_fused without draining the LDS queue (DDMI_WAVE_ORDER) against the round-3
# behaviour (-DFCV_WSYNC_DRAIN build); parity suite on the new default first
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $out/r04_p12_pytest.log 2>&1
tail -4 $out/r04_p12_pytest.log
B=diffdock_amd/csrc/build
DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r04_e12 "A=1" "A=0 -- --lib $B/var_drain.so" "A=1" "A=0 -- --lib $B/var_drain.so" "A=1" "A=0 -- --lib $B/var_drain.so" \
  "A=1 -- --samples 5" "A=0 -- --samples 5 --lib $B/var_drain.so" "A=1 -- --all-atoms" "A=0 -- --all-atoms --lib $B/var_drain.so"
