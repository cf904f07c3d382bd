#!/bin/bash
# round 4, session 3: message values written straight from the coupling registers (direct stores) against the staged rows
# (-DFCV_STAGED build = the round-3 epilogue); parity suite on the new default first
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $out/r04_p15_pytest.log 2>&1
tail -4 $out/r04_p15_pytest.log
B=diffdock_amd/csrc/build
DDMI_TIME_GROUPS=1 timeout 1500 tools/ab.sh r04_e15 "A=1" "A=0 -- --lib $B/var_staged.so" "A=1" "A=0 -- --lib $B/var_staged.so" "A=1" "A=0 -- --lib $B/var_staged.so" \
  "DDMI_STREAMS=1 A=1" "DDMI_STREAMS=1 A=0 -- --lib $B/var_staged.so" "A=1 -- --config configs1" "A=0 -- --config configs1 --lib $B/var_staged.so" \
  "A=1 -- --samples 5" "A=0 -- --samples 5 --lib $B/var_staged.so" "A=1 -- --all-atoms" "A=0 -- --all-atoms --lib $B/var_staged.so" \
  "A=1 -- --config configs4" "A=0 -- --config configs4 --lib $B/var_staged.so"
