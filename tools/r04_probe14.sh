#!/bin/bash
# round 4, session 3: pre-reduction partial sums with ds_add_f32 (no return) instead of load-add-store round trips; var_drain.so is
# the previous epilogue.  Parity + run-to-run determinism tests first.
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $out/r04_p14_pytest.log 2>&1
tail -4 $out/r04_p14_pytest.log
B=diffdock_amd/csrc/build
DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r04_e14 "A=1" "A=0 -- --lib $B/var_drain.so" "A=1" "A=0 -- --lib $B/var_drain.so" "A=1" "A=0 -- --lib $B/var_drain.so" \
  "DDMI_STREAMS=1 A=1" "DDMI_STREAMS=1 A=0 -- --lib $B/var_drain.so" "A=1 -- --config configs1" "A=0 -- --config configs1 --lib $B/var_drain.so"
