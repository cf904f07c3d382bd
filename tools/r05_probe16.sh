#!/bin/bash
# round 5, GPU pass 23: granule prefetch (next granule's chunk-0 weights + bias row requested before the coupling phase, in registers)
# against the same library without it (var_nogp.so) and the round-4 library (var_old.so); parity tests first
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
B=diffdock_amd/csrc/build
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x ) > $out/r05_p19_pytest.log 2>&1
tail -2 $out/r05_p19_pytest.log
DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r05_e15 "A=1" "A=0 -- --lib $B/var_nogp.so" "A=2 -- --lib $B/var_old.so" \
  "A=1" "A=0 -- --lib $B/var_nogp.so" "A=2 -- --lib $B/var_old.so" \
  "DDMI_STREAMS=1 -- --no-serialised-pass" "DDMI_STREAMS=1 -- --no-serialised-pass --lib $B/var_nogp.so" \
  "A=1 -- --samples 5" "A=0 -- --samples 5 --lib $B/var_nogp.so" "A=1 -- --config configs1" "A=0 -- --config configs1 --lib $B/var_nogp.so" \
  "A=1 -- --all-atoms" "A=0 -- --all-atoms --lib $B/var_nogp.so"
