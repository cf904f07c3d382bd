#!/bin/bash
# round 5, GPU pass 27: shared-node loop with the reduce-scatter of a chain staged over three MFMA slots
# against the same library without it (var_nosst.so) and the round-4 library (var_old.so); parity tests first
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
B=diffdock_amd/csrc/build
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x ) > $out/r05_p21_pytest.log 2>&1
tail -2 $out/r05_p21_pytest.log
DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r05_e16 "A=1" "A=0 -- --lib $B/var_nosst.so" "A=2 -- --lib $B/var_old.so" \
  "A=1" "A=0 -- --lib $B/var_nosst.so" "A=2 -- --lib $B/var_old.so" \
  "DDMI_STREAMS=1 -- --no-serialised-pass" "DDMI_STREAMS=1 -- --no-serialised-pass --lib $B/var_nosst.so" \
  "A=1 -- --samples 5" "A=0 -- --samples 5 --lib $B/var_nosst.so" "A=1 -- --config configs1" "A=0 -- --config configs1 --lib $B/var_nosst.so" \
  "A=1 -- --config mix --steps 2" "A=0 -- --config mix --steps 2 --lib $B/var_nosst.so" "A=1 -- --config configs4 --steps 1 --warmup 1" "A=0 -- --config configs4 --steps 1 --warmup 1 --lib $B/var_nosst.so"
