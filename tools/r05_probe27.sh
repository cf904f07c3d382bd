#!/bin/bash
# round 5, GPU pass 40: the in-tile pre-reduction with LDS float adds (ds_add_f32, no return) instead of read-add-write
# against the committed kernel (var_noadd.so); parity tests first (the sums are the same operations in the same order)
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
B=diffdock_amd/csrc/build
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x ) > $out/r05_p31_pytest.log 2>&1
tail -2 $out/r05_p31_pytest.log
DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r05_e24 "A=1" "A=0 -- --lib $B/var_noadd.so" "A=1" "A=0 -- --lib $B/var_noadd.so" \
  "DDMI_STREAMS=1 -- --no-serialised-pass" "DDMI_STREAMS=1 -- --no-serialised-pass --lib $B/var_noadd.so" \
  "A=1 -- --samples 5" "A=0 -- --samples 5 --lib $B/var_noadd.so"
