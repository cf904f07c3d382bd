#!/bin/bash
# round 5, GPU pass 31: gather-node ids of a tile requested before the live count (one dependent round trip less in the tile prologue)
# against the same library without it (var_novn.so); parity tests first
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
B=diffdock_amd/csrc/build
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x ) > $out/r05_p24_pytest.log 2>&1
tail -2 $out/r05_p24_pytest.log
DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r05_e19 "A=1 -- --samples 5" "A=0 -- --samples 5 --lib $B/var_novn.so" "A=1 -- --samples 5" "A=0 -- --samples 5 --lib $B/var_novn.so" \
  "A=1 -- --config configs1" "A=0 -- --config configs1 --lib $B/var_novn.so" "A=1 -- --config configs1" "A=0 -- --config configs1 --lib $B/var_novn.so" \
  "A=1" "A=0 -- --lib $B/var_novn.so" "A=1" "A=0 -- --lib $B/var_novn.so" \
  "A=1 -- --samples 20" "A=0 -- --samples 20 --lib $B/var_novn.so" "A=1 -- --config mix --steps 2" "A=0 -- --config mix --steps 2 --lib $B/var_novn.so"
