#!/bin/bash
# round 5, GPU pass 19: unrolled message-row stores (fc_store_rows_u) and the next-granule weight warm-up:
# parity tests, then A/B  [stores + warm-up] | [stores only (var_nowarm.so = -DFCV_NOWARM)] | [round-4 kernel (var_old.so)]
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
B=diffdock_amd/csrc/build
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x ) > $out/r05_p17_pytest.log 2>&1
tail -2 $out/r05_p17_pytest.log
DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r05_e12 "A=1" "A=0 -- --lib $B/var_nowarm.so" "A=2 -- --lib $B/var_old.so" \
  "A=1" "A=0 -- --lib $B/var_nowarm.so" "A=2 -- --lib $B/var_old.so" \
  "A=1 -- --samples 5" "A=0 -- --samples 5 --lib $B/var_nowarm.so" "A=1 -- --config configs1" "A=0 -- --config configs1 --lib $B/var_nowarm.so"
