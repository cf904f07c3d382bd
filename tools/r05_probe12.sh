#!/bin/bash
# round 5, GPU pass 17: overlapped layer boundaries (ddmi_exec_options.layer_overlap) and the two-request main-loop prologue:
# parity tests, then A/B  [new kernel, overlapped] | [new kernel, joined layers] | [round-4 kernel (var_old.so), joined layers]
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
B=diffdock_amd/csrc/build
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x ) > $out/r05_p15_pytest.log 2>&1
tail -2 $out/r05_p15_pytest.log
DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r05_e11 "A=1" "A=0 -- --layer-overlap joined" "A=2 -- --layer-overlap joined --lib $B/var_old.so" \
  "A=1" "A=0 -- --layer-overlap joined" "A=2 -- --layer-overlap joined --lib $B/var_old.so" \
  "A=1 -- --samples 10" "A=0 -- --samples 10 --layer-overlap joined" "A=1 -- --samples 5 --layer-overlap always" "A=0 -- --samples 5" \
  "A=1 -- --config configs1" "A=0 -- --config configs1 --layer-overlap joined"
