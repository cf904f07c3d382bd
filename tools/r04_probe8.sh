#!/bin/bash
# round 4, session 3: all-atom workload, per-group split and the route knobs (the thresholds were tuned on the CG workload)
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
DDMI_TIME_GROUPS=1 timeout 1500 tools/ab.sh r04_e8 "A=0 -- --all-atoms" "DDMI_FUSED_DENSE=0 -- --all-atoms" "DDMI_FUSED_DENSE=2 -- --all-atoms" \
  "DDMI_FUSED_PRERED=0 -- --all-atoms" "DDMI_STREAMS=1 -- --all-atoms" "DDMI_FUSED_YS=2 -- --all-atoms" "DDMI_FUSED_PACK=0 -- --all-atoms" "DDMI_EH_GRID=8192 -- --all-atoms"
