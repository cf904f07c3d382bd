#!/bin/bash
# round 4, session 3: route knobs at 5 poses (one GPU's share of configs[3] at 8 GPUs)
cd $GRAFT_REPO_ROOT
timeout 1200 tools/ab.sh r04_e16 "A=0 -- --samples 5" "DDMI_FUSED_YS=2 -- --samples 5" "DDMI_FUSED_YS=4 -- --samples 5" "DDMI_FUSED_YS=6 -- --samples 5" "DDMI_FUSED_YS=8 -- --samples 5" \
  "DDMI_STREAMS=1 -- --samples 5" "DDMI_FC1_BATCH=0 -- --samples 5" "DDMI_EH_GRID=256 -- --samples 5" "DDMI_EH_GRID=1024 -- --samples 5" "DDMI_FUSED_PRERED=0 -- --samples 5" \
  "DDMI_FUSED_SHARED=0 -- --samples 5" "DDMI_FUSED_DENSE=0 -- --samples 5" "DDMI_FUSED_PACK=0 -- --samples 5" "A=0 -- --samples 5"
