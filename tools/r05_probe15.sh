#!/bin/bash
# round 5, GPU pass 21: route knobs re-swept on the round-5 kernel (the coupling phase is 40 % shorter than when they were set):
# workgroups per tile (tile_split), workgroups of k_edge_hidden_mm (hidden_grid), one stream
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
DDMI_TIME_GROUPS=1 timeout 1500 tools/ab.sh r05_e14 "A=1" "DDMI_FUSED_YS=2" "DDMI_FUSED_YS=3" "DDMI_FUSED_YS=4" "DDMI_FUSED_YS=5" "DDMI_FUSED_YS=6" "DDMI_FUSED_YS=8" \
  "DDMI_FUSED_YS_SMALL=2" "DDMI_FUSED_YS_SMALL=6" "DDMI_EH_GRID=768" "DDMI_EH_GRID=1024" "DDMI_EH_GRID=4096" "A=1"
