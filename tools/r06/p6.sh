#!/bin/bash
# Round 6, pass 6: finer work items for the last chip-filling launch of each stream in a layer (exec.tile_split_last), 40 and 10 poses
cd $GRAFT_REPO_ROOT
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp DDMI_HARNESS=1
bash tools/ab.sh r06_p6_b40 "DDMI_X=1" "DDMI_FUSED_YS_LAST=2" "DDMI_FUSED_YS_LAST=3" "DDMI_FUSED_YS_LAST=4" "DDMI_X=1" "DDMI_FUSED_YS_LAST=2" "DDMI_FUSED_YS_LAST=8"
bash tools/ab.sh r06_p6_b20 "DDMI_X=1 -- --samples 20" "DDMI_FUSED_YS_LAST=2 -- --samples 20" "DDMI_FUSED_YS_LAST=4 -- --samples 20"
