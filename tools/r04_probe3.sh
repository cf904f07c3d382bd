#!/bin/bash
# round 4: in-tile pre-reduction of the lig<-rec messages: GPU suite (default = pre-reduction on, f32), then A/B on / off for both edge-product routes
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -m gpu -q -s -x ) > $out/r04_p3_pytest_gpu.log 2>&1
tail -3 $out/r04_p3_pytest_gpu.log
DDMI_TIME_GROUPS=1 tools/ab.sh r04_e4 "DDMI_FUSED_PRERED=0" "DDMI_FUSED_PRERED=1" "DDMI_FUSED_PRERED=0" "DDMI_FUSED_PRERED=1" "DDMI_FUSED_PRERED=0 -- --edge-product bf16x4" "DDMI_FUSED_PRERED=1 -- --edge-product bf16x4"
