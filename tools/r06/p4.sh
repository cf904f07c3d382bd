#!/bin/bash
# Round 6, pass 4: k_node_update with one wave per node (16 waves per 16-node tile) against k_reduce_bn + GEMM launches, at 40 / 10 / 5
# poses; the GPU tests that changed.
cd $GRAFT_REPO_ROOT
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp DDMI_HARNESS=1
( time python -m pytest tests/test_gpu_parity.py -x -q -k "layer_overlap or selectable or grouped" ) > $out/r06_p4_pytest_gpu.log 2>&1
tail -4 $out/r06_p4_pytest_gpu.log
bash tools/ab.sh r06_p4_b40 "DDMI_NODE_UPDATE=0" "DDMI_X=1" "DDMI_NODE_UPDATE=0" "DDMI_X=1"
bash tools/ab.sh r06_p4_b10 "DDMI_NODE_UPDATE=0 -- --samples 10" "DDMI_X=1 -- --samples 10" "DDMI_NODE_UPDATE=0 -- --samples 10" "DDMI_X=1 -- --samples 10"
bash tools/ab.sh r06_p4_b5 "DDMI_NODE_UPDATE=0 -- --samples 5" "DDMI_X=1 -- --samples 5" "DDMI_NODE_UPDATE=0 -- --samples 5" "DDMI_X=1 -- --samples 5" "DDMI_GROUPED=2 -- --samples 5"
