#!/bin/bash
# Round 6, pass 14: k_node_update with four waves per node for small node counts (k_reduce_bn's row deal) against k_reduce_bn + GEMM launches
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp DDMI_HARNESS=1
python -m pytest tests/test_gpu_parity.py -x -q -k "selectable" 2>&1 | tail -2
for n in 5 10 20 40; do
  bash tools/ab.sh r06_p14_b$n "DDMI_NODE_UPDATE=0 -- --samples $n" "DDMI_NODE_UPDATE=1 -- --samples $n" "DDMI_NODE_UPDATE=3 -- --samples $n" "DDMI_NODE_UPDATE=0 -- --samples $n" "DDMI_NODE_UPDATE=1 -- --samples $n" | cut -c1-150
done
