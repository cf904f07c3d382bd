#!/bin/bash
# Round 6, pass 9: issue order of a layer's groups on the two streams (exec.group_order), 40 / 20 / 10 / 5 poses
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp DDMI_HARNESS=1
python -m pytest tests/test_gpu_parity.py -x -q -k "ddl_synth" 2>&1 | tail -2
bash tools/ab.sh r06_p9_b40 "DDMI_GROUP_ORDER=0" "DDMI_GROUP_ORDER=1" "DDMI_GROUP_ORDER=2" "DDMI_GROUP_ORDER=3" "DDMI_GROUP_ORDER=0" "DDMI_GROUP_ORDER=1" | cut -c1-150
for n in 20 10 5; do
  bash tools/ab.sh r06_p9_b$n "DDMI_GROUP_ORDER=0 -- --samples $n" "DDMI_GROUP_ORDER=1 -- --samples $n" "DDMI_GROUP_ORDER=2 -- --samples $n" "DDMI_GROUP_ORDER=3 -- --samples $n" | cut -c1-150
done
