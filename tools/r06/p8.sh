#!/bin/bash
# Round 6, pass 8: the round-model tile split also for the groups of small layers (exec.tile_split_rule 2 against 0), 5-14 poses
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp DDMI_HARNESS=1
for n in 5 6 8 10 12 14; do
  bash tools/ab.sh r06_p8_b$n "DDMI_YS_RULE=0 -- --samples $n" "DDMI_YS_RULE=2 -- --samples $n" "DDMI_YS_RULE=0 -- --samples $n" "DDMI_YS_RULE=2 -- --samples $n" | cut -c1-150
done
