#!/bin/bash
# Round 6, pass 7: the round-model tile split of chip-filling groups (exec.tile_split_rule 0) against one item per tile (1), 12-40 poses
cd $GRAFT_REPO_ROOT
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp DDMI_HARNESS=1
python -m pytest tests/test_gpu_parity.py -x -q -k "selectable or grouped or sharded" 2>&1 | tail -2
for n in 40 30 24 20 16 12; do
  bash tools/ab.sh r06_p7_b$n "DDMI_YS_RULE=1 -- --samples $n" "DDMI_YS_RULE=0 -- --samples $n" "DDMI_YS_RULE=1 -- --samples $n" "DDMI_YS_RULE=0 -- --samples $n" | cut -c1-150
done
bash tools/ab.sh r06_p7_mix "DDMI_YS_RULE=1 -- --config mix" "DDMI_YS_RULE=0 -- --config mix" | cut -c1-150
bash tools/ab.sh r06_p7_c4 "DDMI_YS_RULE=1 -- --config configs4" "DDMI_YS_RULE=0 -- --config configs4" | cut -c1-150
bash tools/ab.sh r06_p7_aa "DDMI_YS_RULE=1 -- --all-atoms" "DDMI_YS_RULE=0 -- --all-atoms" | cut -c1-150
