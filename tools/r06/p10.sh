#!/bin/bash
# Round 6, pass 10: work-stealing walk of k_conv_fused (exec.steal) -- parity first, then A/B at 40 / 20 / 10 / 5 poses and the other workloads
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp DDMI_HARNESS=1
DDMI_STEAL=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ddl_synth or selectable or forward_matches" 2>&1 | tail -3
bash tools/ab.sh r06_p10_b40 "DDMI_STEAL=0" "DDMI_STEAL=1" "DDMI_STEAL=0" "DDMI_STEAL=1" | cut -c1-150
for n in 20 10 5; do
  bash tools/ab.sh r06_p10_b$n "DDMI_STEAL=0 -- --samples $n" "DDMI_STEAL=1 -- --samples $n" "DDMI_STEAL=0 -- --samples $n" "DDMI_STEAL=1 -- --samples $n" | cut -c1-150
done
bash tools/ab.sh r06_p10_mix "DDMI_STEAL=0 -- --config mix" "DDMI_STEAL=1 -- --config mix" | cut -c1-150
bash tools/ab.sh r06_p10_c4 "DDMI_STEAL=0 -- --config configs4" "DDMI_STEAL=1 -- --config configs4" | cut -c1-150
