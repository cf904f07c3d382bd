#!/bin/bash
# Round 6, pass 12: tight virtual-node list capacities (exec.list_caps 0) against nodes + edges / 32 (1): whole GPU suite, 5-40 poses, mix, configs4
cd $GRAFT_REPO_ROOT
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp DDMI_HARNESS=1
( time python -m pytest tests -m gpu -x -q ) > $out/r06_p12_pytest_gpu.log 2>&1
tail -4 $out/r06_p12_pytest_gpu.log
for n in 5 10 20 40; do
  bash tools/ab.sh r06_p12_b$n "DDMI_LIST_CAPS=1 -- --samples $n" "DDMI_LIST_CAPS=0 -- --samples $n" "DDMI_LIST_CAPS=1 -- --samples $n" "DDMI_LIST_CAPS=0 -- --samples $n" | cut -c1-150
done
bash tools/ab.sh r06_p12_mix "DDMI_LIST_CAPS=1 -- --config mix" "DDMI_LIST_CAPS=0 -- --config mix" | cut -c1-150
bash tools/ab.sh r06_p12_c4 "DDMI_LIST_CAPS=1 -- --config configs4" "DDMI_LIST_CAPS=0 -- --config configs4" | cut -c1-150
bash tools/ab.sh r06_p12_aa "DDMI_LIST_CAPS=1 -- --all-atoms" "DDMI_LIST_CAPS=0 -- --all-atoms" | cut -c1-150
