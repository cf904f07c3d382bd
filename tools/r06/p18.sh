#!/bin/bash
# Round 6, pass 18: k_time_terms (time embedding + per-graph terms + rec_sigma layer 2 in one launch) against the three-launch form
cd $GRAFT_REPO_ROOT
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp DDMI_HARNESS=1
( time python -m pytest tests -m gpu -x -q ) > $out/r06_p18_pytest_gpu.log 2>&1
tail -4 $out/r06_p18_pytest_gpu.log
for n in 5 10 40; do
  bash tools/ab.sh r06_p18_b$n "DDMI_TIME_TERMS=1 -- --samples $n" "DDMI_TIME_TERMS=0 -- --samples $n" "DDMI_TIME_TERMS=1 -- --samples $n" "DDMI_TIME_TERMS=0 -- --samples $n" | cut -c1-170
done
