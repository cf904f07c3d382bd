#!/bin/bash
# Round 6, pass 17: cross-graph pair search forked at the start of the forward (next to the time terms), ligand encoder + receptor rows on the side stream: whole suite (stream dependencies), A/B against the commit before
# step in one launch: whole GPU suite, then 5 / 10 / 40 poses against the library of the commit before (build/var_prev.so)
cd $GRAFT_REPO_ROOT
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp DDMI_HARNESS=1
( time python -m pytest tests -m gpu -x -q ) > $out/r06_p17_pytest_gpu.log 2>&1
tail -4 $out/r06_p17_pytest_gpu.log
P=diffdock_amd/csrc/build/var_prev.so
for n in 5 10 40; do
  bash tools/ab.sh r06_p17_b$n "A=0 -- --samples $n --lib $P" "A=1 -- --samples $n" "A=0 -- --samples $n --lib $P" "A=1 -- --samples $n" | cut -c1-170
done
