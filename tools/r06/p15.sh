#!/bin/bash
# Round 6, pass 15: read-out prep kernels (edge MLP straight into the attribute rows, one prep launch per chain) + the times of every
# step in one launch: whole GPU suite, then 5 / 10 / 40 poses against the library of the commit before (build/var_prev.so)
cd $GRAFT_REPO_ROOT
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp DDMI_HARNESS=1
( time python -m pytest tests -m gpu -x -q ) > $out/r06_p15_pytest_gpu.log 2>&1
tail -4 $out/r06_p15_pytest_gpu.log
P=diffdock_amd/csrc/build/var_prev.so
for n in 5 10 40; do
  bash tools/ab.sh r06_p15_b$n "A=0 -- --samples $n --lib $P" "A=1 -- --samples $n" "A=0 -- --samples $n --lib $P" "A=1 -- --samples $n" | cut -c1-170
done
