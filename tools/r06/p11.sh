#!/bin/bash
# Round 6, pass 11: the two degree scans of a graph build in one launch (k_exclusive_scan2): whole GPU suite, 5- and 40-pose lines
cd $GRAFT_REPO_ROOT
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp DDMI_HARNESS=1
( time python -m pytest tests -m gpu -x -q ) > $out/r06_p11_pytest_gpu.log 2>&1
tail -4 $out/r06_p11_pytest_gpu.log
bash tools/ab.sh r06_p11 "DDMI_X=1" "DDMI_X=1 -- --samples 5" "DDMI_X=1 -- --samples 10" "DDMI_X=1" "DDMI_X=1 -- --samples 5" | cut -c1-150
