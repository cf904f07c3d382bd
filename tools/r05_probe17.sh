#!/bin/bash
# round 5, GPU passes 24-25: where the wall clock of a forward goes at 5, 10 and 40 poses (tools/timeline.py on a kernel trace),
# with the idle time attributed to the pair of kernels around every gap
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export DDMI_HARNESS=1
for n in 5 10 40; do
  rocprofv3 --kernel-trace --stats -d /tmp/prof_b$n -o kt -- python $GRAFT_REPO_ROOT/bench.py --samples $n --steps 1 --warmup 1 --no-cpu-baseline --no-serialised-pass > /tmp/kt$n.log 2>&1
  python $GRAFT_REPO_ROOT/tools/timeline.py $(find /tmp/prof_b$n -name "*.db" | head -1) k_perturb 10 22 > $out/r05_p20_timeline_b$n.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/timeline.py $(find /tmp/prof_b$n -name "*.db" | head -1) k_perturb 10 > $out/r05_p20_timeline_b${n}_eventpass.txt 2>&1
done
head -8 $out/r05_p20_timeline_b5.txt; grep -A 14 "idle time by" $out/r05_p20_timeline_b5.txt
head -12 $out/r05_p20_timeline_b40.txt; grep -A 10 "idle time by" $out/r05_p20_timeline_b40.txt
