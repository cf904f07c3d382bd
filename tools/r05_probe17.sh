#!/bin/bash
# round 5, GPU pass 24: where the wall clock of a forward goes at 5 and 10 poses (tools/timeline.py on a kernel trace)
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export DDMI_HARNESS=1
for n in 5 10; do
  rocprofv3 --kernel-trace --stats -d /tmp/prof_b$n -o kt -- python $GRAFT_REPO_ROOT/bench.py --samples $n --steps 1 --warmup 1 --no-cpu-baseline --no-serialised-pass > /tmp/kt$n.log 2>&1
  python $GRAFT_REPO_ROOT/tools/timeline.py $(find /tmp/prof_b$n -name "*.db" | head -1) k_perturb 10 > $out/r05_p20_timeline_b$n.txt 2>&1
  tail -1 /tmp/kt$n.log | cut -c1-200
done
head -40 $out/r05_p20_timeline_b5.txt
