#!/bin/bash
# round 4, session 3: exposure timelines of the small batches (5 poses = one GPU's share of configs[3] at 8 GPUs; 10 poses = configs[1])
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for n in 5 10; do
  rm -rf /tmp/prof_kt
  timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --samples $n --steps 2 --warmup 1 --no-cpu-baseline --no-serialised-pass > /tmp/kt.log 2>&1
  python $GRAFT_REPO_ROOT/tools/timeline.py $(find /tmp/prof_kt -name "*.db" | head -1) k_perturb 20 > $out/r04_p6_timeline_b$n.txt 2>&1
  tail -1 /tmp/kt.log | cut -c1-200
  head -30 $out/r04_p6_timeline_b$n.txt
done
