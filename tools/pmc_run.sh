#!/bin/bash
# usage: tools/pmc_run.sh <outfile> <pass1 counters> -- <pass2 counters> ...   (run on the GPU box through gpurun)
# Separate --pmc passes with --kernel-trace only, summarised on the box (the rocpd databases are too big to merge back).
out=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
: > "$GRAFT_REPO_ROOT/gpurun_out/$out"
IFS='|' read -ra PASSES <<< "$*"
for p in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $p -d /tmp/pmc$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmc$i.log 2>&1
  db=$(find /tmp/pmc$i -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $db >> "$GRAFT_REPO_ROOT/gpurun_out/$out"
done
