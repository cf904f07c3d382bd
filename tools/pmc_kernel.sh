#!/bin/bash
# PMC counters of the kernels whose name contains <substr> (GPU box): tools/pmc_kernel.sh <substr> "<counters pass 1>" ["<pass 2>" ...]
sub=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for p in "$@"; do
  i=$((i+1))
  DDMI_STREAMS=1 rocprofv3 --kernel-trace --pmc $p -d /tmp/pk$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pk$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pk$i -name "*.db" | head -1) | grep "$sub"
done
