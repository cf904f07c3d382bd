#!/bin/bash
# sample the shader clock / power while a bench runs (GPU box): tools/clocks.sh [bench args]
cd $GRAFT_REPO_ROOT
( for i in $(seq 1 140); do echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Graphics Package" | sed 's/.*: //' | tr '\n' ' ')"; done ) > /tmp/clk.log 2>&1 &
python bench.py --steps 12 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | cut -c1-200
wait
awk '{print $2,$3,$4,$5,$6}' /tmp/clk.log | uniq -c | head -60
