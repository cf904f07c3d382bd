#!/bin/bash
# per-phase HIP-event times of one bench configuration: tools/phases.sh "<env>" [bench args]  (GPU box)
cd $GRAFT_REPO_ROOT
envpart=$1; shift
env $envpart python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',round(d['value'],1),'ms/step',round(d['ms_per_step'],1))
for k,v in sorted(d['phase_ms_per_forward'].items(), key=lambda kv:-kv[1]): print(f'  {k:28s} {v:8.3f}')
print(json.dumps(d['config']['edges']))
"
