#!/bin/bash
# Round 6, pass 13: the final tree once more -- whole GPU suite (f32), smoke, default bench line
cd $GRAFT_REPO_ROOT
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp DDMI_HARNESS=1
( time python -m pytest tests -m gpu -x -q ) > $out/r06_final_pytest_gpu.log 2>&1
tail -4 $out/r06_final_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/r06_final_smoke.log 2>&1; tail -1 $out/r06_final_smoke.log
python bench.py > $out/r06_final_bench.json 2> $out/r06_final_bench.err
python -c "
import json
d=json.loads(open('$out/r06_final_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('poses/s', round(d['value'],1), 'wall', round(r['frac'],3), 'ser', round(r['serialised']['frac'],3), 'traffic', r['traffic'], 'small', {k:round(v['value'],1) for k,v in d['small_batch'].items() if isinstance(v,dict)}, 'cpu', d['cpu_baseline']['value'])
"
