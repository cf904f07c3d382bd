#!/bin/bash
# round 5, last GPU pass: the committed tree once more -- driver-style GPU suite, smoke, default bench line
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -x -q -m gpu ) > $out/r05_final_pytest_gpu.log 2>&1
tail -3 $out/r05_final_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/r05_final_smoke.log 2>&1
tail -1 $out/r05_final_smoke.log
python bench.py > $out/r05_final_bench.json 2> $out/r05_final_bench.err
tail -c 400 $out/r05_final_bench.json | head -c 200; echo
python -c "
import json
d=json.loads(open('$out/r05_final_bench.json').read().strip().split(chr(10))[-1])
print(d['value'], d['roofline']['frac'], d['roofline']['serialised']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])
"
