#!/bin/bash
# round 4, first GPU pass: chunk-step micro-benchmark variants + the state of HEAD (tests, bench line)
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
./tools/micro/mfma_phase > $out/r04_micro_mfma_phase.txt 2>&1
./tools/micro/mfma_waves >> $out/r04_micro_mfma_phase.txt 2>&1
( time python -m pytest tests -m gpu -q -x ) > $out/r04_p1_pytest_gpu.log 2>&1
python bench.py --no-cpu-baseline > $out/r04_p1_bench.json 2> $out/r04_p1_bench.err
cat $out/r04_micro_mfma_phase.txt; tail -3 $out/r04_p1_pytest_gpu.log; cat $out/r04_p1_bench.json | cut -c1-600
