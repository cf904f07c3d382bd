#!/bin/bash
# round-4 first GPU pass of a session: tests, headline + bf16x4 bench lines, kernel trace with the exposure timeline
tag=${1:-r04_s3}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -m gpu -q -x ) > $out/${tag}_pytest_gpu.log 2>&1
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python bench.py --edge-product bf16x4 --no-cpu-baseline > $out/${tag}_bench_bf16x4.json 2>> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serialised-pass > /tmp/kt.log 2>&1
db=$(find /tmp/prof_kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $db > $out/${tag}_rocprof_kernel_stats.txt 2>&1
tail -1 /tmp/kt.log >> $out/${tag}_rocprof_kernel_stats.txt
python $GRAFT_REPO_ROOT/tools/timeline.py $db k_perturb 10 > $out/${tag}_timeline.txt 2>&1
tail -3 $out/${tag}_pytest_gpu.log; tail -c 600 $out/${tag}_bench.json; cat $out/${tag}_timeline.txt
