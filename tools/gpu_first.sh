#!/bin/bash
# first GPU pass of a round: tests, smoke, bench lines of every workload, one rocprof kernel trace
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py::test_twenty_step_teacher_forced_scores_match_reference_execution --deselect tests/test_gpu_fullsize.py::test_twenty_step_free_running_rmsd_to_reference_execution --deselect tests/test_gpu_fullsize.py::test_large_pocket_forward_matches_reference_execution ) > $out/${tag}_pytest_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/${tag}_smoke.log 2>&1
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python bench.py --config configs1 --no-cpu-baseline > $out/${tag}_bench_configs1.json 2>> $out/${tag}_bench.err
python bench.py --samples 5 --no-cpu-baseline > $out/${tag}_bench_b5.json 2>> $out/${tag}_bench.err
python bench.py --config mix --steps 2 --no-cpu-baseline > $out/${tag}_bench_mix.json 2>> $out/${tag}_bench.err
python bench.py --config configs4 --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_configs4.json 2>> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/prof_kt -name "*.db" | head -1) > $out/${tag}_rocprof_kernel_stats.txt 2>&1
tail -1 /tmp/kt.log >> $out/${tag}_rocprof_kernel_stats.txt
