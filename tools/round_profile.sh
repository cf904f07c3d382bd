#!/bin/bash
# Round-end evidence, run ON THE GPU BOX through gpurun:  tools/round_profile.sh <tag>
# Writes gpurun_out/<tag>_{bench.json,rocprof_kernel_stats.txt,pmc_summary.txt,pytest_gpu.log,smoke.log}; copy them to profiles/.
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $out/${tag}_pytest_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/${tag}_smoke.log 2>&1
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/prof_kt -name "*.db" | head -1) > $out/${tag}_rocprof_kernel_stats.txt 2>&1
tail -1 /tmp/kt.log >> $out/${tag}_rocprof_kernel_stats.txt
: > $out/${tag}_pmc_summary.txt
i=0
for p in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $p -d /tmp/pmc$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmc$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pmc$i -name "*.db" | head -1) >> $out/${tag}_pmc_summary.txt 2>&1
done
