#!/bin/bash
# round 4, session 3: preparation streams (ddmi_config.exec.prep_streams) -- route tests, A/B at 40 and 5 poses, exposure timelines
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "selectable or all_atom_bench_size or sharded or full_size_properties" ) > $out/r04_p4_pytest.log 2>&1
tail -4 $out/r04_p4_pytest.log
timeout 900 tools/ab.sh r04_e5 "DDMI_PREP_STREAMS=0" "DDMI_PREP_STREAMS=1" "DDMI_PREP_STREAMS=2" "DDMI_PREP_STREAMS=0" "DDMI_PREP_STREAMS=1" "DDMI_PREP_STREAMS=2" \
  "DDMI_PREP_STREAMS=0 -- --samples 5" "DDMI_PREP_STREAMS=1 -- --samples 5" "DDMI_PREP_STREAMS=2 -- --samples 5" "DDMI_PREP_STREAMS=0 -- --samples 5" "DDMI_PREP_STREAMS=1 -- --samples 5"
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/prof_kt
  DDMI_PREP_STREAMS=$v timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serialised-pass > /tmp/kt.log 2>&1
  python $GRAFT_REPO_ROOT/tools/timeline.py $(find /tmp/prof_kt -name "*.db" | head -1) k_perturb 10 > $out/r04_p4_timeline_prep$v.txt 2>&1
  head -24 $out/r04_p4_timeline_prep$v.txt
done
