#!/bin/bash
# round 5, GPU pass 35: in-kernel phase clocks of k_edge_hidden_mm (var_prof2.so = -DDDMI_PROFILING=2), one stream
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
DDMI_STREAMS=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serialised-pass --lib diffdock_amd/csrc/build/var_prof2.so > $out/r05_p27_prof2.json 2> $out/r05_p27_prof2.err
grep -E "EHPROF|FCPROF" $out/r05_p27_prof2.err > $out/r05_p27_phase_clocks.txt
grep EHPROF $out/r05_p27_phase_clocks.txt | head -4
