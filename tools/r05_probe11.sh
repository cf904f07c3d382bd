#!/bin/bash
# round 5, GPU pass 15: in-kernel phase clocks of k_conv_fused after the epilogue diet (var_prof2c = -DDDMI_PROFILING=2 on the shipped source)
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
DDMI_STREAMS=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serialised-pass --lib diffdock_amd/csrc/build/var_prof2c.so > $out/r05_p16_prof2.json 2> $out/r05_p16_prof2.err
grep FCPROF $out/r05_p16_prof2.err > $out/r05_p16_phase_clocks.txt
cat $out/r05_p16_phase_clocks.txt | head -4
