#!/bin/bash
# round-3 first GPU probe: baseline line, in-kernel phase clocks of k_conv_fused per edge group, L2 hit counters
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/p1_base.json 2> $out/p1_base.err
for v in PROF PROF2; do
  DDMI_STREAMS=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --lib diffdock_amd/csrc/build/var_$v.so > $out/p1_$v.json 2> $out/p1_$v.err
  grep FCPROF $out/p1_$v.err > $out/p1_$v.txt
done
DDMI_STREAMS=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --samples 5 --lib diffdock_amd/csrc/build/var_PROF.so > $out/p1_PROF_b5.json 2> $out/p1_PROF_b5.err
grep FCPROF $out/p1_PROF_b5.err > $out/p1_PROF_b5.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "TCC_HIT[A-Za-z_]*\|TCC_MISS[A-Za-z_]*\|TCP_TCC_READ_REQ[A-Za-z_]*\|TCC_REQ[A-Za-z_]*\|TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCP_TOTAL_CACHE_ACCESSES[A-Za-z_]*\|TCC_READ[A-Za-z_]*" | sort | uniq > $out/p1_counters.txt
cd $GRAFT_REPO_ROOT
bash tools/pmc_kernel.sh k_conv_fused "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" > $out/p1_tcc.txt 2>&1
tail -3 /tmp/pk1.log >> $out/p1_tcc.txt
