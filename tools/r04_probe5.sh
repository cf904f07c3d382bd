#!/bin/bash
# round 4, session 3: dense- vs sparse-row loop on the workloads whose gather nodes leave a short last virtual node
# (configs4: 80 edges = 32 + 32 + 16; mix: 45-atom ligands = 32 + 13), and the shard-deviation print of the sampling test
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -s -k "sharded or all_atom_bench_size" ) > $out/r04_p5_pytest.log 2>&1
grep -n "max |dx|\|passed\|failed" $out/r04_p5_pytest.log
DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r04_e6 "DDMI_FUSED_DENSE=1 -- --config configs4" "DDMI_FUSED_DENSE=0 -- --config configs4" \
  "DDMI_FUSED_DENSE=1 -- --config mix" "DDMI_FUSED_DENSE=0 -- --config mix" "DDMI_FUSED_DENSE=1 -- --config configs4" "DDMI_FUSED_DENSE=0 -- --config configs4"
