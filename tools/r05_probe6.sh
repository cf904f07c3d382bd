#!/bin/bash
# round 5, GPU pass 6: two-workgroups-per-CU micro variant (k_duo), SQ counters of the shipped k_conv_fused (same sets as the micro-kernel's
# in r05_p5_counters_micro.txt), sidechain_pred on the GPU
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
( cd tools/micro && timeout 300 ./pc_ring 6 ) > $out/r05_micro_pc_ring_v3.txt 2>&1
grep -E "classic" $out/r05_micro_pc_ring_v3.txt
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "sidechain or tiny_l1" ) > $out/r05_p6_pytest.log 2>&1
tail -2 $out/r05_p6_pytest.log
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS"
: > $out/r05_p5_counters_kernel.txt
i=0
for p in "$P1" "$P2"; do
  i=$((i+1))
  DDMI_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --pmc $p -d /tmp/k$i -o k$i -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-serialised-pass > /tmp/k$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/k$i -name "*.db" | head -1) | grep k_conv_fused >> $out/r05_p5_counters_kernel.txt
done
grep "grid=  744448" $out/r05_p5_counters_kernel.txt | grep "ELi5E"
tail -2 /tmp/k1.log | cut -c 1-300
