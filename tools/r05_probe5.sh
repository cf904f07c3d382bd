#!/bin/bash
# round 5, GPU pass 5: the same SQ counters for the shipped k_conv_fused and for the lock-step micro-kernel (tools/micro/pc_ring.hip,
# 34 cycles per MFMA): what does the real kernel spend per MFMA that the micro-kernel does not?
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export DDMI_HARNESS=1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS"
: > $out/r05_p5_counters_micro.txt
: > $out/r05_p5_counters_kernel.txt
i=0
for p in "$P1" "$P2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $p -d /tmp/m$i -o m$i -- $GRAFT_REPO_ROOT/tools/micro/pc_ring 6 > /tmp/m$i.log 2>&1
  python - $(find /tmp/m$i -name "*.db" | head -1) >> $out/r05_p5_counters_micro.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
q = """select s.kernel_name, p.name, count(distinct d.id), sum(e.value) * 1.0 / count(distinct d.id), avg(d.end - d.start) / 1e3
       from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id
       join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name order by 1, 2"""
for r in cur.execute(q):
    print(f"{r[0][:40]:40s} {r[1]:28s} n={r[2]:3d} per_dispatch={r[3]:.5g} avg_us={r[4]:.1f}")
PY
  DDMI_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --pmc $p -d /tmp/k$i -o k$i -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-serialised-pass > /tmp/k$i.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/k$i -name "*.db" | head -1) | grep k_conv_fused >> $out/r05_p5_counters_kernel.txt
done
grep -E "k_lockILi1ELb0ELb0|k_lockILi0ELb0ELb0" $out/r05_p5_counters_micro.txt | head -40
grep "grid=  744448" $out/r05_p5_counters_kernel.txt | grep "ELi5E"
