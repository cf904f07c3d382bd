#!/bin/bash
# Round 6, pass 5: k_node_update with the weight fragments requested in front of the barrier + two MFMA chains, against
# k_reduce_bn + GEMM launches; upper bound of re-ordering the attribute rows of k_edge_hidden_mm (timing-only build EHV_SEQATTR,
# frozen poses).
cd $GRAFT_REPO_ROOT
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp DDMI_HARNESS=1
( time python -m pytest tests/test_gpu_parity.py -x -q -k "selectable or grouped" ) > $out/r06_p5_pytest_gpu.log 2>&1
tail -4 $out/r06_p5_pytest_gpu.log
bash tools/ab.sh r06_p5_b40 "DDMI_NODE_UPDATE=0" "DDMI_X=1" "DDMI_NODE_UPDATE=0" "DDMI_X=1"
bash tools/ab.sh r06_p5_b5 "DDMI_NODE_UPDATE=0 -- --samples 5" "DDMI_X=1 -- --samples 5" "DDMI_NODE_UPDATE=0 -- --samples 5" "DDMI_X=1 -- --samples 5"
B=diffdock_amd/csrc/build
bash tools/ab.sh r06_p5_seqattr "DDMI_FREEZE_POSE=1 DDMI_BENCH_NOCHECK=1 -- --lib $B/var_p1.so" "DDMI_FREEZE_POSE=1 DDMI_BENCH_NOCHECK=1 -- --lib $B/var_p1seq.so" \
  "DDMI_FREEZE_POSE=1 DDMI_BENCH_NOCHECK=1 -- --lib $B/var_p1.so" "DDMI_FREEZE_POSE=1 DDMI_BENCH_NOCHECK=1 -- --lib $B/var_p1seq.so" \
  "DDMI_FREEZE_POSE=1 DDMI_BENCH_NOCHECK=1 DDMI_STREAMS=1 -- --lib $B/var_p1.so --no-serialised-pass" "DDMI_FREEZE_POSE=1 DDMI_BENCH_NOCHECK=1 DDMI_STREAMS=1 -- --lib $B/var_p1seq.so --no-serialised-pass"
