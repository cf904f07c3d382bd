#!/bin/bash
# round 5, GPU pass 37: k_edge_hidden_mm with the waves of a SIMD started a quarter cycle apart (HW_ID.wave_id) against the
# kernel without the offset (var_nostag.so); in-kernel clocks of the new form; parity tests first
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
B=diffdock_amd/csrc/build
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x ) > $out/r05_p28_pytest.log 2>&1
tail -2 $out/r05_p28_pytest.log
DDMI_STREAMS=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serialised-pass --lib $B/var_prof2.so > $out/r05_p28_prof2.json 2> $out/r05_p28_prof2.err
grep -E "EHPROF" $out/r05_p28_prof2.err | head -2
DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r05_e21 "A=1" "A=0 -- --lib $B/var_nostag.so" "A=1" "A=0 -- --lib $B/var_nostag.so" \
  "DDMI_STREAMS=1 -- --no-serialised-pass" "DDMI_STREAMS=1 -- --no-serialised-pass --lib $B/var_nostag.so" \
  "A=1 -- --samples 5" "A=0 -- --samples 5 --lib $B/var_nostag.so" "A=1 -- --config configs1" "A=0 -- --config configs1 --lib $B/var_nostag.so"
