#!/bin/bash
# round 5, GPU pass 13: the whole library without SLP vectorisation (-fno-slp-vectorize: no v_pk_* packing of the epilogue's scalar
# f32 arithmetic, MI355X_MICROARCH.md "packed f32 VALU is an anti-lever") against the default build, same box
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
B=diffdock_amd/csrc/build
DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r05_e9 "A=1" "A=0 -- --lib $B/var_noslp.so" "A=1" "A=0 -- --lib $B/var_noslp.so" \
  "A=1 -- --samples 5" "A=0 -- --samples 5 --lib $B/var_noslp.so"
