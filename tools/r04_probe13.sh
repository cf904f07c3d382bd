#!/bin/bash
# round 4, session 3: what a k_conv_fused workgroup costs without main loops / coupling / message stores (DDMI_ABLATE bits of the
# -DDDMI_PROFILING=1 build, frozen poses): 128 = no main loops, +32 = no coupling phase, +256 = no message stores
cd $GRAFT_REPO_ROOT
B=diffdock_amd/csrc/build
DDMI_FREEZE_POSE=1 DDMI_BENCH_NOCHECK=1 DDMI_TIME_GROUPS=1 timeout 900 tools/ab.sh r04_e13 "DDMI_ABLATE=0 -- --lib $B/var_prof.so" "DDMI_ABLATE=128 -- --lib $B/var_prof.so" \
  "DDMI_ABLATE=384 -- --lib $B/var_prof.so" "DDMI_ABLATE=160 -- --lib $B/var_prof.so" "DDMI_ABLATE=416 -- --lib $B/var_prof.so" \
  "DDMI_ABLATE=256 -- --lib $B/var_prof.so" "DDMI_ABLATE=32 -- --lib $B/var_prof.so" \
  "DDMI_STREAMS=1 DDMI_ABLATE=0 -- --lib $B/var_prof.so" "DDMI_STREAMS=1 DDMI_ABLATE=128 -- --lib $B/var_prof.so" "DDMI_STREAMS=1 DDMI_ABLATE=416 -- --lib $B/var_prof.so"
