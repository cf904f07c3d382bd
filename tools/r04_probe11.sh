#!/bin/bash
# round 4, session 3: per-chunk skeleton of k_conv_fused with FROZEN poses (-DDDMI_PROFILING=1 builds, DDMI_FREEZE_POSE=1: the
# timing-only variants produce garbage scores, which must not move the ligands and change the graphs)
cd $GRAFT_REPO_ROOT
B=diffdock_amd/csrc/build
DDMI_FREEZE_POSE=1 DDMI_BENCH_NOCHECK=1 DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r04_e11 "A=0 -- --lib $B/var_prof.so" "A=1 -- --lib $B/var_pnoemma.so" \
  "A=2 -- --lib $B/var_pnoyst.so" "A=3 -- --lib $B/var_pnoq.so" "A=4 -- --lib $B/var_pskel.so" "A=0 -- --lib $B/var_prof.so" \
  "DDMI_ABLATE=128 -- --lib $B/var_prof.so" "A=0 -- --lib $B/var_prof.so --all-atoms" "A=1 -- --lib $B/var_pnoemma.so --all-atoms" "A=4 -- --lib $B/var_pskel.so --all-atoms"
