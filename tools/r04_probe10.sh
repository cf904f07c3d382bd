#!/bin/bash
# round 4, session 3: what the per-chunk skeleton of k_conv_fused is made of -- timing-only builds (garbage scores):
# FCV_NOYST no chunk stores, FCV_NOXW no x-fragment reads (packed granules), FCV_NOQ no chunk reads of the edge product,
# skel = all three + no edge-product MFMAs
cd $GRAFT_REPO_ROOT
B=diffdock_amd/csrc/build
DDMI_BENCH_NOCHECK=1 DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r04_e10 "A=0" "A=1 -- --lib $B/var_noyst.so" "A=2 -- --lib $B/var_noxw.so" "A=3 -- --lib $B/var_noq.so" \
  "A=4 -- --lib $B/var_noemma.so" "A=5 -- --lib $B/var_skel.so" "A=0" "A=1 -- --lib $B/var_noyst.so"
