#!/bin/bash
# round 4, session 3: timing-only upper bound of a cheaper edge product -- build without the edge-product MFMAs (FCV_NOEMMA, garbage scores)
cd $GRAFT_REPO_ROOT
DDMI_BENCH_NOCHECK=1 DDMI_TIME_GROUPS=1 timeout 1200 tools/ab.sh r04_e9 "A=0 -- --all-atoms" "A=1 -- --all-atoms --lib diffdock_amd/csrc/build/var_noemma.so" \
  "A=0" "A=1 -- --lib diffdock_amd/csrc/build/var_noemma.so"
