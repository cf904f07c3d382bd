#!/bin/bash
# round 5, GPU pass 3: pose shards on one GPU (A/B), micro-kernel with the x-fragment window / spread requests
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export DDMI_HARNESS=1
( cd tools/micro && timeout 300 ./pc_ring 6 ) > $out/r05_micro_pc_ring_v2.txt 2>&1
grep -E "LOCK" $out/r05_micro_pc_ring_v2.txt
timeout 1500 tools/ab.sh r05_e2 "A=0" "A=1 -- --pose-shards 2" "A=2 -- --pose-shards 4" "A=4 -- --pose-shards 8" \
  "A=0" "A=1 -- --pose-shards 2" "A=5 -- --pose-shards 2 --config configs1" "A=6 -- --config configs1" "A=7 -- --pose-shards 2 --config configs4 --steps 1" "A=8 -- --config configs4 --steps 1"
