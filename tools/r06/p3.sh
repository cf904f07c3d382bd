#!/bin/bash
# Round 6, pass 3: whole GPU suite on the new default route (merged virtual-node list build, fused node update), A/B of both
# against the round-5 forms at 40 / 10 / 5 poses, kernel-trace timelines at 5 and 40 poses.
cd $GRAFT_REPO_ROOT
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp DDMI_HARNESS=1
( time python -m pytest tests -m gpu -x -q ) > $out/r06_p3_pytest_gpu.log 2>&1
tail -4 $out/r06_p3_pytest_gpu.log
bash tools/ab.sh r06_p3_b40 "DDMI_NODE_UPDATE=0 DDMI_VN_BUILD=1" "DDMI_X=1" "DDMI_NODE_UPDATE=0" "DDMI_VN_BUILD=1" "DDMI_NODE_UPDATE=0 DDMI_VN_BUILD=1" "DDMI_X=1" "DDMI_GROUPED=2"
bash tools/ab.sh r06_p3_b10 "DDMI_NODE_UPDATE=0 DDMI_VN_BUILD=1 -- --samples 10" "DDMI_X=1 -- --samples 10" "DDMI_NODE_UPDATE=0 -- --samples 10" "DDMI_VN_BUILD=1 -- --samples 10" "DDMI_GROUPED=2 -- --samples 10"
bash tools/ab.sh r06_p3_b5 "DDMI_NODE_UPDATE=0 DDMI_VN_BUILD=1 -- --samples 5" "DDMI_X=1 -- --samples 5" "DDMI_NODE_UPDATE=0 -- --samples 5" "DDMI_VN_BUILD=1 -- --samples 5" "DDMI_GROUPED=2 -- --samples 5" "DDMI_GROUPED=2 DDMI_FUSED_YS=8 -- --samples 5"
cd /tmp
for n in 5 40; do
  rm -rf /tmp/prof_b$n
  rocprofv3 --kernel-trace --stats -d /tmp/prof_b$n -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --samples $n --no-cpu-baseline --no-serialised-pass > /tmp/kt_$n.log 2>&1
  python $GRAFT_REPO_ROOT/tools/timeline.py $(find /tmp/prof_b$n -name "*.db" | head -1) k_perturb 10 22 > $GRAFT_REPO_ROOT/$out/r06_p3_timeline_b$n.txt 2>&1
  head -45 $GRAFT_REPO_ROOT/$out/r06_p3_timeline_b$n.txt
done
