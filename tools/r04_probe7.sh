#!/bin/bash
# round 4, session 3: the forward of the device loop as a captured HIP graph (ddmi_config.exec.step_graph) -- parity of the
# trajectories, A/B at 40 / 10 / 5 poses
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
( DDMI_STEP_GRAPH=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "device_loop or sharded or free_running or teacher" ) > $out/r04_p7_pytest.log 2>&1
tail -5 $out/r04_p7_pytest.log
timeout 1200 tools/ab.sh r04_e7 "DDMI_STEP_GRAPH=0" "DDMI_STEP_GRAPH=1" "DDMI_STEP_GRAPH=0" "DDMI_STEP_GRAPH=1" \
  "DDMI_STEP_GRAPH=0 -- --samples 5" "DDMI_STEP_GRAPH=1 -- --samples 5" "DDMI_STEP_GRAPH=0 -- --samples 5" "DDMI_STEP_GRAPH=1 -- --samples 5" \
  "DDMI_STEP_GRAPH=0 -- --config configs1" "DDMI_STEP_GRAPH=1 -- --config configs1"
