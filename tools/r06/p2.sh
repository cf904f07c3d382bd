#!/bin/bash
# Round 6, pass 2: (a) workgroup stamps WITHOUT the phase clocks (build/var_wgs.so: -DDDMI_PROFILING=1 -DDDMI_WG_STAMPS), per-group
# launches vs grouped dispatch, at 40 / 10 / 5 poses; (b) grouped dispatch A/B on the production build; (c) what stalls the
# multi-rank step on a shared GPU: hardware-queue aliasing (GPU_MAX_HW_QUEUES), the collective behind the loop vs the loop drained
# first vs a host-tensor collective, over 6 timed steps (the slow steps of pass 1 came in a period-4 pattern).
cd $GRAFT_REPO_ROOT
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -k "selectable or ddl_synth" 2>&1 | tail -3
lib=diffdock_amd/csrc/build/var_wgs.so
for n in 40 10 5; do
  for grp in 1 2; do
    rm -f /tmp/wg.bin
    DDMI_GROUPED=$grp DDMI_WG_DUMP=/tmp/wg.bin python bench.py --lib $lib --steps 1 --warmup 1 --samples $n --no-cpu-baseline --no-serialised-pass > /tmp/wg.json 2> /tmp/wg.err
    nrec=$(grep FCWG /tmp/wg.err | head -1 | awk '{print $2}')
    echo "== $n poses, DDMI_GROUPED=$grp: $(python -c "import json;d=json.loads(open('/tmp/wg.json').read().strip().splitlines()[-1]);print(round(d['value'],1),'poses/s (stamp build)')")"
    python tools/wg_idle.py /tmp/wg.bin --records $nrec --forwards 40 > $out/r06_p2_wg_idle_b${n}_g$grp.txt 2>&1
    head -12 $out/r06_p2_wg_idle_b${n}_g$grp.txt
  done
done
bash tools/ab.sh r06_p2_b40 "DDMI_GROUPED=1" "DDMI_GROUPED=2" "DDMI_GROUPED=2 DDMI_GROUPED_YS=2" "DDMI_GROUPED=1" "DDMI_GROUPED=2"
bash tools/ab.sh r06_p2_b10 "DDMI_GROUPED=1 -- --samples 10" "DDMI_GROUPED=2 -- --samples 10" "DDMI_GROUPED=2 DDMI_GROUPED_YS=2 -- --samples 10" "DDMI_GROUPED=2 DDMI_GROUPED_YS=4 -- --samples 10" "DDMI_GROUPED=2 DDMI_GROUPED_YS=6 -- --samples 10"
bash tools/ab.sh r06_p2_b5 "DDMI_GROUPED=1 -- --samples 5" "DDMI_GROUPED=2 -- --samples 5" "DDMI_GROUPED=2 DDMI_GROUPED_YS=3 -- --samples 5" "DDMI_GROUPED=2 DDMI_GROUPED_YS=4 -- --samples 5" "DDMI_GROUPED=2 DDMI_GROUPED_YS=8 -- --samples 5"
run_share() {   # name, env, extra bench args
  name=$1; envs=$2; shift; shift
  env $envs DDMI_BENCH_SHARE_GPU=1 DDMI_BENCH_TRACE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 6 --warmup 1 --fixed-center-conv --scaling weak --samples 20 "$@" > $out/r06_p2_share2_$name.log 2>&1
  echo "== $name: seconds per step (rank 0)"
  grep "trace rank 0" $out/r06_p2_share2_$name.log | grep "all_gather returned\|enqueue sample_job" | awk '{print $4, $6, $7, $8}' | paste - - | awk '{printf "%.2f ", $5-$1} END {print ""}'
  tail -1 $out/r06_p2_share2_$name.log | cut -c 1-200
}
run_share enqueue "DDMI_BENCH_GATHER=enqueue"
run_share enqueue_hwq8 "DDMI_BENCH_GATHER=enqueue GPU_MAX_HW_QUEUES=8"
run_share enqueue_1stream "DDMI_BENCH_GATHER=enqueue DDMI_STREAMS=1 DDMI_GROUPED=1"
run_share host "DDMI_BENCH_GATHER=host"
run_share drain "DDMI_BENCH_GATHER=drain"
