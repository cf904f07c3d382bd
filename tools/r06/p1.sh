#!/bin/bash
# Round 6, pass 1: (a) today's baseline lines at 40 / 10 / 5 poses, (b) workgroup stamps of k_conv_fused (profiling build
# build/var_wg.so) at 40 / 10 / 5 poses -> CU-idle time inside the fused launches (tools/wg_idle.py), (c) the multi-rank strong-path
# anomaly of round 5 (38-68 s per step on a shared GPU) with per-rank wall clocks (DDMI_BENCH_TRACE).
cd $GRAFT_REPO_ROOT
out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 1 > $out/r06_p1_bench.json 2> $out/r06_p1_bench.err
python bench.py --steps 3 --warmup 1 --config configs1 --no-cpu-baseline > $out/r06_p1_bench_configs1.json 2>> $out/r06_p1_bench.err
python bench.py --steps 3 --warmup 1 --samples 5 --no-cpu-baseline > $out/r06_p1_bench_b5.json 2>> $out/r06_p1_bench.err
for f in $out/r06_p1_bench.json $out/r06_p1_bench_configs1.json $out/r06_p1_bench_b5.json; do
  python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1], "poses/s", round(d["value"], 1), "wall", round(r["frac"], 3), "ser", round((r.get("serialised") or {}).get("frac", 0), 3))
PY
done
lib=diffdock_amd/csrc/build/var_wg.so
for n in 40 10 5; do
  rm -f /tmp/wg_$n.bin*
  DDMI_WG_DUMP=/tmp/wg_$n.bin python bench.py --lib $lib --steps 1 --warmup 1 --samples $n --no-cpu-baseline --no-serialised-pass > /tmp/wg_$n.json 2> /tmp/wg_$n.err
  grep FCWG /tmp/wg_$n.err | head -3
  nrec=$(grep FCWG /tmp/wg_$n.err | head -1 | awk '{print $2}')
  python tools/wg_idle.py /tmp/wg_$n.bin --records $nrec --forwards 40 > $out/r06_p1_wg_idle_b$n.txt 2>&1
  cat $out/r06_p1_wg_idle_b$n.txt
done
run_share() {   # name, extra bench args
  name=$1; shift
  DDMI_BENCH_SHARE_GPU=1 DDMI_BENCH_TRACE=${TRACE:-1} timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 2 --warmup 1 --fixed-center-conv --verify-shards "$@" > $out/r06_p1_share2_$name.log 2>&1
  grep "trace rank" $out/r06_p1_share2_$name.log | head -80
  tail -1 $out/r06_p1_share2_$name.log | cut -c 1-300
}
run_share strong
TRACE=2 run_share strong_sync
run_share weak --scaling weak
run_share notpp --no-tile-per-pose
run_share s80 --samples 80
