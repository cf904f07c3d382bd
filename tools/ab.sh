#!/bin/bash
# A/B timing on the GPU box: tools/ab.sh <tag> "<env assignments>[ -- extra bench args]" ...   -> gpurun_out/<tag>_ab.txt
# every variant: python bench.py --steps 2 --warmup 1 --no-cpu-baseline under the given environment
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_ab.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
: > $out
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  envpart="${v%% -- *}"; extra=""
  if [[ "$v" == *" -- "* ]]; then extra="${v#* -- }"; fi
  line=$(env $envpart python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-small-batch $extra 2>/tmp/ab.err | tail -1)
  python - "$v" "$line" >> $out <<'PY'
import json, sys
v, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    p = d["phase_ms_per_forward"]
    r = d.get("roofline") or {}
    s = (r.get("serialised") or {})
    print(f"{v:60s} poses/s={d['value']:7.1f} fwd={p.get('forward_total',0):6.2f} fused={p.get('k_conv_fused',0):6.2f} "
          f"hidden={p.get('k_edge_hidden',0):5.2f} gemms={p.get('conv_fc1_gemms',0):5.2f} reduce={p.get('k_reduce_bn',0):5.2f} "
          f"frac={r.get('frac',0):.3f} ser={s.get('frac',0) or 0:.3f}"
          + "".join(f" {k[13:]}={x:.2f}" for k, x in sorted(p.items()) if k.startswith("k_conv_fused:")))
except Exception as e:
    print(f"{v:60s} FAILED {e} {line[:200]}")
    print(open('/tmp/ab.err').read()[-600:])
PY
done
cat $out
