#!/bin/bash
# round 5, last GPU pass (2): the multi-rank bench path once more on ONE GPU (2 ranks over gloo, shard check) with the final bench.py
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
DDMI_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 2 --warmup 1 --verify-shards > $out/r05_final_bench_share2.log 2>&1
tail -1 $out/r05_final_bench_share2.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['n_gpus'], d['scaling'], d['config'].get('tile_per_pose'), d['shard_check'])
"
