#!/bin/bash
# the bench lines of every workload (round_profile.sh's last block) -- re-run after the scatter-row count was corrected
tag=${1:-r04_v1}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python bench.py --config configs1 > $out/${tag}_bench_configs1.json 2>> $out/${tag}_bench.err
python bench.py --config mix --steps 2 > $out/${tag}_bench_mix.json 2>> $out/${tag}_bench.err
python bench.py --config configs4 --steps 2 --warmup 1 > $out/${tag}_bench_configs4.json 2>> $out/${tag}_bench.err
python bench.py --samples 5 --no-cpu-baseline > $out/${tag}_bench_b5.json 2>> $out/${tag}_bench.err
python bench.py --all-atoms > $out/${tag}_bench_all_atoms.json 2>> $out/${tag}_bench.err
python bench.py --edge-product bf16x4 --no-cpu-baseline > $out/${tag}_bench_bf16x4.json 2>> $out/${tag}_bench.err
tail -c 1500 $out/${tag}_bench.json
