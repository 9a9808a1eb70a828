#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/bench_records.sh r2'): the bench.py records committed under profiles/ --
# the driver's default command, the Dirichlet and training variants, 2048x1024 in bf16 and fp8, the per-layer table.
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
python3 bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
python3 bench.py --fusion dirichlet --no-accuracy --no-extra > $OUT/${TAG}_bench_dirichlet.json 2>/dev/null
python3 bench.py --mode train --no-accuracy --no-extra > $OUT/${TAG}_bench_train.json 2>/dev/null
python3 bench.py --height 1024 --width 2048 --batch 4 --steps 10 --no-accuracy --no-extra > $OUT/${TAG}_bench_2048.json 2>/dev/null
python3 bench.py --dtype fp8 --height 1024 --width 2048 --batch 4 --steps 10 --no-accuracy --no-extra > $OUT/${TAG}_bench_fp8_2048.json 2>/dev/null
python3 bench.py --layer-profile --no-accuracy --no-extra --no-cpu-baseline 2>&1 >/dev/null | grep "conv launch" > $OUT/${TAG}_bench_layers.txt
# round 5: the label-exact mode with its per-layer table, the host boundary, batch 1, batch-norm training
python3 tools/exact_bench.py 16 10 > $OUT/${TAG}_bench_exact.json 2> $OUT/${TAG}_bench_exact_layers.txt
XV_EXACT_SCALAR=1 python3 tools/exact_bench.py 4 2 > $OUT/${TAG}_bench_exact_scalar.json 2>/dev/null
python3 tools/host_path_bench.py 256 16 > $OUT/${TAG}_host_path.json 2>/dev/null
XV_HOST_PIPELINE=0 python3 tools/host_path_bench.py 128 16 > $OUT/${TAG}_host_path_serial.json 2>/dev/null
python3 bench.py --batch 1 --steps 50 --warmup 5 --layer-profile --no-cpu-baseline --no-accuracy --no-extra > $OUT/${TAG}_bench_b1.json 2> $OUT/${TAG}_bench_b1_layers.txt
python3 bench.py --mode train --batch-norm --no-accuracy --no-extra > $OUT/${TAG}_bench_train_bn.json 2>/dev/null
for f in default dirichlet train train_bn 2048 fp8_2048 exact host_path b1; do cut -c1-260 $OUT/${TAG}_bench_$f.json 2>/dev/null || cut -c1-260 $OUT/${TAG}_$f.json; done
