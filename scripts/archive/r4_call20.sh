#!/bin/bash
# round 4, visit 20: the driver's bench command with the QConv2d record
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_default_bench_line.json 2> $OUT/r04_default_bench.err; echo "exit=$? bytes=$(wc -c < $OUT/r04_default_bench_line.json)"
tail -3 $OUT/r04_default_bench.err | cut -c1-300
