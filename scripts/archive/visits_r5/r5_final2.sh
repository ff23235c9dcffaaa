#!/bin/bash
# round 5, closing visit on the tree with the convolution's row form: the convolution GPU suite, smoke, `python bench.py` exactly as the driver issues it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5final5; mkdir -p $OUT; export TMPDIR=/tmp
echo "== conv gpu tests"; timeout 600 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -4 | tee $OUT/r05_conv_gpu_tests_tail.txt
echo "== smoke"; timeout 200 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/r05_smoke_tail.txt
echo "== bench (driver command)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_bench_line_rows_tree.json 2> $OUT/r05_bench_rows_tree.err; echo "exit=$? bytes=$(wc -c < $OUT/r05_bench_line_rows_tree.json)"
