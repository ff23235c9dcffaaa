#!/bin/bash
# round 5, visit 23: convolution epilogue through LDS (full 256-byte runs per channel plane): whole conv GPU suite, probes, graph timings, bench sub-result
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c23; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --maxfail 12 --timeout 300 2>&1 | tail -6 | tee $OUT/conv_parity_tail.txt
( export QUANTO_HIP_EXPERIMENT=1 QUANTO_HIP_CONV_SPLIT=1
  for A in 0 128; do timeout 60 scripts/probes/conv_ablate_$A.bin | tee -a $OUT/conv_rows_ablations.jsonl; done )
( export QUANTO_HIP_EXPERIMENT=1
  for W in qint8 qint4; do TIME_CONV2D_DIRECT_ONLY=1 timeout 200 python scripts/time_conv2d.py $W 2>&1 | grep "^{" | tee -a $OUT/conv_direct.jsonl; done
  TIME_CONV2D_DIRECT_ONLY=1 timeout 300 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" | tee -a $OUT/conv_direct_grid.jsonl )
timeout 400 python bench.py --sub qconv2d_3x3 --no-cfg5 --no-cpu-baseline --profile > $OUT/bench_conv_sub.json 2> $OUT/bench_conv_sub.err; tail -c 1500 $OUT/bench_conv_sub.json
