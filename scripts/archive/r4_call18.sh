#!/bin/bash
# round 4, visit 18: pointwise convolutions through the convolution kernel against permute + GEMM and the reference's path
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 400 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" > $OUT/r04_qconv2d_paths_grid.jsonl
timeout 400 python scripts/time_conv2d.py qint4 grid 2>&1 | grep "^{" >> $OUT/r04_qconv2d_paths_grid.jsonl
wc -l $OUT/r04_qconv2d_paths_grid.jsonl
