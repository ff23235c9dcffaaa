#!/bin/bash
# round 4, visit 17+: QConv2d - parity of the implicit-GEMM kernels, then the paths over a grid of shapes (implicit GEMM / im2col + GEMM / the
# reference's dequantize + float convolution)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6
timeout 400 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" > $OUT/r04_qconv2d_paths_grid.jsonl
timeout 400 python scripts/time_conv2d.py qint4 grid 2>&1 | grep "^{" >> $OUT/r04_qconv2d_paths_grid.jsonl
wc -l $OUT/r04_qconv2d_paths_grid.jsonl
