#!/bin/bash
# round 5, visit 20: the ROW form of the convolution (three-tap-wide windows at stride 1): parity (both load variants), A/B against the tap gather
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c20; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --maxfail 12 --timeout 300 -k "row_form or pair_gather or implicit_gemm_gpu or k_split" 2>&1 | tail -25 | tee $OUT/conv_rows_parity_tail.txt
export QUANTO_HIP_EXPERIMENT=1
for R in 0 1 2; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_ROWS=$R timeout 200 python scripts/time_conv2d.py qint8 2>&1 | grep "^{" | tee -a $OUT/conv_rows_ab.jsonl
done
for R in 0 1; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_ROWS=$R timeout 300 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" | tee -a $OUT/conv_rows_grid.jsonl
done
