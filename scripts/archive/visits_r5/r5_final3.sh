#!/bin/bash
# round 5, last visit: the row form with one pixel per thread (any width stride, odd OW): the whole convolution GPU file, then A/B against the tap gather
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5final6; mkdir -p $OUT; export TMPDIR=/tmp
timeout 75 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --maxfail 8 --timeout 60 2>&1 | tail -12 | tee $OUT/r05_conv_gpu_tests_final_tail.txt
export QUANTO_HIP_EXPERIMENT=1
for W in qint8 qint4; do for R in 1 0; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_ROWS=$R timeout 20 python scripts/time_conv2d.py $W strided 2>&1 | grep "^{" | tee -a $OUT/conv_rows_one_pixel_ab.jsonl
done; done
