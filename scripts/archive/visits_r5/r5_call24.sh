#!/bin/bash
# round 5, visit 24: int4 / int2 weights on the row form (dequantize once + dense row form): conv GPU suite, graph timings
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c24; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --maxfail 12 --timeout 300 2>&1 | tail -15 | tee $OUT/conv_parity_tail.txt
export QUANTO_HIP_EXPERIMENT=1
for R in 1 0; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_ROWS=$R timeout 200 python scripts/time_conv2d.py qint4 2>&1 | grep "^{" | tee -a $OUT/conv_int4_rows_ab.jsonl
done
