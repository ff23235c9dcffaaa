#!/bin/bash
# round 5, visit 25: row form with sixteen waves in two groups (grids of one workgroup per CU): conv GPU suite, A/B against eight waves
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c25; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --maxfail 12 --timeout 300 2>&1 | tail -15 | tee $OUT/conv_parity_tail.txt
export QUANTO_HIP_EXPERIMENT=1
for G in 1 2; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_ROWS_GROUPS=$G timeout 200 python scripts/time_conv2d.py qint8 2>&1 | grep "^{" | tee -a $OUT/conv_rows_groups_ab.jsonl
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_ROWS_GROUPS=$G timeout 300 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" | tee -a $OUT/conv_rows_groups_ab.jsonl
done
TIME_CONV2D_DIRECT_ONLY=1 timeout 200 python scripts/time_conv2d.py qint4 2>&1 | grep "^{" | tee -a $OUT/conv_int4_auto.jsonl
