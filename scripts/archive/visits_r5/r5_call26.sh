#!/bin/bash
# round 5, visit 26: two groups one phase apart (multiply while the other stages): parity of the row-form tests, A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c26; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --maxfail 12 --timeout 300 -k "row_form" 2>&1 | tail -6 | tee $OUT/conv_parity_tail.txt
export QUANTO_HIP_EXPERIMENT=1
for V in "1 1" "2 0" "2 1"; do set -- $V
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_ROWS_GROUPS=$1 QUANTO_HIP_CONV_ROWS_SKEW=$2 timeout 200 python scripts/time_conv2d.py qint8 2>&1 | grep "^{" | tee -a $OUT/conv_rows_skew_ab.jsonl
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_ROWS_GROUPS=$1 QUANTO_HIP_CONV_ROWS_SKEW=$2 timeout 300 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" | tee -a $OUT/conv_rows_skew_ab.jsonl
done
