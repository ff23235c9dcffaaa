#!/bin/bash
# round 5, visit 19: convolution with 64-pixel workgroups (four waves) on grids of up to 256 full tiles (parity incl. bit-equality with the 128-pixel form, A/B)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c19; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --maxfail 10 --timeout 300 2>&1 | tail -4 | tee $OUT/conv_parity_tail.txt
export QUANTO_HIP_EXPERIMENT=1
for T in 128 64; do for W in qint8 qint4; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_BMT=$T timeout 200 python scripts/time_conv2d.py $W 2>&1 | grep "^{" | sed "s/^{/{\"bmt\": $T, /" | tee -a $OUT/conv_bmt_ab.jsonl
done; done
for T in 128 64; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_BMT=$T timeout 300 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" | sed "s/^{/{\"bmt\": $T, /" | tee -a $OUT/conv_bmt_grid.jsonl
done
