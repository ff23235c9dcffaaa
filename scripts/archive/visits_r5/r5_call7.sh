#!/bin/bash
# round 5, visit 7: convolution gather with two output pixels per load (parity incl. bit-equality with the one-pixel gather, A/B); large-tile int4
# GEMM group sizes 96 / 32 after the lane-group fix (parity); large-tile int4 GEMM against dequantize + dense at the sizes AUTO hands it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c7; mkdir -p $OUT; export TMPDIR=/tmp
echo "== conv parity"
timeout 900 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --maxfail 10 --timeout 300 2>&1 | tail -15 | tee $OUT/conv_parity_tail.txt
echo "== large4 HG parity"
timeout 400 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider --maxfail 10 --timeout 300 -k "large_tile_int4" 2>&1 | tail -8 | tee $OUT/large4_parity_tail.txt
export QUANTO_HIP_EXPERIMENT=1
echo "== conv pair A/B"
for PAIRV in 0 1; do for W in qint8 qint4; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_PAIR=$PAIRV timeout 200 python scripts/time_conv2d.py $W 2>&1 | grep "^{" | sed "s/^{/{\"pair\": $PAIRV, /" | tee -a $OUT/conv_pair_ab.jsonl
done; done
for PAIRV in 0 1; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_PAIR=$PAIRV timeout 300 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" | sed "s/^{/{\"pair\": $PAIRV, /" | tee -a $OUT/conv_pair_grid.jsonl
done
echo "== large4 vs dequant + dense"
timeout 200 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -k "auto_takes_the_large" 2>&1 | tail -3 | tee $OUT/large4_auto_tail.txt
