#!/bin/bash
# round 5, visit 8: convolution kernel with the magic-number tap table (parity, timing), compile-time ablations of the convolution kernel (where a
# K-tile's 2 us go), the vendor's 8-bit GEMMs next to qmm_native8
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c8; mkdir -p $OUT; export TMPDIR=/tmp
echo "== conv parity"
timeout 900 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --maxfail 10 --timeout 300 2>&1 | tail -4 | tee $OUT/conv_parity_tail.txt
echo "== conv ablations"
for A in 0 1 2 4 9 16 17 32 63; do
  [ -x scripts/probes/conv_ablate_$A.bin ] && timeout 60 scripts/probes/conv_ablate_$A.bin 2>&1 | grep "^{" | tee -a $OUT/conv_ablations.jsonl
done
export QUANTO_HIP_EXPERIMENT=1
echo "== conv timing"
for W in qint8 qint4; do
  TIME_CONV2D_DIRECT_ONLY=1 timeout 200 python scripts/time_conv2d.py $W 2>&1 | grep "^{" | tee -a $OUT/conv_default_shapes.jsonl
done
echo "== vendor 8-bit GEMMs"
timeout 200 python scripts/dense_reference_point.py --eight-bit 4096x4096x4096 512x8192x8192 2>&1 | grep "^{" | tee $OUT/vendor_reference_points.jsonl
