#!/bin/bash
# round 5, visit 6: convolution gather through range-checked buffer loads + two K-tiles in flight (parity, depth A/B); large-tile int4 GEMM with
# group sizes 96 / 32 (parity, timing against dequantize + dense); cfg2 wave layouts with conversion counts 1.5 / 0.75 VALU per MFMA + power;
# batched decode with 128-feature blocks
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c6; mkdir -p $OUT; export TMPDIR=/tmp
echo "== conv parity"
timeout 600 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider -x --timeout 300 2>&1 | tail -5 | tee $OUT/conv_parity_tail.txt
echo "== large4 HG parity"
timeout 400 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x --timeout 300 -k "large_tile_int4" 2>&1 | tail -5 | tee $OUT/large4_parity_tail.txt
export QUANTO_HIP_EXPERIMENT=1
echo "== conv depth A/B"
for D in 1 2; do for W in qint8 qint4; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_DEPTH=$D timeout 200 python scripts/time_conv2d.py $W 2>&1 | grep "^{" | tee -a $OUT/conv_depth_ab.jsonl
done; done
TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_DEPTH=1 timeout 300 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" | tee -a $OUT/conv_depth_grid.jsonl
TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_DEPTH=2 timeout 300 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" | tee -a $OUT/conv_depth_grid.jsonl
echo "== cfg2 layouts"
timeout 200 python scripts/ab.py --rounds 5 --sequential --workloads cfg2 --env QUANTO_HIP_LARGE_CFG=0,1,3 2>&1 | grep "^{" | tee $OUT/cfg2_layouts.jsonl
timeout 200 python scripts/power_probe.py --cfgs 0 1 3 --matmul --seconds 3 2>&1 | grep "^{" | tee $OUT/cfg2_layouts_power.jsonl
echo "== batched decode 128-feature blocks"
for SP in 4 8; do
  QUANTO_HIP_SKINNY_SPLIT=$SP timeout 100 python scripts/ab.py --rounds 5 --workloads int4_decode32 --env QUANTO_HIP_SKINNY_WAVES=0,8 2>&1 | grep "^{" | sed "s/^{/{\"split\": $SP, /" | tee -a $OUT/decode32_waves8.jsonl
done
echo "== large4 group 96 / 32 vs dequant + dense"
timeout 300 python scripts/time_group_sizes.py large 2>&1 | grep "^{" | tee $OUT/large4_group_sizes.jsonl
