#!/bin/bash
# round 5, visit 15: convolution pair gather with a per-tap LDS table (parity incl. bit-equality of the three gather forms, A/B, ablations)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c15; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --maxfail 10 --timeout 300 2>&1 | tail -4 | tee $OUT/conv_parity_tail.txt
for A in 0 9 63; do timeout 60 scripts/probes/conv_ablate_$A.bin 2>&1 | grep "^{" | tee -a $OUT/conv_ablations.jsonl; done
export QUANTO_HIP_EXPERIMENT=1
for T in 0 1; do for W in qint8 qint4; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_TAB=$T timeout 200 python scripts/time_conv2d.py $W 2>&1 | grep "^{" | sed "s/^{/{\"tab\": $T, /" | tee -a $OUT/conv_tab_ab.jsonl
done; done
for T in 0 1; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_TAB=$T timeout 300 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" | sed "s/^{/{\"tab\": $T, /" | tee -a $OUT/conv_tab_grid.jsonl
done
