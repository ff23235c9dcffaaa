#!/bin/bash
# round 5, visit 16: the convolution's K split reduced inside the kernel (parity, run-to-run bits, timing, split re-sweep)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c16; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_qconv2d.py tests/test_reference_style_gpu.py -m gpu -q -p no:cacheprovider --maxfail 10 --timeout 300 2>&1 | tail -4 | tee $OUT/conv_parity_tail.txt
for A in 0 63; do timeout 60 scripts/probes/conv_ablate_$A.bin 2>&1 | grep "^{" | tee -a $OUT/conv_ablations.jsonl; done
export QUANTO_HIP_EXPERIMENT=1
for S in 0 2 3 4 6 9; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_SPLIT=$S timeout 300 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" | tee -a $OUT/conv_split_sweep.jsonl
done
TIME_CONV2D_DIRECT_ONLY=1 timeout 200 python scripts/time_conv2d.py qint4 2>&1 | grep "^{" | tee -a $OUT/conv_int4.jsonl
