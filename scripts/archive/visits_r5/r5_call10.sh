#!/bin/bash
# round 5, visit 10: convolution epilogue with 8-byte stores / one division per four pixels (parity, ablations, timing)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c10; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --maxfail 10 --timeout 300 2>&1 | tail -4 | tee $OUT/conv_parity_tail.txt
for A in 0 9 63; do timeout 60 scripts/probes/conv_ablate_$A.bin 2>&1 | grep "^{" | tee -a $OUT/conv_ablations.jsonl; done
export QUANTO_HIP_EXPERIMENT=1
for W in qint8 qint4; do TIME_CONV2D_DIRECT_ONLY=1 timeout 200 python scripts/time_conv2d.py $W 2>&1 | grep "^{" | tee -a $OUT/conv_default_shapes.jsonl; done
TIME_CONV2D_DIRECT_ONLY=1 timeout 300 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" | tee -a $OUT/conv_grid.jsonl
