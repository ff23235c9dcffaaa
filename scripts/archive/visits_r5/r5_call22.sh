#!/bin/bash
# round 5, visit 22: row form with the hand-scheduled MFMA phase (loads first, conversion after the MFMAs): parity, ablations incl. load-shape variants
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c22; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --maxfail 12 --timeout 300 -k "row_form" 2>&1 | tail -6 | tee $OUT/conv_rows_parity_tail.txt
export QUANTO_HIP_EXPERIMENT=1 QUANTO_HIP_CONV_SPLIT=1
for A in 0 1 4 5 23 256 512; do
  timeout 60 scripts/probes/conv_ablate_$A.bin | tee -a $OUT/conv_rows_ablations.jsonl
done
