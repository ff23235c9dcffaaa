#!/bin/bash
# round 5, visit 21: where the row form's time goes - compile-time ablations and K scaling (one split)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c21; mkdir -p $OUT; export TMPDIR=/tmp
export QUANTO_HIP_EXPERIMENT=1 QUANTO_HIP_CONV_SPLIT=1
for A in 0 1 2 4 16 64 23 87 128 215; do
  timeout 60 scripts/probes/conv_ablate_$A.bin | tee -a $OUT/conv_rows_ablations.jsonl
done
QUANTO_HIP_CONV_ROWS=0 timeout 60 scripts/probes/conv_ablate_0.bin | sed 's/"ablate": 0/"ablate": "taps"/' | tee -a $OUT/conv_rows_ablations.jsonl
