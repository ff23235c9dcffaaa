#!/bin/bash
# round 5, visit 27: K-split sweep for the row form (its K-tiles are 96 deep and cheaper than the tap gather's: is the tap kernel's split rule still right?)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c27; mkdir -p $OUT; export TMPDIR=/tmp
export QUANTO_HIP_EXPERIMENT=1
for S in 1 2 3 6 12; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_SPLIT=$S timeout 200 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" | grep rows | tee -a $OUT/conv_rows_split_sweep.jsonl
done
