#!/bin/bash
# round 5, visit 11: the convolution's K split re-swept with the cheaper gather / epilogue
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c11; mkdir -p $OUT; export TMPDIR=/tmp
export QUANTO_HIP_EXPERIMENT=1
for S in 0 1 2 3 4 6 9; do
  TIME_CONV2D_DIRECT_ONLY=1 QUANTO_HIP_CONV_SPLIT=$S timeout 300 python scripts/time_conv2d.py qint8 grid 2>&1 | grep "^{" | tee -a $OUT/conv_split_sweep.jsonl
done
