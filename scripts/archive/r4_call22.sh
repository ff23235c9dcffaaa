#!/bin/bash
# round 4, visit 22: the convolution's K split forced to 2 and 3 on the grid (which tile counts above 128 would gain from a split)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=$PWD/gpurun_out; mkdir -p $OUT
export QUANTO_HIP_EXPERIMENT=1
for s in 0 2 3; do
  QUANTO_HIP_CONV_SPLIT=$s timeout 200 python scripts/time_conv2d.py qint8 grid 2>/dev/null | grep "^{" | sed "s/^{/{\"forced_split\": $s, /" >> $OUT/r04_qconv2d_forced_split.jsonl
done
wc -l $OUT/r04_qconv2d_forced_split.jsonl
