#!/bin/bash
# round 5, visit 4: where the batched-decode kernel's 9.8 us go - ring depth x split sweep, phase timeline
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c4; mkdir -p $OUT; export TMPDIR=/tmp
export QUANTO_HIP_EXPERIMENT=1
for SPLIT in 2 4 8; do
  QUANTO_HIP_SKINNY_SPLIT=$SPLIT timeout 200 python scripts/ab.py --rounds 5 --workloads int4_decode32 --env QUANTO_HIP_SKINNY_LDS_KB=50,76,100,150 2>&1 | grep -v Warning | sed "s/^{/{\"split\": $SPLIT, /" | tee -a $OUT/ring_depth_x_split.jsonl
done
for A in 0 1 3 19 31; do
  QUANTO_HIP_SKINNY_LDS_KB=100 QUANTO_HIP_SKINNY_ABLATE=$A timeout 100 python scripts/ab.py --rounds 5 --workloads int4_decode32 --env QUANTO_HIP_SKINNY_SPLIT=4 2>&1 | grep -v Warning | sed "s/^{/{\"ablate\": $A, \"lds_kb\": 100, /" | tee -a $OUT/ablate_deep_ring.jsonl
done
timeout 200 python scripts/skinny_timeline.py 2>&1 | grep -v Warning | tail -40 | tee $OUT/timeline.txt
