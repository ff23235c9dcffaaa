#!/bin/bash
# round 5, visit 9: fixed vs per-K cost of the 8-bit GEMMs, ours and the vendor's; the vendor kernels' names
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c9; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python scripts/native8_k_scaling.py 2>&1 | grep "^{" | tee $OUT/native8_k_scaling.jsonl
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vendor_trace -o v -- python $REPO/scripts/dense_reference_point.py --eight-bit 4096x4096x4096 > /dev/null 2> $OUT/vendor_trace.log)
f=$(find $OUT/vendor_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-400 "$f" | head -12 | tee $OUT/vendor_kernel_stats_head.csv
find $OUT/vendor_trace -name "*kernel_trace.csv" -delete
