#!/bin/bash
# round 4, visit 19: rocprofv3 kernel statistics of the late additions (group-96 streaming kernel, convolution kernels)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
D=$OUT/prof_r04_new; rm -rf $D
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o new -- python $REPO/scripts/profile_new_kernels.py > $D.log 2>&1)
f=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "^\"Name|qbits_skinny_kernel|qconv2d_" "$f" | cut -c1-300 > $OUT/r04_group96_and_int4_conv_kernel_stats.csv; cut -c1-220 $OUT/r04_group96_and_int4_conv_kernel_stats.csv
rm -rf $OUT/prof_r04_*
