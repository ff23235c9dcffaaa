#!/bin/bash
# closing run of round 3, part e: kernel statistics of cfg5 decode (batch 1, hipGraph driver) at the final kernel state
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=$PWD/gpurun_out/r3final_e; mkdir -p $O; export TMPDIR=/tmp; REPO=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg5_b1 -o cfg5 -- python $REPO/scripts/bench_generate.py --batch 1 --prompt 512 --new 128 --drivers graph --fuse > $O/trace_cfg5_b1.jsonl 2> $O/trace_cfg5_b1.err)
f=$(find $O/trace_cfg5_b1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/cfg5_b1_decode_kernel_stats.csv
find $O/trace_cfg5_b1 -name "*kernel_trace.csv" -delete
cat $O/trace_cfg5_b1.jsonl | cut -c1-300; grep "qh::" $O/cfg5_b1_decode_kernel_stats.csv | cut -c1-200 | head -12
