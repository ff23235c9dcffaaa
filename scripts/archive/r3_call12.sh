#!/bin/bash
# round 3, GPU visit 12: fixed dispatch expectations + new tests, cfg5 by the reference's method, decode kernel stats, GEMV vs MFMA decode at M = 1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3i; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -k "c_api_multi or auto_picks or batched_decode_llama or plugin_mode_runs or llama3_8b_layer or gemv_other_group or gemv_int2 or fused4" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
QUANTO_HIP_MMV_MAX_N=16384 timeout 600 python scripts/ab.py --workloads northstar cfg3 --env QUANTO_HIP_GEMV_MAX_M=4,0 --rounds 7 > $O/ab_gemv_vs_mmv.jsonl 2> $O/ab_gemv.err; cat $O/ab_gemv_vs_mmv.jsonl; tail -2 $O/ab_gemv.err
timeout 1500 python scripts/bench_generate.py --batch 1 32 --prompt 512 --new 512 --drivers reference graph --iterations 2 --fuse > $O/cfg5_generation.jsonl 2> $O/cfg5_generation.err; cat $O/cfg5_generation.jsonl; tail -3 $O/cfg5_generation.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/trace_cfg5_b1 -o cfg5 -- python $OLDPWD/scripts/bench_generate.py --batch 1 --prompt 512 --new 128 --drivers graph --fuse > $OLDPWD/$O/trace_cfg5_b1.jsonl 2> $OLDPWD/$O/trace_cfg5_b1.err)
f=$(find $O/trace_cfg5_b1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/cfg5_b1_decode_kernel_stats.csv && head -25 $O/cfg5_b1_decode_kernel_stats.csv
find $O/trace_cfg5_b1 -name "*kernel_trace.csv" -delete
cat $O/trace_cfg5_b1.jsonl
