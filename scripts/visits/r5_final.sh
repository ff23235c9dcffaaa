#!/bin/bash
# round 5, closing visit: full GPU suite, smoke, `python bench.py` exactly as the driver issues it, the rocprofv3 kernel statistics of that same
# command, and the statistics of the workloads whose kernels changed this round.  Outputs under gpurun_out/r5final (copied to profiles/r05_* afterwards).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5final; mkdir -p $OUT; export TMPDIR=/tmp
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | tail -6 | tee $OUT/r05_gpu_tests_tail.txt
echo "== perf-marked tests (timing assertions, not part of the parity run)"; timeout 300 python -m pytest tests -m perf -q -p no:cacheprovider --timeout 300 2>&1 | tail -3 | tee $OUT/r05_perf_tests_tail.txt
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/r05_smoke_tail.txt
echo "== bench (driver command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_default_bench_line.json 2> $OUT/r05_default_bench.err; echo "exit=$? bytes=$(wc -c < $OUT/r05_default_bench_line.json)"
echo "== kernel stats of the same command"
D=$OUT/prof_default; rm -rf $D
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o default -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-cfg5 > $D.log 2>&1)
f=$(find $D -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r05_default_kernel_stats.csv; grep "^{" $D.log | tail -1 > $OUT/r05_default_bench_under_rocprof.json
head -40 $OUT/r05_default_kernel_stats.csv | cut -c1-220
rm -rf $OUT/prof_default
