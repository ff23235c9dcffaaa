#!/bin/bash
# round 4, closing visit: full GPU suite, smoke, the reference's own tests in plug-in mode (G4), `python bench.py` exactly as the driver issues it,
# and the rocprofv3 kernel statistics of that same command.  Outputs under gpurun_out/ (copied to profiles/r04_* afterwards).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== gpu tests"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | tail -4 | tee $OUT/r04_gpu_tests_tail.txt
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
echo "== G4"; G4_TIMEOUT=900 bash scripts/run_reference_tests_gpu.sh 2>&1 | tail -6; cp $OUT/g4/reference_tests_plugin_mode.log $OUT/r04_g4_reference_tests_plugin_mode.log 2>/dev/null
echo "== bench (driver command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_default_bench_line.json 2> $OUT/r04_default_bench.err; echo "exit=$? bytes=$(wc -c < $OUT/r04_default_bench_line.json)"
echo "== kernel stats of the same command"
D=$OUT/prof_r04_default; rm -rf $D
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o default -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-cfg5 > $D.log 2>&1)
f=$(find $D -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r04_default_kernel_stats.csv; grep "^{" $D.log | tail -1 > $OUT/r04_default_bench_under_rocprof.json
head -30 $OUT/r04_default_kernel_stats.csv | cut -c1-200
for w in int4_prefill; do
  D=$OUT/prof_r04_$w; rm -rf $D
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o $w -- python $REPO/bench.py --workload $w --no-sub --no-cpu-baseline --steps 20 --warmup 5 > $D.log 2>&1)
  f=$(find $D -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r04_${w}_kernel_stats.csv; grep "^{" $D.log | tail -1 > $OUT/r04_${w}_bench_under_rocprof.json
  head -6 $OUT/r04_${w}_kernel_stats.csv | cut -c1-200
done
echo "== kernel stats: group-size-96 streaming kernel and the int4 implicit-GEMM convolution"
D=$OUT/prof_r04_new; rm -rf $D
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o new -- python $REPO/scripts/profile_new_kernels.py > $D.log 2>&1)
f=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "^\"Name|qbits_skinny_kernel|qconv2d_" "$f" | cut -c1-300 > $OUT/r04_group96_and_int4_conv_kernel_stats.csv; cut -c1-200 $OUT/r04_group96_and_int4_conv_kernel_stats.csv
rm -rf $OUT/prof_r04_* $OUT/g4
