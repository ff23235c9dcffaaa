set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/s3final; mkdir -p $OUT
export TMPDIR=/tmp
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt
tail -c 600 $OUT/bench_default.json; echo; cat $OUT/bench_time.txt
PMC_FOR="" bash scripts/collect_profiles.sh r06s3 default > $OUT/collect.log 2>&1
cp gpurun_out/profiles_r06s3/default_kernel_stats.csv $OUT/ 2>/dev/null; head -12 $OUT/default_kernel_stats.csv | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( time timeout 1400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 --durations=0 > $OUT/pytest_gpu.log 2>&1 ) 2> $OUT/pytest_time.txt
tail -3 $OUT/pytest_gpu.log; cat $OUT/pytest_time.txt
