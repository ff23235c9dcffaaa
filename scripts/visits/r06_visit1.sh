set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/v1; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fp32_gpu.py -q -x -p no:cacheprovider --timeout 300 > $OUT/fp32.log 2>&1; echo "fp32 exit=$?"; tail -15 $OUT/fp32.log
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --timeout 300 > $OUT/pytest_gpu.log 2>&1 ) 2> $OUT/pytest_time.txt; echo "pytest exit=$?"; tail -25 $OUT/pytest_gpu.log; cat $OUT/pytest_time.txt
( time timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt; echo "bench exit=$?"; tail -c 3000 $OUT/bench_default.json; tail -5 $OUT/bench_default.err; cat $OUT/bench_time.txt
G4_TIMEOUT=1200 bash scripts/run_reference_tests_gpu.sh
