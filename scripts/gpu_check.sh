#!/bin/bash
# One GPU-box visit: smoke, parity tests, quick bench of each workload, rocprofv3 kernel stats.
# Usage (from the build container): gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh [tests|bench|prof ...]'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
STAGES="${*:-smoke tests bench prof}"
for s in $STAGES; do case $s in
smoke)
  timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit=$?" | tee -a $OUT/summary.txt; tail -5 $OUT/smoke.log;;
tests)
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --timeout 300 > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit=$?" | tee -a $OUT/summary.txt; tail -40 $OUT/pytest_gpu.log;;
bench)
  for w in cfg2 cfg3 northstar cfg4; do
    timeout 600 python bench.py --workload $w --steps 50 --warmup 10 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
    echo "bench $w exit=$?"; cat $OUT/bench_$w.json; tail -3 $OUT/bench_$w.err
  done;;
prof)
  for w in cfg2 cfg3; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_$w -o $w -- python $OLDPWD/bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline > $OLDPWD/$OUT/prof_$w.log 2>&1)
    echo "prof $w exit=$?"
    find $OUT/prof_$w -name "*kernel_stats*" | head -2 | while read f; do echo "== $f"; head -12 "$f"; done
  done;;
esac; done
