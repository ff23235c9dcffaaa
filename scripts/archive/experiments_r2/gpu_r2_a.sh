#!/bin/bash
# round 2, GPU call A: parity suite, default bench line, GEMV load-policy A/B, kernel trace of the default bench
set -u
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.log
( timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) ; tail -c 400 $O/bench_default.err
( timeout 300 python scripts/ab.py --workloads northstar cfg3 qkv_fused gateup_fused --env QUANTO_HIP_GEMV_VARIANT=0,1,2,3,4,6 --rounds 7 > $O/ab_gemv.jsonl 2> $O/ab_gemv.err )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_default -o default -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err )
find $O/prof_default -name "*kernel_stats*" | head -3
cat $O/pytest_gpu.log | tail -5
cat $O/ab_gemv.jsonl
