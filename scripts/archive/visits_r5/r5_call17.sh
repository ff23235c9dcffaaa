#!/bin/bash
# round 5, visit 17: AUTO against every forced kernel with the round's kernels (the dense kernel got faster: does the fused / dequantize + dense hand-over still sit right?)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c17; mkdir -p $OUT; export TMPDIR=/tmp
export QUANTO_HIP_EXPERIMENT=1
timeout 600 python scripts/auto_vs_best.py --out $OUT/auto_vs_best.jsonl 2>&1 | grep "^{" | cut -c1-100 | tail -3
timeout 200 python -m pytest tests/test_dispatch_auto_gpu.py tests/test_dispatch_fuzz_gpu.py -q -p no:cacheprovider -m "gpu or perf" 2>&1 | tail -3 | tee $OUT/dispatch_tests_tail.txt
