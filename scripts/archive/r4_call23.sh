#!/bin/bash
# round 4, last visit: the GPU suite and smoke() on the final tree
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 480 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x 2>&1 | tail -3 | tee $OUT/r04_gpu_tests_tail.txt
timeout 60 python __graft_entry__.py smoke 2>&1 | tail -2
