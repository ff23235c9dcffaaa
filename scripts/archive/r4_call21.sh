#!/bin/bash
# round 4, visit 21: AUTO against every forced kernel off the fitted grid, final dispatch
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 330 python scripts/auto_vs_best.py --out $OUT/r04_auto_vs_best.jsonl > /dev/null 2> $OUT/r04_auto_vs_best.err; echo "exit=$?"; tail -2 $OUT/r04_auto_vs_best.err; wc -l $OUT/r04_auto_vs_best.jsonl
