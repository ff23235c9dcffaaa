#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3y; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -p no:cacheprovider -x -k "fused4" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout 900 python scripts/ab_prefill.py --shapes 4096x4096 --ms 128 256 512 1024 1536 --variants mfma_fused4 --fused-env "" "BM=64,SPLIT=1" "BM=128,SPLIT=1" > $O/pre.jsonl 2> $O/pre.err; cat $O/pre.jsonl; tail -2 $O/pre.err
