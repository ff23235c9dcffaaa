#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3h; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "fused4 or prefill" -p no:cacheprovider > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
timeout 1200 python scripts/ab_prefill.py --shapes 4096x4096 1024x4096 4096x14336 --ms 72 128 192 256 384 --variants mfma_fused4 skinny --fused-env "BM=64,SPLIT=1" "BM=64,SPLIT=2" "BM=64,SPLIT=4" "BM=64,SPLIT=8" > $O/split.jsonl 2> $O/split.err; cat $O/split.jsonl; tail -2 $O/split.err
