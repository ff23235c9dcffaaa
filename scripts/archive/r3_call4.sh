#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3d; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "fused4" -p no:cacheprovider > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
timeout 900 python scripts/ab_prefill.py --shapes 4096x4096 --ms 128 512 1024 --variants mfma_fused4 dequant_mfma --fused-env "" "BM=64,SPLIT=1" "BM=64,SPLIT=1,ABLATE=1" "BM=64,SPLIT=1,ABLATE=3" "BM=64,SPLIT=1,ABLATE=7" "BM=64,SPLIT=1,ABLATE=4" "BM=64,SPLIT=1,ABLATE=2" "BM=128,SPLIT=1" "BM=128,SPLIT=1,ABLATE=1" "BM=128,SPLIT=1,ABLATE=7" > $O/ablate.jsonl 2> $O/ablate.err; cat $O/ablate.jsonl; tail -2 $O/ablate.err
