#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3f; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python scripts/ab_prefill.py --shapes 4096x4096 --ms 512 4096 --variants mfma_fused4 --fused-env "BM=64,SPLIT=1" "BM=64,SPLIT=1,ABLATE=8" "BM=64,SPLIT=1,ABLATE=16" "BM=64,SPLIT=1,ABLATE=24" "BM=128,SPLIT=1" "BM=128,SPLIT=1,ABLATE=8" "BM=128,SPLIT=1,ABLATE=16" "BM=128,SPLIT=1,ABLATE=24" > $O/ablate2.jsonl 2> $O/ablate2.err; cat $O/ablate2.jsonl; tail -2 $O/ablate2.err
