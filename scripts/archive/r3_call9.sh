#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3g; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "fused4 or prefill" -p no:cacheprovider > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
timeout 1200 python scripts/ab_prefill.py --shapes 4096x4096 --ms 72 96 128 160 192 256 320 384 512 640 768 1024 1536 2048 --variants mfma_fused4 dequant_mfma skinny --fused-env "" "BM=64,SPLIT=1" "BM=64,SPLIT=2" "BM=64,SPLIT=4" "BM=128,SPLIT=1" > $O/sweep_4096.jsonl 2> $O/sweep.err; cat $O/sweep_4096.jsonl
timeout 1200 python scripts/ab_prefill.py --shapes 14336x4096 4096x14336 1024x4096 --ms 96 128 256 512 1024 --variants mfma_fused4 dequant_mfma skinny --fused-env "" "BM=64,SPLIT=1" "BM=64,SPLIT=2" "BM=64,SPLIT=4" "BM=128,SPLIT=1" "BM=128,SPLIT=2" > $O/sweep_llama.jsonl 2>> $O/sweep.err; cat $O/sweep_llama.jsonl; tail -2 $O/sweep.err
