#!/bin/bash
# round 3, GPU visit 3: fused int4 GEMM v2 (64-token tiles, XS once per token half, 7-op operand construction): parity, then where it wins
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3c; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "fused4 or large_tile_configurations or four_wave or prefill" -p no:cacheprovider > $O/parity.log 2>&1; echo "parity rc=$?"; tail -4 $O/parity.log
timeout 900 python scripts/ab_prefill.py --shapes 4096x4096 --ms 96 128 192 256 384 512 768 1024 2048 > $O/ab_prefill_4096.jsonl 2> $O/ab_prefill.err; cat $O/ab_prefill_4096.jsonl; tail -2 $O/ab_prefill.err
timeout 900 python scripts/ab_prefill.py --shapes 14336x4096 4096x14336 1024x4096 --ms 128 256 512 1024 > $O/ab_prefill_llama.jsonl 2>> $O/ab_prefill.err; cat $O/ab_prefill_llama.jsonl
