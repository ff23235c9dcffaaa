#!/bin/bash
# round 2, GPU call B: 32x32x16 large-tile kernel - parity, then A/B against the 16x16x32 configurations
set -u
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
( timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "large_tile_configurations or 32x32x16 or cfg2_bf16 or cfg4 or w8a8 or fp8a8 or int4_prefill" 2>&1 | tail -15 ) > $O/pytest_large.log
tail -6 $O/pytest_large.log
( timeout 400 python scripts/ab.py --workloads cfg2 fp8_4k int8_8k --env QUANTO_HIP_LARGE_CFG=0,4,3,1 --rounds 7 > $O/ab_large.jsonl 2> $O/ab_large.err )
cat $O/ab_large.jsonl; tail -3 $O/ab_large.err
