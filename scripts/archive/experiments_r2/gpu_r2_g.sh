#!/bin/bash
# round 2, GPU call G: fused int4 GEMM - shape ladder in subprocesses, parity, then the kernel-vs-M map
set -u
export TMPDIR=/tmp
O=gpurun_out/r2g; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
python scripts/debug_fused.py > $O/ladder.txt 2>&1; cat $O/ladder.txt
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_backward_and_workspace.py -m gpu -x -q -k "fused4 or gemv or misaligned or naive_any" 2>&1 | tail -30 ) > $O/pytest.log
grep -v "^  File\|^$" $O/pytest.log | tail -8
timeout 900 python scripts/ab_prefill.py > $O/ab_prefill.jsonl 2> $O/err.txt
cat $O/ab_prefill.jsonl; tail -3 $O/err.txt
