#!/bin/bash
# round 5, visit 12: native8 on a ring of five operand parts (parity, A/B against the two-buffer form, K scaling)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c12; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider --maxfail 10 --timeout 300 -k "native8 or dense_gemm or w8a8 or int4_prefill or int8_int8 or e4m3fnuz" 2>&1 | tail -6 | tee $OUT/parity_tail.txt
export QUANTO_HIP_EXPERIMENT=1
timeout 300 python scripts/ab.py --rounds 7 --sequential --workloads w8a8 cfg4_w8a8 int4_prefill int4_prefill512 --env QUANTO_HIP_NATIVE8_RING=0,1 2>&1 | grep "^{" | tee $OUT/ring_ab.jsonl
timeout 300 python scripts/native8_k_scaling.py 2>&1 | grep "^{" | grep -v "hipBLASLt" | tee $OUT/native8_k_scaling.jsonl
