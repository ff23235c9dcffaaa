#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3t; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_multi_linear.py -q -m gpu -p no:cacheprovider -x -k "skinny or multi or fused4 or prefill or qbytes" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 600 python scripts/ab.py --workloads int4_decode32 qkv_fused32 int4_decode64 gateup_fused32 int8_decode32 int8_gateup_fused32 int4_prefill512 --env QUANTO_HIP_DUMMY=0 --rounds 5 > $O/ab.jsonl 2> $O/ab.err; cut -c1-150 $O/ab.jsonl; tail -2 $O/ab.err
timeout 600 python scripts/ab_prefill.py --shapes 4096x4096 --ms 128 256 512 1024 --variants mfma_fused4 > $O/pre.jsonl 2>$O/pre.err; cat $O/pre.jsonl
