#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3l; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_multi_linear.py -q -m gpu -p no:cacheprovider -k "(qbits_gemv and bf16) or eight_rows or cfg3 or bit_identical or zero_point_and_fallbacks or fused_decode_projections_on_device" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 600 python scripts/ab.py --workloads northstar cfg3 qkv_fused gateup_fused --env QUANTO_HIP_GEMV_RR=4,8 --rounds 7 > $O/ab_rr.jsonl 2> $O/ab_rr.err; cat $O/ab_rr.jsonl; tail -2 $O/ab_rr.err
