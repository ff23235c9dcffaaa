#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3q; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python scripts/stress_splitk.py > $O/stress.log 2>&1; echo "stress rc=$?"; tail -3 $O/stress.log
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_multi_linear.py tests/test_backward_and_workspace.py -q -m gpu -p no:cacheprovider -x -k "skinny or multi or split or decode or fused4 or large or graph or stream" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 900 python scripts/ab_prefill.py --shapes 4096x4096 4096x14336 --ms 128 256 512 1024 --variants mfma_fused4 --fused-env "" "BM=64,SPLIT=1" "BM=64,SPLIT=2" "BM=64,SPLIT=4" "BM=64,SPLIT=8" "BM=128,SPLIT=2" > $O/split.jsonl 2> $O/split.err; cat $O/split.jsonl; tail -2 $O/split.err
timeout 600 python scripts/ab.py --workloads cfg4 --env QUANTO_HIP_LARGE_SPLIT=1,2 --rounds 5 > $O/ab_large.jsonl 2> $O/ab_large.err; cat $O/ab_large.jsonl; tail -2 $O/ab_large.err
timeout 600 python scripts/ab.py --workloads int4_decode32 qkv_fused32 int4_decode64 int4_decode32_down int4_decode32_up int8_decode32 int8_qkv_fused32 --env QUANTO_HIP_DUMMY=0 --rounds 5 > $O/ab.jsonl 2> $O/ab.err; cat $O/ab.jsonl; tail -2 $O/ab.err
