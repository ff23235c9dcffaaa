#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3m; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python scripts/stress_splitk.py > $O/stress.log 2>&1; echo "stress rc=$?"; tail -5 $O/stress.log
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_multi_linear.py tests/test_backward_and_workspace.py -q -m gpu -p no:cacheprovider -x -k "skinny or multi or split or decode or fuzz or graph or stream" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 600 python scripts/ab.py --workloads int4_decode32 qkv_fused32 int4_decode64 int4_decode32_down int4_decode32_up int4_decode8_kv --env QUANTO_HIP_DUMMY=0 --rounds 7 > $O/ab.jsonl 2> $O/ab.err; cat $O/ab.jsonl; tail -2 $O/ab.err
