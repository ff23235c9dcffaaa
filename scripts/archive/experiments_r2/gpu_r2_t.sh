#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2t; O=gpurun_out/r2t
timeout 900 python -m pytest tests/test_multi_linear.py tests/test_hip_parity.py -x -q -m gpu -k "multi or skinny or batched or fused_decode" > $O/pytest.log 2>&1; tail -n 6 $O/pytest.log
timeout 600 python scripts/ab.py --workloads qkv_fused8 qkv_fused32 gateup_fused8 gateup_fused32 --env QUANTO_HIP_SKINNY_MULTI_MAX_M=0,64 --rounds 5 > $O/ab.txt 2>&1
grep -o '"workload": "[a-z0-9_]*", "QUANTO_HIP_SKINNY_MULTI_MAX_M": "[0-9]*", "kernel": "[a-z_0-9]*"\|"us_median": [0-9.]*' $O/ab.txt | paste - -
tail -n 3 $O/ab.txt | cut -c1-300
