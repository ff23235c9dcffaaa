#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2v; O=gpurun_out/r2v
timeout 900 python -m pytest tests/test_multi_linear.py tests/test_hip_parity.py -x -q -m gpu -k "multi or skinny or batched or mmv" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
timeout 600 python scripts/ab.py --workloads int4_decode8 int4_decode32 int4_decode64 int4_decode32_up int4_decode32_down qkv_fused32 gateup_fused32 int8_decode32 --env QUANTO_HIP_SKINNY_NT=1 --rounds 5 > $O/ab.txt 2>&1
grep -o '"workload": "[a-z0-9_]*", "QUANTO_HIP_SKINNY_NT": "[0-9]*", "kernel": "[a-z_0-9]*"\|"us_median": [0-9.]*' $O/ab.txt | paste - -
