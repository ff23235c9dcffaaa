#!/bin/bash
# two wave sets per 64-feature block (8 waves): parity + A/B against the four-wave block
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2x; O=gpurun_out/r2x
timeout 900 python -m pytest tests/test_multi_linear.py tests/test_hip_parity.py tests/test_backward_and_workspace.py -x -q -m gpu -k "qbits and (multi or skinny or batched or workspace)" > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
timeout 600 python scripts/ab.py --workloads int4_decode32 int4_decode64 int4_decode32_up int4_decode32_down qkv_fused8 qkv_fused32 gateup_fused8 gateup_fused32 --env QUANTO_HIP_SKINNY_SETS=1,2 --rounds 5 > $O/ab.txt 2>&1
grep -o '"workload": "[a-z0-9_]*", "QUANTO_HIP_SKINNY_SETS": "[0-9]*", "kernel": "[a-z_0-9]*"\|"us_median": [0-9.]*' $O/ab.txt | paste - -
