#!/bin/bash
# two wave sets in the 8-bit streaming kernel: parity + A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2x; O=gpurun_out/r2x
timeout 900 python -m pytest tests/test_multi_linear.py tests/test_hip_parity.py tests/test_backward_and_workspace.py -x -q -m gpu -k "qbytes and (multi or skinny or batched or workspace)" -n 4 > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
timeout 600 python scripts/ab.py --workloads int8_decode32 int8_qkv_fused32 int8_gateup_fused32 --env QUANTO_HIP_SKINNY_SETS=1,2 --rounds 5 > $O/ab.txt 2>&1
grep -o '"workload": "[a-z0-9_]*", "QUANTO_HIP_SKINNY_SETS": "[0-9]*", "kernel": "[a-z_0-9]*"\|"us_median": [0-9.]*' $O/ab.txt | paste - -
