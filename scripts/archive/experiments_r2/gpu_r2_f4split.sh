#!/bin/bash
# split-K in the fused int4 GEMM: parity + timing against the unsplit kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2f4; O=gpurun_out/r2f4
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_backward_and_workspace.py -x -q -m gpu -k "fused4 or workspace" > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log
timeout 300 python scripts/ab.py --workloads int4_prefill512 --env QUANTO_HIP_FUSED4_SPLIT=1,2,4 --rounds 5 --steps 50 > $O/ab.txt 2>&1
grep -o '"workload": "[a-z0-9_]*", "QUANTO_HIP_FUSED4_SPLIT": "[0-9]*", "kernel": "[a-z_0-9]*"\|"us_median": [0-9.]*' $O/ab.txt | paste - -
timeout 400 python scripts/ab_prefill.py --ms 256 512 1024 2>&1 | tail -n 12 | cut -c1-200
