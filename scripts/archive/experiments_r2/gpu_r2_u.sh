#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2u; O=gpurun_out/r2u
timeout 600 python scripts/ab.py --workloads gateup_fused8 gateup_fused32 int4_decode32_up --env QUANTO_HIP_SKINNY_SPLIT=1,2,4 --rounds 5 > $O/ab.txt 2>&1
grep -o '"workload": "[a-z0-9_]*", "QUANTO_HIP_SKINNY_SPLIT": "[0-9]*", "kernel": "[a-z_0-9]*"\|"us_median": [0-9.]*' $O/ab.txt | paste - -
