#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2r; O=gpurun_out/r2r
timeout 600 python scripts/ab.py --workloads int4_decode8_kv int4_decode8_13b int4_decode8_70b int4_decode8 --env QUANTO_HIP_MMV_MAX_M=0,16 --rounds 5 > $O/ab.txt 2>&1
grep -o '"workload": "[a-z0-9_]*", "QUANTO_HIP_MMV_MAX_M": "[0-9]*", "kernel": "[a-z_0-9]*"\|"us_median": [0-9.]*' $O/ab.txt | paste - -
QUANTO_HIP_MMV_FG=2 timeout 600 python scripts/ab.py --workloads int4_decode8_13b int4_decode8_70b int4_decode8 --env QUANTO_HIP_MMV_MAX_M=16 --rounds 5 > $O/ab2.txt 2>&1
echo "FG=2"; grep -o '"workload": "[a-z0-9_]*", "QUANTO_HIP_MMV_MAX_M": "[0-9]*", "kernel": "[a-z_0-9]*"\|"us_median": [0-9.]*' $O/ab2.txt | paste - -
