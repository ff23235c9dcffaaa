#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2q; O=gpurun_out/r2q
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "mmv or skinny" > $O/pytest.log 2>&1; tail -n 8 $O/pytest.log
for mm in 0 32; do
echo "== MMV_MAX_M=$mm"
QUANTO_HIP_MMV_MAX_M=$mm timeout 600 python scripts/ab.py --workloads int4_decode8 int4_decode16 int4_decode8_up int4_decode8_down int4_decode16_up int4_decode16_down --env QUANTO_HIP_MMV_DEEP=0 --rounds 5 > $O/ab$mm.txt 2>&1
grep -o '"workload": "[a-z0-9_]*", "QUANTO_HIP_MMV_DEEP": "[0-9]*", "kernel": "[a-z_0-9]*"\|"us_median": [0-9.]*' $O/ab$mm.txt | paste - -
done
