#!/bin/bash
# padded scale/shift table rows (bank conflicts in the fill): parity + timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2x; O=gpurun_out/r2x
timeout 900 python -m pytest tests/test_multi_linear.py tests/test_hip_parity.py tests/test_backward_and_workspace.py -x -q -m gpu -k "qbits and (multi or skinny or batched or mmv or workspace)" > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log
timeout 600 python scripts/ab.py --workloads int4_decode8 int4_decode32 int4_decode64 int4_decode32_up int4_decode32_down qkv_fused8 qkv_fused32 gateup_fused8 gateup_fused32 --env QUANTO_HIP_SKINNY_NT=1 --rounds 5 > $O/ab.txt 2>&1
grep -o '"workload": "[a-z0-9_]*", "QUANTO_HIP_SKINNY_NT": "[0-9]*", "kernel": "[a-z_0-9]*"\|"us_median": [0-9.]*' $O/ab.txt | paste - -
bash scripts/pmc.sh gateup_fused32 sq2b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE 2>&1 | tail -n 1 | cut -c1-400
