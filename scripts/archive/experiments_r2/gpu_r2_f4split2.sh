#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2f4; O=gpurun_out/r2f4
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_backward_and_workspace.py tests/test_dispatch_fuzz_gpu.py tests/test_causal_lm.py tests/test_reference_style_gpu.py -x -q -m gpu -k "fused4 or workspace or fuzz or causal or weight_qbits or prefill or dequant" -n 4 > $O/pytest2.log 2>&1; tail -n 4 $O/pytest2.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
