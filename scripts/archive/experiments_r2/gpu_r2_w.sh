#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2w; O=gpurun_out/r2w
timeout 1200 python -m pytest tests/test_multi_linear.py tests/test_hip_parity.py -x -q -m gpu -k "multi or qbytes or fused" -n 4 > $O/pytest.log 2>&1; tail -n 6 $O/pytest.log
