#!/bin/bash
# round 4, visit 15+: implicit-GEMM convolution (int8 / fp8 / int4) - parity, then time against im2col + GEMM and dequantize + float convolution
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8
timeout 300 python scripts/time_conv2d.py qint8 2>&1 | grep "^{" | tee $OUT/r04_qconv2d_implicit_gemm_v2.jsonl
timeout 300 python scripts/time_conv2d.py qint4 2>&1 | grep "^{" | tee $OUT/r04_qconv2d_int4_implicit_gemm_v2.jsonl
