#!/bin/bash
# round 5, last visit: the convolution GPU file on the final tree (sub-byte dense route from 8 pixel tiles on)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5final7; mkdir -p $OUT; export TMPDIR=/tmp
timeout 58 python -m pytest tests/test_qconv2d.py -m gpu -q -p no:cacheprovider --maxfail 8 --timeout 50 2>&1 | tail -12 | tee $OUT/r05_conv_gpu_tests_final_tail.txt
