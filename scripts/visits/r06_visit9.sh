set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/v9; mkdir -p $OUT
export TMPDIR=/tmp QUANTO_HIP_EXPERIMENT=1
timeout 900 python -m pytest tests/test_qconv2d.py -q -x -m gpu -k "depthwise or other_groupings" -p no:cacheprovider --timeout 300 > $OUT/dw.log 2>&1; echo "dw exit=$?"; tail -8 $OUT/dw.log
timeout 300 python scripts/time_depthwise.py 2>&1 | grep -v amdgpu.ids | tee $OUT/depthwise_timing.jsonl
timeout 600 python -m pytest tests/test_w4a8_gpu.py tests/test_reference_style_activations.py -q -x -p no:cacheprovider --timeout 300 > $OUT/w4a8.log 2>&1; echo "w4a8 exit=$?"; tail -3 $OUT/w4a8.log
