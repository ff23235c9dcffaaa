set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/v5; mkdir -p $OUT
export TMPDIR=/tmp QUANTO_HIP_EXPERIMENT=1
timeout 900 python -m pytest tests/test_native8_split_gpu.py -q -p no:cacheprovider --timeout 300 > $OUT/split.log 2>&1; echo "split exit=$?"; tail -8 $OUT/split.log
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_activations.py tests/test_reference_style_activations.py -q -k "native8 or w8a8 or fp8a8 or dense_gemm or activation or qbytes" -p no:cacheprovider --timeout 300 > $OUT/native8.log 2>&1; echo "native8 exit=$?"; tail -5 $OUT/native8.log
SHAPES="512x8192x8192 512x4096x4096 1024x4096x4096 256x4096x4096 128x4096x4096 64x4096x4096 512x4096x14336 512x14336x4096 2048x4096x4096 1024x8192x8192 768x8192x4096 256x8192x8192 384x4096x4096 4096x4096x4096 32x4096x14336"
timeout 600 python scripts/microbench_qbytes.py --graph --pairs i8:i8 f8:f8 --shapes $SHAPES > $OUT/auto_shapes.jsonl 2>&1; cat $OUT/auto_shapes.jsonl
QUANTO_HIP_NATIVE8_SPLIT=1 timeout 600 python scripts/microbench_qbytes.py --graph --pairs i8:i8 f8:f8 --shapes $SHAPES > $OUT/nosplit_shapes.jsonl 2>&1; cat $OUT/nosplit_shapes.jsonl
timeout 300 python scripts/stress_splitk.py > $OUT/stress.log 2>&1; tail -4 $OUT/stress.log
