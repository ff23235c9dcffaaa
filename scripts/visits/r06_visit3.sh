set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/v3; mkdir -p $OUT
export TMPDIR=/tmp QUANTO_HIP_EXPERIMENT=1
timeout 120 python scripts/probes/w4a8_fp8_diag.py > $OUT/w4a8_diag.log 2>&1; echo "diag exit=$?"; cat $OUT/w4a8_diag.log | tail -25
timeout 900 python -m pytest tests/test_native8_split_gpu.py -q -x -p no:cacheprovider --timeout 300 > $OUT/split.log 2>&1; echo "split exit=$?"; tail -15 $OUT/split.log
for small in 0 1; do
  export QUANTO_HIP_NATIVE8_SMALL=$small
  timeout 600 python scripts/ab.py --workloads cfg4_fp8a8 --env QUANTO_HIP_NATIVE8_SPLIT=1,2,4,8 --rounds 5 > $OUT/ab_cfg4_fp8a8_small$small.jsonl 2>&1; tail -4 $OUT/ab_cfg4_fp8a8_small$small.jsonl
done
unset QUANTO_HIP_NATIVE8_SMALL
timeout 600 python scripts/microbench_qbytes.py --graph --pairs i8:i8 f8:f8 --shapes 512x8192x8192 512x4096x4096 1024x4096x4096 256x4096x4096 128x4096x4096 512x4096x14336 512x14336x4096 2048x4096x4096 1024x8192x8192 768x8192x4096 > $OUT/auto_shapes.jsonl 2>&1; cat $OUT/auto_shapes.jsonl
QUANTO_HIP_NATIVE8_SPLIT=1 timeout 600 python scripts/microbench_qbytes.py --graph --pairs i8:i8 f8:f8 --shapes 512x8192x8192 512x4096x4096 1024x4096x4096 256x4096x4096 128x4096x4096 512x4096x14336 512x14336x4096 2048x4096x4096 1024x8192x8192 768x8192x4096 > $OUT/nosplit_shapes.jsonl 2>&1; cat $OUT/nosplit_shapes.jsonl
