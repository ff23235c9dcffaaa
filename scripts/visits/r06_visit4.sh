set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/v4; mkdir -p $OUT
export TMPDIR=/tmp QUANTO_HIP_EXPERIMENT=1
timeout 120 python scripts/probes/w4a8_fp8_diag.py > $OUT/w4a8_diag.log 2>&1; echo "diag exit=$?"; tail -22 $OUT/w4a8_diag.log
timeout 900 python -m pytest tests/test_w4a8_gpu.py -q -p no:cacheprovider --timeout 300 > $OUT/w4a8.log 2>&1; echo "w4a8 exit=$?"; tail -5 $OUT/w4a8.log
timeout 900 python -m pytest tests/test_native8_split_gpu.py -q -x -p no:cacheprovider --timeout 300 > $OUT/split.log 2>&1; echo "split exit=$?"; tail -15 $OUT/split.log
timeout 600 python -m pytest tests/test_hip_parity.py -q -x -k "native8 or w8a8 or fp8a8 or dense_gemm" -p no:cacheprovider --timeout 300 > $OUT/native8.log 2>&1; echo "native8 exit=$?"; tail -5 $OUT/native8.log
for small in 0 1; do
  export QUANTO_HIP_NATIVE8_SMALL=$small
  timeout 600 python scripts/ab.py --workloads cfg4_fp8a8 --env QUANTO_HIP_NATIVE8_SPLIT=1,2,4,8 --rounds 5 > $OUT/ab_cfg4_fp8a8_small$small.jsonl 2>&1; tail -4 $OUT/ab_cfg4_fp8a8_small$small.jsonl
done
unset QUANTO_HIP_NATIVE8_SMALL
SHAPES="512x8192x8192 512x4096x4096 1024x4096x4096 256x4096x4096 128x4096x4096 512x4096x14336 512x14336x4096 2048x4096x4096 1024x8192x8192 768x8192x4096 256x8192x8192 384x4096x4096"
timeout 600 python scripts/microbench_qbytes.py --graph --pairs i8:i8 f8:f8 --shapes $SHAPES > $OUT/auto_shapes.jsonl 2>&1; cat $OUT/auto_shapes.jsonl
for sm in 0 1; do for sp in 2 4 8; do
QUANTO_HIP_NATIVE8_SMALL=$sm QUANTO_HIP_NATIVE8_SPLIT=$sp timeout 600 python scripts/microbench_qbytes.py --graph --pairs i8:i8 f8:f8 --shapes $SHAPES 2>&1 | grep '^{' | sed "s/^{/{\"small\": $sm, \"split\": $sp, /" >> $OUT/forced_shapes.jsonl
done; done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/v4/forced_shapes.jsonl")]
auto=[json.loads(l) for l in open("gpurun_out/v4/auto_shapes.jsonl") if l.startswith("{")]
best={}
for r in rows:
    k=(r["M"],r["N"],r["K"],r["a"])
    if k not in best or r["us"]<best[k]["us"]: best[k]=r
for a in auto:
    k=(a["M"],a["N"],a["K"],a["a"]); b=best.get(k)
    print(k, "auto", a["us"], "best forced", b and (b["us"], b["small"], b["split"]))
PY
