set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/v2; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_w4a8_gpu.py -q -x -p no:cacheprovider --timeout 300 > $OUT/w4a8.log 2>&1; echo "w4a8 exit=$?"; tail -30 $OUT/w4a8.log
timeout 300 python -m pytest tests/test_qconv2d.py -q -m gpu -k "fused_gemm_gpu and fp32" -p no:cacheprovider > $OUT/conv_fp32.log 2>&1; echo "conv exit=$?"; tail -3 $OUT/conv_fp32.log
for w in w4a8 w4a8_512 w4afp8 w4afp8_512; do
  timeout 300 python bench.py --workload $w --no-sub --no-cpu-baseline --no-profile --steps 50 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "bench $w exit=$?"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$w.json")); print("$w", d["ms_per_step"]*1e3, "us", d["roofline"]["frac"], d["roofline"]["kernel"], "ref_rocm_us", d.get("ref_rocm_us"))
except Exception as e: print("$w", "failed", e); print(open("$OUT/bench_$w.err").read()[-1500:])
PY
done
