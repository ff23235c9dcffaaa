#!/bin/bash
# round 5, visit 3: convolution kernel with ragged K / wide windows / qint2 (parity + timing), bench line with the fixed steady-state trace and
# the batched-decode ablation, host cost with resolved op overloads
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c3; mkdir -p $OUT; export TMPDIR=/tmp
echo "== conv parity"
timeout 900 python -m pytest tests/test_qconv2d.py tests/test_reference_style_gpu.py -m gpu -q -p no:cacheprovider -x --timeout 300 2>&1 | tail -6 | tee $OUT/conv_parity_tail.txt
echo "== native8 / plan cache parity"
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x --timeout 300 \
  -k "native8 or int8_int8 or fp8_fp8 or w8a8 or fp8a8 or int4_prefill or dense_gemm or plan_cache" 2>&1 | tail -3 | tee $OUT/native8_parity_tail.txt
echo "== conv timing"
for W in qint8 qint4; do timeout 300 python scripts/time_conv2d.py $W stems 2>&1 | grep -v Warning | tee -a $OUT/conv_stems.jsonl; done
timeout 300 python scripts/time_conv2d.py qint8 2>&1 | grep -v Warning | tee $OUT/conv_default_shapes.jsonl
echo "== host overhead"
timeout 120 python scripts/host_overhead.py 2>&1 | grep -v Warning | tee $OUT/host_overhead_after.jsonl
echo "== bench line"
QH_BENCH_KEEP_TRACE=$OUT/bench_kernel_trace.csv timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cfg5 > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 800 $OUT/bench_err.txt; wc -c $OUT/bench_line.json
python - "$OUT/bench_line.json" <<'PY'
import json, sys
try:
    p = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("headline", p["value"], p["ms_per_step"], p["roofline"]["frac"], "kernel_us", p["roofline"].get("kernel_us"), p["roofline"].get("kernel_us_min"), "event", p["roofline"]["event_us"], "traffic", p["roofline"].get("traffic"), p.get("profile_passes", {}).get("seconds"))
    for sr in p["sub_results"]:
        print({k: v for k, v in sr.items() if k in ("name", "us_per_step", "us_per_layer", "event_us", "kernel_us", "kernel_us_min", "frac", "traffic", "alg_bytes", "ablate_us", "ref_rocm_us")})
except Exception as e:
    print("bench line unreadable:", e)
PY
