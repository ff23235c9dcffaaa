set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/v11; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --timeout 300 > $OUT/pytest_gpu.log 2>&1 ) 2> $OUT/pytest_time.txt; echo "pytest exit=$?"; tail -12 $OUT/pytest_gpu.log; cat $OUT/pytest_time.txt
( time timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt; echo "bench exit=$?"; wc -c $OUT/bench_default.json; tail -3 $OUT/bench_default.err; cat $OUT/bench_time.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/v11/bench_default.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for s in d.get("sub_results", []):
    print(s.get("name"), s.get("us_per_step"), s.get("kernel"), s.get("frac"), s.get("ref_rocm_us"), {k: v for k, v in s.items() if k.endswith("tok_s") or k in ("us_per_layer", "int8_us", "int4_us", "dw_us", "ref_rocm_dw_us", "dw_frac_hbm")})
PY
