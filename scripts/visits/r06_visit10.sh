set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/v10; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt; echo "bench exit=$?"; wc -c $OUT/bench_default.json; tail -3 $OUT/bench_default.err; cat $OUT/bench_time.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/v10/bench_default.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for s in d.get("sub_results", []):
    print(s.get("name"), s.get("us_per_step"), s.get("kernel"), s.get("frac"), s.get("ref_rocm_us"), {k: v for k, v in s.items() if k.endswith("tok_s") or k in ("us_per_layer", "int8_us", "int4_us")})
PY
