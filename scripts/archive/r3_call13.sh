#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3j; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_causal_lm.py -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? bytes=$(wc -c < $O/bench_default.json)"; tail -2 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3j/bench_default.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"]["launch_us"])
for s in d["sub_results"]: print(s["name"], s["launch_us"], s["frac"], s["kernel"])
PY
bash scripts/collect_profiles.sh r03 default int4_prefill512 > $O/collect.log 2>&1; tail -5 $O/collect.log
