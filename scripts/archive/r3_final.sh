#!/bin/bash
# closing run of round 3: full GPU suite, smoke, driver-style bench, profile refresh (kernel stats of the default run + HBM counters)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3final; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; tail -n 3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/rc.txt
PMC_FOR="cfg2 northstar gateup_fused int4_decode32 qkv_fused32 int4_prefill512" timeout 1200 bash scripts/collect_profiles.sh r03 default cfg2 northstar gateup_fused int4_decode32 qkv_fused32 int4_prefill512 > $O/collect.log 2>&1; echo "collect rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3final/bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["roofline"]["frac"], [(s["name"], s["us_per_step"], s["frac"]) for s in d.get("sub_results", [])])
PY
