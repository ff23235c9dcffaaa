#!/bin/bash
# closing run of round 2: full GPU suite, smoke, driver-style bench, profile refresh of the kernels that changed last
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2y; O=gpurun_out/r2y
timeout 900 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -n 2 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt
PMC_FOR="none" timeout 600 bash scripts/collect_profiles.sh r02c default int4_decode32 gateup_fused32 > $O/collect.log 2>&1
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2y/bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["roofline"]["frac"], [(s["config"]["name"], s["roofline"]["launch_us"], s["roofline"]["frac"]) for s in d.get("sub_results", [])])
PY
