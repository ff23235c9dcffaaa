#!/bin/bash
# last check of round 3: the full GPU suite, smoke and the default bench on the final sources
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3final_d; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r3final_d/bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["roofline"]["frac"], [(s["name"], s["us_per_step"], s["frac"]) for s in d.get("sub_results", [])])
PY
