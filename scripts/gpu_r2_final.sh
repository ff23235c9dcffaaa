#!/bin/bash
# round-2 final validation + profile refresh (one gpurun call)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2z; O=gpurun_out/r2z
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -n 3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt
PMC_FOR="int8_gateup_fused qkv_fused32" timeout 900 bash scripts/collect_profiles.sh r02b default int4_decode32 qkv_fused32 gateup_fused32 int4_decode8 int8_gateup_fused int8_qkv_fused32 > $O/collect.log 2>&1
timeout 500 python scripts/bench_generate.py --batch 1 32 --drivers graph --fuse --new 256 > $O/gen_fuse.log 2>&1
tail -n 4 $O/gen_fuse.log | cut -c1-400
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2z/bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["roofline"]["frac"], [(s["name"], s["roofline"]["launch_us"], s["roofline"]["frac"]) for s in d.get("sub_results", [])])
PY
