#!/bin/bash
# round 2, GPU call H: what the driver runs at round end - full GPU suite, smoke(), default bench, sharded bench at world 1
set -u
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
cd "$GRAFT_REPO_ROOT"
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.log; grep -v "^  \|^$\|Warning" $O/pytest_gpu.log | tail -6
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 ); tail -12 $O/smoke.log
( timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err ); python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2h/bench_default.json").read().strip().splitlines()[-1])
print("cfg2", d["value"], d["roofline"]["frac"], d["roofline"]["launch_us"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["seconds_per_call"])
for s in d.get("sub_results", []):
    print(s["config"]["name"], s["value"], s["unit"], s["roofline"]["frac"], s["roofline"]["launch_us"], s["roofline"]["kernel"], "cpu", s["cpu_baseline"]["value"], s["cpu_baseline"].get("tinygemm", {}).get("value"))
PY
( timeout 300 python bench.py --shard --steps 20 > $O/bench_shard.json 2> $O/bench_shard.err ); cat $O/bench_shard.json | cut -c1-600
for W in cfg4 int4_prefill int4_prefill512 int4_decode32 w8a8; do timeout 200 python bench.py --workload $W --no-cpu-baseline --no-sub 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['name'], d['value'], d['unit'], d['roofline']['frac'], d['roofline']['launch_us'], d['roofline']['kernel'])"; done
