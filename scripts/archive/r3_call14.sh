#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r3k; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_multi_linear.py -q -m gpu -p no:cacheprovider -k "gemv or multi or decode" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sub northstar cfg3 qkv_fused gateup_fused > $O/bench_decode.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3k/bench_decode.json"))
for s in d["sub_results"]: print(s["name"], s["launch_us"], s["frac"], s["kernel"])
PY
