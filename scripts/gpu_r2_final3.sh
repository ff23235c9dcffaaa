#!/bin/bash
# closing run of round 2: full GPU suite, smoke, driver-style bench, SQ counters of the streaming kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2y; O=gpurun_out/r2y
timeout 900 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -n 2 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt
for W in int4_decode32 gateup_fused32; do
bash scripts/pmc.sh $W sq SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES 2>&1 | tail -n 2 | cut -c1-600
bash scripts/pmc.sh $W sq2 GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD 2>&1 | tail -n 2 | cut -c1-600
done
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2y/bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["roofline"]["frac"], [(s["config"]["name"], s["roofline"]["launch_us"], s["roofline"]["frac"]) for s in d.get("sub_results", [])])
PY
