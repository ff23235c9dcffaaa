#!/bin/bash
# round 4, GPU visit 1: touched parity tests, the new default bench line, AUTO-vs-best sweep, prefetch variants, L1-fill counters
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== tests"; timeout 600 python -m pytest tests/test_hip_parity.py tests/test_activations.py tests/test_multi_linear.py tests/test_host_cpu.py -m gpu -q -x -k "golden or calibrated or sibling or fused_decode or prefetch or symbols" -p no:cacheprovider 2>&1 | tail -5
echo "== bench default"; timeout 900 python bench.py > $OUT/r4_bench_default.json 2> $OUT/r4_bench_default.err; echo "exit=$?"; wc -c $OUT/r4_bench_default.json; tail -3 $OUT/r4_bench_default.err
echo "== prefetch variants"
for wg in 16 64 256; do for nt in 1 0; do
  QUANTO_HIP_EXPERIMENT=1 QUANTO_HIP_PREFETCH_NT=$nt timeout 300 python bench.py --workload northstar --sub layer_decode_b1 layer_decode_b32 --no-cpu-baseline --no-ref-rocm --prefetch-wgs $wg --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for sr in d.get('sub_results',[]): print('wg=$wg nt=$nt', sr['name'], sr['us_per_layer'], sr.get('prefetch_us_per_layer'), sr['frac'], sr.get('prefetch_frac'))"
done; done | tee $OUT/r4_prefetch_variants.txt
echo "== auto vs best"; timeout 900 python scripts/auto_vs_best.py --out $OUT/r04_auto_vs_best.jsonl > /dev/null 2> $OUT/r04_auto_vs_best.err; tail -2 $OUT/r04_auto_vs_best.err
python - <<'PY'
import json
for l in open("gpurun_out/r04_auto_vs_best.jsonl"):
    r = json.loads(l)
    if r["ratio"] > 1.08: print(r["fmt"], r["M"], r["K"], r["N"], r["auto_kernel"], r["auto_us"], r["best"], r["best_us"], r["ratio"])
PY
echo "== L1 fill counters"
C="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
bash scripts/pmc_matmul.sh l1 $C 2>&1 | tail -3
bash scripts/pmc.sh cfg2 l1 $C 2>&1 | tail -3
bash scripts/pmc.sh int4_prefill l1 $C 2>&1 | tail -4
