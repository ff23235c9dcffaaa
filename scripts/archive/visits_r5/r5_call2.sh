#!/bin/bash
# round 5, visit 2: tile raster + scheduling variants of the 128-byte-row kernels, SQ / TCP counters, extended packed-fp32 probe, host cost with
# the plan cache, first full run of the reworked bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c2; mkdir -p $OUT; export TMPDIR=/tmp
export QUANTO_HIP_EXPERIMENT=1
echo "== parity"
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_qconv2d.py -m gpu -q -p no:cacheprovider -x --timeout 300 \
  -k "native8 or int8_int8 or fp8_fp8 or w8a8 or fp8a8 or int4_prefill or dense_gemm or int8_activations or plan_cache or qconv2d" 2>&1 | tail -4 | tee $OUT/parity_tail.txt
echo "== raster"
timeout 300 python scripts/ab.py --sequential --rounds 5 --workloads w8a8 fp8a8 cfg4_fp8a8 cfg4_w8a8 int4_prefill --env QUANTO_HIP_NATIVE8_GROUP_M=1,2,4,8 2>&1 | grep -v Warning | tee $OUT/ab_group_m.jsonl
echo "== variants"
timeout 300 python scripts/ab.py --sequential --rounds 5 --workloads w8a8 int4_prefill --env QUANTO_HIP_R128_VARIANT=0,1,2,3 2>&1 | grep -v Warning | tee $OUT/ab_variants.jsonl
echo "== counters"
for GM in 1 4; do
  for G in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum"; do
    D=$OUT/pmc_gm${GM}_$(echo $G | cut -d' ' -f1)
    (cd /tmp && QUANTO_HIP_NATIVE8_GROUP_M=$GM timeout 200 rocprofv3 --pmc $G --output-format csv -d $D -o p -- \
       python $REPO/scripts/ab.py --rounds 2 --steps 4 --ramp-ms 0 --workloads w8a8 fp8a8 int4_prefill cfg4_fp8a8 --env QUANTO_HIP_NATIVE8_GROUP_M=$GM > $D.log 2>&1)
  done
done
python - "$OUT" <<'PY' | tee $OUT/pmc_summary.jsonl
import collections, csv, glob, json, sys
for GM in (1, 4):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{sys.argv[1]}/pmc_gm{GM}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "native8" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(json.dumps({"group_m": GM, "kernel": k, **{c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches": max(len(v) for v in d.values())}))
PY
echo "== packed fp32 probe"
timeout 300 scripts/probes/pk_probe.bin 10000 2>&1 | tee $OUT/pk_probe.jsonl
echo "== host overhead (plan cache; experiment knobs off so that it is active)"
QUANTO_HIP_EXPERIMENT=0 timeout 120 python scripts/host_overhead.py 2>&1 | grep -v Warning | tee $OUT/host_overhead_after.jsonl
echo "== bench line"
QUANTO_HIP_EXPERIMENT=0 QH_BENCH_KEEP_TRACE=$OUT/bench_kernel_trace.csv timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 1500 $OUT/bench_err.txt; wc -c $OUT/bench_line.json
python - "$OUT/bench_line.json" <<'PY'
import json, sys
try:
    p = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("headline", p["value"], p["ms_per_step"], p["roofline"]["frac"], "kernel_us", p["roofline"].get("kernel_us"), "traffic", p["roofline"].get("traffic"), p.get("ref_rocm_us"), p.get("profile_passes"))
    for sr in p["sub_results"]:
        print({k: v for k, v in sr.items() if k not in ("cpu", "alg_bytes", "method", "model", "shape", "bound", "ref_rocm", "launches")})
except Exception as e:
    print("bench line unreadable:", e)
PY
