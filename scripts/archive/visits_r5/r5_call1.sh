#!/bin/bash
# round 5, visit 1: 128-byte-row native8 kernels (parity + A/B + L1 fill counters), packed-fp32 probe, host cost before the plan cache
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out/r5c1; mkdir -p $OUT; export TMPDIR=/tmp
export QUANTO_HIP_EXPERIMENT=1
echo "== parity" 
timeout 420 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x --timeout 300 \
  -k "native8 or int8_int8 or fp8_fp8 or w8a8 or fp8a8 or int4_prefill or dense_gemm or int8_activations" 2>&1 | tail -5 | tee $OUT/parity_tail.txt
echo "== A/B sequential"
timeout 300 python scripts/ab.py --sequential --rounds 7 --workloads w8a8 fp8a8 cfg4_fp8a8 cfg4_w8a8 int4_prefill --env QUANTO_HIP_NATIVE8_ROW128=0,1 2>&1 | grep -v Warning | tee $OUT/ab_row128.jsonl
echo "== kernel trace"
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- \
   python $REPO/scripts/ab.py --sequential --rounds 3 --workloads w8a8 fp8a8 int4_prefill cfg4_fp8a8 --env QUANTO_HIP_NATIVE8_ROW128=0,1 > $OUT/trace.log 2>&1)
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200 | tee $OUT/kernel_stats_row128_ab.csv
echo "== L1 fill counters"
for R in 0 1; do
  for G in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
    D=$OUT/pmc_r${R}_$(echo $G | cut -d' ' -f1)
    (cd /tmp && QUANTO_HIP_NATIVE8_ROW128=$R timeout 200 rocprofv3 --pmc $G --output-format csv -d $D -o p -- \
       python $REPO/scripts/ab.py --rounds 2 --steps 4 --ramp-ms 0 --workloads w8a8 fp8a8 int4_prefill --env QUANTO_HIP_NATIVE8_ROW128=$R > $D.log 2>&1)
  done
done
python - "$OUT" <<'PY' | tee $OUT/pmc_summary.jsonl
import collections, csv, glob, json, sys
for R in (0, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{sys.argv[1]}/pmc_r{R}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "native8" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(json.dumps({"row128": R, "kernel": k, **{c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches": max(len(v) for v in d.values())}))
PY
echo "== packed fp32 probe"
timeout 300 scripts/probes/pk_probe.bin 10000 2>&1 | tee $OUT/pk_probe.jsonl
timeout 200 python scripts/probes/identity_check.py --launches 60 2>&1 | grep -v Warning | tee $OUT/identity_product.jsonl
timeout 200 python scripts/probes/identity_check.py --launches 60 --lib scripts/probes/libquanto_hip_slp.so 2>&1 | grep -v Warning | tee $OUT/identity_slp.jsonl
echo "== host overhead (before plan cache)"
timeout 120 python scripts/host_overhead.py 2>&1 | grep -v Warning | tee $OUT/host_overhead_before.jsonl
