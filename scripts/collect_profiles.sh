#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + HBM traffic counters for the bench workloads.
#   gpurun --timeout 1200 -- 'bash scripts/collect_profiles.sh r01 cfg2 cfg3 northstar cfg4'
# Results land in gpurun_out/profiles_<round>/ ; copy the summaries you want judged into profiles/ (tracked).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; ROUND=$1; shift
export TMPDIR=/tmp; OUT=$REPO/gpurun_out/profiles_$ROUND; mkdir -p $OUT
# workloads whose HBM-side traffic is collected too (two extra --pmc passes each); override with PMC_FOR="..."
PMC_FOR=${PMC_FOR:-"cfg2 northstar cfg3 gateup_fused qkv_fused cfg4 int4_decode32 int4_prefill int4_prefill512"}
for W in "$@"; do
  # 1. kernel trace + stats of the command bench.py is judged on (default steps/warmup); "default" = the driver's exact command
  #    (cfg2 + the int4 sub_results in one process), any other name = that workload alone
  if [ "$W" = default ]; then WARGS="--no-cpu-baseline"; else WARGS="--workload $W --no-sub --no-cpu-baseline"; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -o $W -- \
      python $REPO/bench.py $WARGS > $OUT/trace_$W.bench.json 2> $OUT/trace_$W.log)
  f=$(find $OUT/trace_$W -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${W}_kernel_stats.csv
  find $OUT/trace_$W -name "*kernel_trace.csv" -delete  # tens of thousands of rows with the clock ramp: only the stats travel back
  case " $PMC_FOR " in *" $W "*) ;; *) [ -f $OUT/${W}_kernel_stats.csv ] && head -4 $OUT/${W}_kernel_stats.csv; continue;; esac
  # 2. HBM traffic: separate --pmc passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2)
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${W}_$C -o pmc -- \
        python $REPO/bench.py --workload $W --no-sub --steps 8 --warmup 2 --ramp-ms 0 --no-cpu-baseline --eager > /dev/null 2> $OUT/pmc_${W}_$C.log)
  done
  python - "$OUT" "$W" <<'PY'
import csv, glob, json, sys, collections
out, w = sys.argv[1], sys.argv[2]
res = {"workload": w}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_{w}_{c}/**/*counter_collection.csv", recursive=True)
    vals = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if "qh::" in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals[r["Kernel_Name"].split("(")[0][:80]].append(float(r["Counter_Value"]))
    res[c] = {k: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for k, v in vals.items()}
json.dump(res, open(f"{out}/{w}_hbm_counters.json", "w"), indent=1)
print(json.dumps(res))
PY
  [ -f $OUT/${W}_kernel_stats.csv ] && head -5 $OUT/${W}_kernel_stats.csv
done
