#!/bin/bash
# rocprofv3 PMC passes for one bench workload: bash scripts/pmc.sh <workload> <tag> <counters...>
cd "${GRAFT_REPO_ROOT:-/root/repo}"; W=$1; TAG=$2; shift 2
export TMPDIR=/tmp; OUT=$PWD/gpurun_out/pmc_${W}_${TAG}; mkdir -p $OUT; REPO=$PWD
(cd /tmp && timeout 300 rocprofv3 --pmc $* --output-format csv -d $OUT -o pmc -- python $REPO/bench.py --workload $W --no-sub --steps 6 --warmup 2 --ramp-ms 0 --no-cpu-baseline --no-ref-rocm --eager > $OUT/log.txt 2>&1)
python - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter csv", glob.glob(sys.argv[1] + "/**/*", recursive=True)[:10]); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "qh::" in k:
        print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
