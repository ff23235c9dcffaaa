#!/bin/bash
# where a decode step's GPU time goes: rocprofv3 kernel stats of the cfg5 harness (batch 1, fused projections, hipGraph decode)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; export TMPDIR=/tmp; O=$REPO/gpurun_out/r2gen; mkdir -p $O
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o gen -- \
   python $REPO/scripts/bench_generate.py --batch 1 --drivers graph --fuse --prompt 64 --new 256 > $O/gen.log 2>&1)
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/cfg5_b1_kernel_stats.csv
find $O/trace -name "*kernel_trace.csv" -delete
tail -n 2 $O/gen.log | cut -c1-300
python - "$O/cfg5_b1_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
ours = sum(float(r["TotalDurationNs"]) for r in rows if "qh::" in r["Name"])
print(f"total kernel time {tot/1e6:.1f} ms, qh:: kernels {ours/1e6:.1f} ms = {100*ours/tot:.1f} %")
for r in rows[:12]:
    print(f'{float(r["Percentage"]):6.2f} %  {int(r["Calls"]):7d} x {float(r["AverageNs"])/1e3:8.2f} us  {r["Name"][:110]}')
PY
