#!/bin/bash
# round 4: why the hipGraph decode driver is slower than generate() at batch 32 (r3 review): kernel-time breakdown of both drivers
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for drv in graph reference; do
  D=$OUT/prof_cfg5_b32_$drv; rm -rf $D
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python $REPO/scripts/bench_generate.py --batch 32 --drivers $drv --fuse --prompt 512 --new 64 --iterations 1 > $D.log 2>&1)
  grep "^{" $D.log | tail -2
  f=$(find $D -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
def cls(n):
    n = n.lower()
    if "qh::" in n: return "library (qh::)"
    if "attn" in n or "sdpa" in n or "flash" in n or "softmax" in n or "fmha" in n: return "attention"
    if "cijk" in n or "gemm" in n: return "dense gemm (lm_head, bmm)"
    if "index" in n or "copy" in n or "cat" in n or "scatter" in n: return "cache / copies"
    return "elementwise / other"
agg = {}
for r in rows:
    c = cls(r["Name"]); agg[c] = agg.get(c, 0) + float(r["TotalDurationNs"])
print({k: f"{v / tot:.1%}" for k, v in sorted(agg.items(), key=lambda kv: -kv[1])}, "total kernel ms", round(tot / 1e6, 1))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    print("  ", r["Name"][:90], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 2), "ms")
PY
done 2>&1 | tee $OUT/r04_cfg5_b32_graph_vs_generate.txt
