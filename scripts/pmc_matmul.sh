#!/bin/bash
# SQ counters of the vendor dense bf16 GEMM (torch.matmul -> hipBLASLt) next to this library's cfg2 kernel, same shape, same counters.
#   bash scripts/pmc_matmul.sh <tag> <counters...>      -> gpurun_out/pmc_matmul_<tag>/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; TAG=$1; shift
export TMPDIR=/tmp; OUT=$PWD/gpurun_out/pmc_matmul_${TAG}; mkdir -p $OUT; REPO=$PWD
cat > /tmp/_mm.py <<'PY'
import torch
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
w = (torch.randn(4096, 4096, device="cuda") * 0.02).to(torch.bfloat16)
for _ in range(8):
    torch.matmul(a, w.t())
torch.cuda.synchronize()
PY
(cd /tmp && timeout 300 rocprofv3 --pmc $* --output-format csv -d $OUT -o pmc -- python /tmp/_mm.py > $OUT/log.txt 2>&1)
python - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter csv"); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    n = len(next(iter(d.values())))
    if n >= 4:
        print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", n)
PY
