#!/bin/bash
# round 4, GPU visit 3: large-tile int4 GEMM - full parity list + SQ counters next to the int8 kernel's
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "large_tile_int4 or int4_prefill_4096" -p no:cacheprovider 2>&1 | tail -15
cat > /tmp/_l4.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
import optimum_quanto_amd
from optimum_quanto_amd.library.hip import quanto_hip
lib = quanto_hip.lib
g = torch.Generator(device="cuda").manual_seed(0)
M = K = N = 4096
x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
w = torch.randint(0, 256, (N // 2 * (K // 128), 128), generator=g, device="cuda", dtype=torch.uint8)
sc = (torch.rand((N * K // 128, 1), generator=g, device="cuda") * 0.01 + 0.001).to(torch.bfloat16)
sh = (torch.rand((N * K // 128, 1), generator=g, device="cuda") * 0.1).to(torch.bfloat16)
for _ in range(6):
    lib.qbits_mm(x, w, sc, sh, None, 4, 128, N, K, kernel=sys.argv[1])
torch.cuda.synchronize()
PY
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  D=$OUT/pmc_l4_$(echo $grp | cut -c1-12 | tr ' ' '_'); rm -rf $D
  (cd /tmp && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $D -o pmc -- python /tmp/_l4.py mfma_large4 > $D.log 2>&1)
  python - "$D" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f: print("no csv"); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "qh::" in r["Kernel_Name"]: agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items(): print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
