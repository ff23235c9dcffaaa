#!/bin/bash
# round 4, GPU visit 4: large-tile int4 GEMM after the LDS-weights rewrite: parity (x3 for races), identity checks, timing of both rounding forms
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3; do echo "== parity run $i"; timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "large_tile_int4 or int4_prefill_4096" -p no:cacheprovider 2>&1 | tail -4; done
echo "== identity x3"; for i in 1 2 3; do python scripts/debug_l4.py 2>&1 | grep mismatches | tr '\n' ' '; echo; done
echo "== identity, C++ rounding"; QUANTO_HIP_EXPERIMENT=1 QUANTO_HIP_L4_ASM_ROUND=0 python scripts/debug_l4.py 2>&1 | grep mismatches | tr '\n' ' '; echo
echo "== timing"
cat > /tmp/_t.py <<'PY'
import sys, torch, json, time, os
sys.path.insert(0, "/root/repo/scripts"); sys.path.insert(0, "/root/repo")
import optimum_quanto_amd
from auto_vs_best import _time_graph
from optimum_quanto_amd.library.hip import quanto_hip, QuantoHipError
lib = quanto_hip.lib
g = torch.Generator(device="cuda").manual_seed(0)
for (M, K, N) in [(4096, 4096, 4096), (2048, 4096, 4096), (8192, 4096, 4096), (4096, 4096, 14336), (4096, 14336, 4096), (1024, 4096, 14336), (8192, 8192, 8192)]:
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = torch.randint(0, 256, (N // 2 * (K // 128), 128), generator=g, device="cuda", dtype=torch.uint8)
    sc = (torch.rand((N * K // 128, 1), generator=g, device="cuda") * 0.01 + 0.001).to(torch.bfloat16)
    sh = (torch.rand((N * K // 128, 1), generator=g, device="cuda") * 0.1).to(torch.bfloat16)
    row = {"M": M, "K": K, "N": N}
    for k, env in (("mfma_large4", "1"), ("mfma_large4", "0"), ("dequant_mfma", "1")):
        os.environ["QUANTO_HIP_L4_ASM_ROUND"] = env
        fn = lambda: lib.qbits_mm(x, w, sc, sh, None, 4, 128, N, K, kernel=k)
        fn(); torch.cuda.synchronize()
        t0 = time.time()
        while time.time() - t0 < 0.3:
            for _ in range(10): fn()
            torch.cuda.synchronize()
        row[k + ("_asm" if env == "1" and k == "mfma_large4" else "_cpp" if k == "mfma_large4" else "")] = round(_time_graph(fn, 10), 2)
    print(json.dumps(row), flush=True)
PY
QUANTO_HIP_EXPERIMENT=1 timeout 600 python /tmp/_t.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r4_large4_timing2.txt
