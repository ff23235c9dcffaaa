#!/bin/bash
# round 4, GPU visit 2: the large-tile int4 GEMM (qbits_mfma_large.hip): parity, then time it against dequantize + dense and the fused kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "large_tile_int4 or int4_prefill_4096" -p no:cacheprovider 2>&1 | tail -15
echo "== timing"
timeout 600 python - <<'PY' 2>&1 | tee $OUT/r4_large4_timing.txt
import sys, torch, json
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import optimum_quanto_amd
from auto_vs_best import _time_graph
from optimum_quanto_amd.library.hip import quanto_hip, QuantoHipError
lib = quanto_hip.lib
g = torch.Generator(device="cuda").manual_seed(0)
for (M, K, N) in [(4096, 4096, 4096), (2048, 4096, 4096), (1024, 4096, 4096), (8192, 4096, 4096), (4096, 4096, 14336), (4096, 14336, 4096), (2048, 4096, 14336), (1024, 4096, 14336), (8192, 8192, 8192)]:
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = torch.randint(0, 256, (N // 2 * (K // 128), 128), generator=g, device="cuda", dtype=torch.uint8)
    sc = (torch.rand((N * K // 128, 1), generator=g, device="cuda") * 0.01 + 0.001).to(torch.bfloat16)
    sh = (torch.rand((N * K // 128, 1), generator=g, device="cuda") * 0.1).to(torch.bfloat16)
    row = {"M": M, "K": K, "N": N}
    for k in ("mfma_large4", "dequant_mfma", "mfma_fused4", "auto"):
        try:
            fn = lambda: lib.qbits_mm(x, w, sc, sh, None, 4, 128, N, K, kernel=k)
            fn(); torch.cuda.synchronize()
            # 300 ms of the same call first: steady clock
            import time
            t0 = time.time()
            while time.time() - t0 < 0.3:
                for _ in range(10): fn()
                torch.cuda.synchronize()
            row[k] = round(_time_graph(fn, 10), 2)
        except QuantoHipError as e:
            row[k] = None
    row["auto_kernel"] = lib.last_kernel()
    print(json.dumps(row), flush=True)
PY
