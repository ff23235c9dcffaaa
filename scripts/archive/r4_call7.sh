#!/bin/bash
# round 4, GPU visit 7: where the large-tile int4 GEMM wins (AUTO threshold), bf16, group size 128; graph-timed after a 300 ms ramp
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cat > /tmp/_t.py <<'PY'
import sys, torch, json, time
sys.path.insert(0, "/root/repo/scripts"); sys.path.insert(0, "/root/repo")
import optimum_quanto_amd
from auto_vs_best import _time_graph
from optimum_quanto_amd.library.hip import quanto_hip
lib = quanto_hip.lib
g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(4096, 8192, 8192), (2048, 8192, 8192), (8192, 8192, 8192), (16384, 4096, 4096), (16384, 8192, 8192), (4096, 8192, 28672), (4096, 28672, 8192),
          (8192, 4096, 14336), (8192, 14336, 4096), (4096, 5120, 13824), (4096, 13824, 5120), (4096, 6144, 6144), (8192, 6144, 6144), (4096, 8192, 4096), (4096, 4096, 8192)]
for (M, K, N) in shapes:
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = torch.randint(0, 256, (N // 2 * (K // 128), 128), generator=g, device="cuda", dtype=torch.uint8)
    sc = (torch.rand((N * K // 128, 1), generator=g, device="cuda") * 0.01 + 0.001).to(torch.bfloat16)
    sh = (torch.rand((N * K // 128, 1), generator=g, device="cuda") * 0.1).to(torch.bfloat16)
    row = {"M": M, "K": K, "N": N}
    for k in ("mfma_large4", "dequant_mfma"):
        fn = lambda: lib.qbits_mm(x, w, sc, sh, None, 4, 128, N, K, kernel=k)
        fn(); torch.cuda.synchronize()
        t0 = time.time()
        while time.time() - t0 < 0.3:
            for _ in range(5): fn()
            torch.cuda.synchronize()
        row[k] = round(_time_graph(fn, 5), 1)
    row["ratio"] = round(row["mfma_large4"] / row["dequant_mfma"], 3)
    print(json.dumps(row), flush=True)
    del x, w, sc, sh
    torch.cuda.empty_cache()
PY
timeout 900 python /tmp/_t.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r04_large4_vs_dequant_dense.jsonl
