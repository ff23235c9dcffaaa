#!/bin/bash
# round 4, visit 16: the register-streaming kernel with 32-feature blocks at M = 17..32 (no split-K tail; x re-read by 128 instead of 256 blocks)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=$PWD/gpurun_out; mkdir -p $OUT
export QUANTO_HIP_EXPERIMENT=1
timeout 300 python - <<'PY' 2>&1 | tee $OUT/r04_mmv_fg2_batched_decode.jsonl
import json, os, sys, torch
sys.path.insert(0, ".")
from scripts.auto_vs_best import _time_graph
from optimum_quanto_amd.library.hip import quanto_hip, QuantoHipError
lib = quanto_hip.lib
for (M, K, N) in ((32, 4096, 4096), (24, 4096, 4096), (16, 4096, 4096), (32, 4096, 1024), (32, 4096, 6144)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    nb = 32
    ws = [(torch.randint(0, 256, (N // 2 * (K // 128), 128), generator=g, device="cuda", dtype=torch.uint8),
           (torch.rand((N * K // 128, 1), generator=g, device="cuda") * 0.01 + 0.001).to(torch.bfloat16),
           (torch.rand((N * K // 128, 1), generator=g, device="cuda") * 0.1).to(torch.bfloat16)) for _ in range(nb)]
    st = {"i": 0}
    def call(kernel):
        w = ws[st["i"] % nb]; st["i"] += 1
        return lib.qbits_mm(x, w[0], w[1], w[2], None, 4, 128, N, K, kernel=kernel)
    row = {"M": M, "K": K, "N": N}
    for name, kernel, env in (("skinny", "skinny", {}), ("mmv_fg1", "mmv", {"QUANTO_HIP_MMV_FG": "1"}), ("mmv_fg2", "mmv", {"QUANTO_HIP_MMV_FG": "2"})):
        for k, v in env.items():
            os.environ[k] = v
        try:
            call(kernel)
            row[name] = round(_time_graph(lambda: call(kernel), nb), 2)
        except QuantoHipError as e:
            row[name] = str(e)[:40]
        for k in env:
            os.environ.pop(k)
    print(json.dumps(row), flush=True)
PY
