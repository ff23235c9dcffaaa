#!/bin/bash
# round 4, visit 14: group size 96 in the streaming kernel - parity and time against the kernels AUTO used before
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x -k "small_groups_and_per_channel or other_group_sizes or test_qbits_skinny" 2>&1 | tail -5
timeout 300 python - <<'PY' 2>&1 | tee $OUT/r04_group96_streaming.jsonl
import json, sys, torch
sys.path.insert(0, ".")
from scripts.auto_vs_best import _time_graph
from optimum_quanto_amd.library.hip import quanto_hip, QuantoHipError
lib = quanto_hip.lib
for (M, K, N) in ((8, 1152, 4096), (32, 4800, 4096), (64, 4800, 4096), (32, 2880, 8192), (128, 4800, 4096)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    nb = 24
    ws = [(torch.randint(0, 256, (N // 2 * (K // 96), 96), generator=g, device="cuda", dtype=torch.uint8),
           (torch.rand((N * K // 96, 1), generator=g, device="cuda") * 0.01 + 0.001).to(torch.bfloat16),
           (torch.rand((N * K // 96, 1), generator=g, device="cuda") * 0.1).to(torch.bfloat16)) for _ in range(nb)]
    st = {"i": 0}
    def call(kernel):
        w = ws[st["i"] % nb]; st["i"] += 1
        return lib.qbits_mm(x, w[0], w[1], w[2], None, 4, 96, N, K, kernel=kernel)
    row = {"M": M, "K": K, "N": N, "group": 96}
    for k in ("auto", "skinny", "gemv", "dequant_mfma"):
        try:
            call(k)
        except QuantoHipError:
            continue
        if k == "auto":
            row["auto_kernel"] = lib.last_kernel()
        row[k] = round(_time_graph(lambda: call(k), nb), 2)
    print(json.dumps(row), flush=True)
PY
