#!/usr/bin/env python3
"""sha256 of mfma_fused4 outputs on fixed inputs: run under two builds of libquanto_hip.so and diff (a refactor that must be bit-identical)."""
import hashlib
import json
import os
import sys

import torch

os.environ.setdefault("QUANTO_HIP_EXPERIMENT", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import optimum_quanto_amd  # noqa: F401,E402
from optimum_quanto_amd.library.hip import quanto_hip  # noqa: E402

dev = torch.device("cuda", 0)
for dt in (torch.bfloat16, torch.float16):
    for (M, K, N), env in [((512, 4096, 4096), {}), ((100, 4096, 4096), {}), ((256, 4096, 14336), {}), ((128, 4096, 4096), {"QUANTO_HIP_FUSED4_SPLIT": "4"}),
                           ((640, 14336, 4096), {}), ((1024, 4096, 4096), {"QUANTO_HIP_FUSED4_BM": "128"})]:
        g = torch.Generator(device=dev).manual_seed(7)
        w = (torch.randn((N, K), generator=g, device=dev) * 0.02).to(torch.bfloat16).float()
        packed, scale, shift = bench.quantize_int4(w)
        x = torch.randn((M, K), generator=g, device=dev).to(dt)
        os.environ.update(env)
        y = quanto_hip.lib.qbits_mm(x, packed, scale.to(dt), shift.to(dt), None, 4, 128, N, K, kernel="mfma_fused4")
        torch.cuda.synchronize()
        for k in env:
            os.environ.pop(k, None)
        print(json.dumps({"dtype": str(dt), "M": M, "K": K, "N": N, "env": env, "sha256": hashlib.sha256(y.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:24]}))
