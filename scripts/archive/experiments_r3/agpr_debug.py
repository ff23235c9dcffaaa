#!/usr/bin/env python3
"""Which accumulator blocks of the 4-wave AGPR layouts go wrong, by K (number of K-tiles through the steady-state loop)?"""
import os
import sys

os.environ["QUANTO_HIP_EXPERIMENT"] = "1"
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import optimum_quanto_amd  # noqa: F401,E402
from optimum_quanto_amd.library.hip import quanto_hip  # noqa: E402

lib = quanto_hip.lib
torch.manual_seed(0)
for cfg in ("1", "5"):
    os.environ["QUANTO_HIP_LARGE_CFG"] = cfg
    for K in (256, 320, 384, 448, 512, 640, 1024, 4096):
        M, N = 256, 256
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8)
        s = torch.ones(N, 1, device="cuda", dtype=torch.bfloat16)
        want = (x.float() @ w.float().t())
        bad_runs = []
        for rep in range(3):
            y = lib.qbytes_mm(x, w, s, kernel="mfma_large")
            err = (y.float() - want).abs() / (want.abs().max())
            blk = err.reshape(16, 16, 16, 16).amax(dim=(1, 3)).cpu().numpy()  # [token block][feature block]
            bad_runs.append((blk > 2e-2))
        bad = bad_runs[0]
        same = all((b == bad).all() for b in bad_runs)
        print(f"cfg {cfg} K={K:5d} nk={K // 64:3d} kernel={lib.last_kernel()} bad blocks {int(bad.sum()):3d}/256 deterministic={same}")
        if bad.any():
            for r in range(16):
                print("   tok blk %2d: " % r + "".join("X" if v else "." for v in bad[r]))
