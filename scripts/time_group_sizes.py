#!/usr/bin/env python3
"""int4 with group sizes 64 / 32 and per-channel scales: us per call of the streaming kernel against what those formats used before
(GEMV passes of 4 rows up to 24 rows, dequantize + dense GEMM beyond); hipGraph of 20 calls, median of 5 replays."""
import json
import os
import sys

import numpy as np
import torch

os.environ.setdefault("QUANTO_HIP_EXPERIMENT", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimum_quanto_amd.library.hip import quanto_hip  # noqa: E402

lib = quanto_hip.lib
dev = torch.device("cuda", 0)


def problem(M, N, K, gs, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    G = K // gs if gs else 1
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16, generator=g)
    packed = torch.randint(0, 256, (N // 2, K), device=dev, dtype=torch.uint8, generator=g)
    scale = (torch.rand(N * G, 1, device=dev, generator=g) * 0.02 + 0.01).to(torch.bfloat16)
    shift = (torch.rand(N * G, 1, device=dev, generator=g) * 0.2).to(torch.bfloat16)
    return x, packed, scale, shift


def time_us(fn, n=20, reps=5):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / n)
    return float(np.median(out))


if len(sys.argv) > 1 and sys.argv[1] == "large":  # r5: prefill sizes - the large-tile int4 GEMM (no dense weight) against dequantize + dense GEMM
    for (M, N, K, gs) in ((4096, 4096, 4800, 96), (4096, 4096, 4096, 32), (4096, 4096, 4096, 128), (8192, 8192, 8256, 96), (8192, 8192, 8192, 32),
                          (8192, 8192, 8192, 128), (2048, 14336, 4800, 96), (16384, 8192, 8192, 128), (4096, 28672, 8192, 128), (4096, 8192, 28672, 128)):
        x, packed, scale, shift = problem(M, N, K, gs)
        row = {"M": M, "N": N, "K": K, "group_size": gs}
        for kernel in ("auto", "mfma_large4", "dequant_mfma"):
            try:
                us = time_us(lambda: lib.qbits_mm(x, packed, scale, shift, None, 4, gs, N, K, kernel=kernel), n=5)
                row[kernel] = round(us, 1)
                if kernel == "auto":
                    row["auto_kernel"] = lib.last_kernel()
            except Exception as e:  # noqa: BLE001
                row[kernel] = type(e).__name__
        print(json.dumps(row), flush=True)
    sys.exit(0)

for gs in (64, 32, None):
    for M in (8, 24, 32, 64, 128, 192):
        N = K = 4096
        x, packed, scale, shift = problem(M, N, K, gs)
        row = {"M": M, "N": N, "K": K, "group_size": gs}
        for kernel in ("auto", "skinny", "gemv", "dequant_mfma"):
            if kernel == "gemv" and M > 24:
                continue
            try:
                us = time_us(lambda: lib.qbits_mm(x, packed, scale, shift, None, 4, gs, N, K, kernel=kernel))
                row[kernel] = round(us, 2)
                if kernel == "auto":
                    row["auto_kernel"] = lib.last_kernel()
            except Exception as e:  # noqa: BLE001
                row[kernel] = type(e).__name__
        print(json.dumps(row), flush=True)
