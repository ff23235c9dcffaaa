#!/usr/bin/env python3
"""Fixed cost and per-K cost of the quantized-activation GEMMs (qmm_native8.hip) next to the vendor's: (4096, 4096, K) for K = 1024 .. 8192,
hipGraph of 20 calls, 300 ms clock ramp, best of 5 replays; a line per (op, K) and a least-squares fit t = F + P * K / 128 per op."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimum_quanto_amd.library.hip import quanto_hip  # noqa: E402

lib = quanto_hip.lib
dev = torch.device("cuda", 0)
M = N = 4096


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        g.replay()
        torch.cuda.synchronize()
    best = float("inf")
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
    return best


rows = {}
for K in (1024, 2048, 4096, 8192):
    a8 = torch.randint(-127, 128, (M, K), device=dev, dtype=torch.int8)
    w8 = torch.randint(-127, 128, (N, K), device=dev, dtype=torch.int8)
    s = (torch.rand(N, 1, device=dev) * 1e-3 + 1e-3).to(torch.bfloat16)
    af = torch.randn(M, K, device=dev).to(torch.float8_e4m3fn)
    wf = torch.randn(N, K, device=dev).to(torch.float8_e4m3fn)
    ab = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    wb = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    one = torch.ones((), device=dev, dtype=torch.float32)
    ops = {
        "quanto_hip int8 x int8 (+ rescale, bf16 out)": lambda: lib.qbytes_mm(a8, w8, s),
        "quanto_hip e4m3fn x e4m3fn (+ rescale, bf16 out)": lambda: lib.qbytes_mm(af, wf, s),
        "hipBLASLt _int_mm (int32 out, no rescale)": lambda: torch._int_mm(a8, w8.t()),
        "hipBLASLt _scaled_mm e4m3fn (bf16 out)": lambda: torch._scaled_mm(af, wf.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16),
        "hipBLASLt matmul bf16": lambda: torch.matmul(ab, wb.t()),
    }
    for name, fn in ops.items():
        try:
            us = timed(fn)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"op": name, "K": K, "error": str(e)[:120]}), flush=True)
            continue
        rows.setdefault(name, []).append((K, us))
        print(json.dumps({"op": name, "M": M, "N": N, "K": K, "us": round(us, 2)}), flush=True)
for name, pts in rows.items():
    k = np.array([p[0] for p in pts], dtype=np.float64)
    t = np.array([p[1] for p in pts])
    A = np.stack([np.ones_like(k), k / 128.0], axis=1)
    (F, P), *_ = np.linalg.lstsq(A, t, rcond=None)
    print(json.dumps({"op": name, "fit_fixed_us": round(float(F), 2), "fit_us_per_128_bytes_of_K": round(float(P), 4)}), flush=True)
