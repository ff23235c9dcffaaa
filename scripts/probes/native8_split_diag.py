#!/usr/bin/env python3
"""Where does a forced split of the quantized-activation GEMM differ from the unsplit kernel?"""
import os, sys
import numpy as np
import torch
os.environ["QUANTO_HIP_EXPERIMENT"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from optimum_quanto_amd.library.hip import quanto_hip

def run(M, N, K, small, split, ticks="20000"):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda", generator=g)
    b = torch.randint(-127, 128, (N, K), dtype=torch.int8, device="cuda", generator=g)
    s = torch.ones(N, device="cuda", dtype=torch.float32)
    os.environ["QUANTO_HIP_NATIVE8_SMALL"] = small
    os.environ["QUANTO_HIP_NATIVE8_POLL_TICKS"] = ticks
    os.environ["QUANTO_HIP_NATIVE8_SPLIT"] = "1"
    ref = quanto_hip.lib.qbytes_mm(a, b, s, kernel="mfma_native8")
    want = (a.double() @ b.double().t()).float()
    os.environ["QUANTO_HIP_NATIVE8_SPLIT"] = split
    for rep in range(3):
        y = quanto_hip.lib.qbytes_mm(a, b, s, kernel="mfma_native8")
        bad = (y != want).nonzero()
        rows = sorted(set(bad[:, 0].tolist()))
        cols = sorted(set(bad[:, 1].tolist()))
        print(f"M{M} N{N} K{K} small={small} S={split} ticks={ticks} rep{rep}: ref_ok={bool((ref == want).all())} bad={len(bad)}",
              f"rows {rows[:6]}..{rows[-3:]} ({len(rows)}) cols {cols[:6]}..{cols[-3:]} ({len(cols)})" if len(bad) else "")
        if len(bad):
            r, c = bad[0].tolist()
            print("   first", r, c, float(y[r, c]), float(want[r, c]), "diff", float(y[r, c] - want[r, c]))

for small in ("1", "0"):
    for split in ("2", "4"):
        run(256, 256, 3072, small, split)
        run(128, 128, 3072, small, split)
        run(300, 700, 6144, small, split)
run(256, 256, 3072, "1", "2", "0")
