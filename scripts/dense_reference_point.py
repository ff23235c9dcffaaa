#!/usr/bin/env python3
"""Reference point for the MFMA roofline fractions: the vendor dense bf16 GEMM (torch.matmul -> hipBLASLt) on the same box,
same shapes, same clock-ramp discipline as bench.py.  Not part of the product path; prints one JSON line per shape."""
import json
import sys
import time

import torch


def timed(fn, name, M, N, K):
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
    print(json.dumps({"op": name, "M": M, "N": N, "K": K, "us": round(best, 1), "tflops": round(2.0 * M * N * K / best / 1e6, 1)}), flush=True)


def main():
    const = "--const" in sys.argv
    argv = [a for a in sys.argv[1:] if a not in ("--const", "--eight-bit")]
    shapes = [tuple(int(v) for v in s.split("x")) for s in (argv or ["4096x4096x4096", "512x8192x8192", "8192x8192x8192"])]
    dev = torch.device("cuda", 0)
    for M, N, K in shapes:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        if const:  # constant operands: separates data-dependent power / clock effects
            a.fill_(1.0)
            w.fill_(1.0)
        for _ in range(3):
            torch.matmul(a, w.t())
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                torch.matmul(a, w.t())
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            g.replay()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(json.dumps({"op": "torch.matmul bf16 (hipBLASLt)", "M": M, "N": N, "K": K, "us": round(us, 1), "tflops": round(2.0 * M * N * K / us / 1e6, 1)}), flush=True)
        if "--eight-bit" in sys.argv:  # r5: the vendor's 8-bit GEMMs next to qmm_native8.hip's int8 x int8 / fp8 x fp8 kernels (the bare GEMM: no rescale pass)
            timed(lambda a8=torch.randint(-127, 128, (M, K), device=dev, dtype=torch.int8), w8=torch.randint(-127, 128, (N, K), device=dev, dtype=torch.int8):
                  torch._int_mm(a8, w8.t()), "torch._int_mm int8 x int8 -> int32 (hipBLASLt)", M, N, K)
            try:
                f8 = torch.float8_e4m3fn
                one = torch.ones((), device=dev, dtype=torch.float32)
                timed(lambda af=a.to(f8), wf=w.to(f8): torch._scaled_mm(af, wf.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16),
                      "torch._scaled_mm e4m3fn x e4m3fn -> bf16 (hipBLASLt)", M, N, K)
            except Exception as e:  # noqa: BLE001
                print(json.dumps({"op": "torch._scaled_mm e4m3fn", "error": str(e)[:200]}), flush=True)


if __name__ == "__main__":
    main()
