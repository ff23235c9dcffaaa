#!/usr/bin/env python3
"""Time quanto_hip.qbytes_mm for (activation dtype, weight dtype) pairs at GEMM shapes; prints one JSON line per case.

    python scripts/microbench_qbytes.py [--shapes 4096x4096x4096 ...] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="+", default=["4096x4096x4096", "512x8192x8192", "8192x8192x8192", "2048x14336x4096"])
    ap.add_argument("--pairs", nargs="+", default=["i8:i8", "f8:f8", "bf16:i8", "bf16:f8"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--kernel", default="auto")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--ramp-ms", type=float, default=100.0)
    ap.add_argument("--const", action="store_true", help="constant operands (all ones): separates data-dependent power / clock effects")
    args = ap.parse_args()
    from optimum_quanto_amd.library.hip import quanto_hip

    lib = quanto_hip.lib
    dev = torch.device("cuda", 0)

    def make(kind, rows, K):
        if args.const:
            dt = {"i8": torch.int8, "f8": torch.float8_e4m3fn}.get(kind, torch.bfloat16)
            return torch.ones(rows, K, device=dev, dtype=torch.float32).to(dt)
        if kind == "i8":
            return torch.randint(-127, 128, (rows, K), dtype=torch.int8, device=dev)
        if kind == "f8":
            return torch.randn(rows, K, device=dev).to(torch.float8_e4m3fn)
        return torch.randn(rows, K, device=dev, dtype=torch.bfloat16)

    for shp in args.shapes:
        M, N, K = (int(v) for v in shp.split("x"))
        for pair in args.pairs:
            ak, bk = pair.split(":")
            a, b = make(ak, M, K), make(bk, N, K)
            s = (torch.rand(N, device=dev) * 1e-3).to(torch.bfloat16)
            for _ in range(3):
                lib.qbytes_mm(a, b, s, kernel=args.kernel)
            kern = lib.last_kernel()
            torch.cuda.synchronize()
            import time
            t_ramp = time.perf_counter()  # clock ramp: an idle device needs ~100 ms of load to reach its steady state
            while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
                for _ in range(20):
                    lib.qbytes_mm(a, b, s, kernel=args.kernel)
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if args.graph:  # replay: a call of a few microseconds is shorter than its Python issue time
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(args.iters):
                        lib.qbytes_mm(a, b, s, kernel=args.kernel)
                g.replay()
                torch.cuda.synchronize()
                e0.record()
                g.replay()
                e1.record()
            else:
                e0.record()
                for _ in range(args.iters):
                    lib.qbytes_mm(a, b, s, kernel=args.kernel)
                e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            print(json.dumps({"M": M, "N": N, "K": K, "a": ak, "b": bk, "kernel": kern, "us": round(us, 1),
                              "tops": round(2.0 * M * N * K / us / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
