#!/usr/bin/env python3
"""cfg2 (bf16 x int8, 4096^3): what could overlapping the work OUTSIDE the K loop buy?  (r5 review, item 3)

All 256 tiles of 256 x 256 start and end together on the 256 CUs, so entry (first K-tile's latency), epilogue (32 MB leaving at once) and
launch / drain are not hidden behind anything.  Three measurements, no kernel change, each shape alone in its steady clock / power state
(hipGraph of 20 calls, 300 ms ramp, 7 timed replays):

  k-sweep  (4096, K, 4096) for K = 512 .. 16384 on the product kernel: t = F + P * K / 64.  F is everything a perfect overlap could hide;
           P * 64 is what the launch would cost if ALL of it were hidden - the bound on ANY overlap scheme for this loop.
  m-sweep  (M, 4096, 4096) for M = 4096, 8192, 16384: 1, 2, 4 tiles per CU back to back (no co-residency: 144 KiB of LDS per workgroup).
  cfgs     4096^3 on the other tile configurations (QUANTO_HIP_LARGE_CFG): 2 = 128 x 128 tiles, 1024 workgroups, TWO co-resident per CU
           (72 KiB of LDS each) - the "co-scheduled smaller tiles whose epilogues overlap the neighbour's loop" structure, already in the
           library; 3 = 256 x 256 as 1 x 8 waves.

One JSON line per point; the fit is printed last."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("QUANTO_HIP_EXPERIMENT", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def time_shape(M, K, N, cfg, dev, lib, rounds=7, steps=20, ramp_ms=300.0, kind="qbytes_i8"):
    if cfg is None:
        os.environ.pop("QUANTO_HIP_LARGE_CFG", None)
    else:
        os.environ["QUANTO_HIP_LARGE_CFG"] = str(cfg)
    x, sets = bench.build_inputs(kind, M, K, N, dev, 1, seed=1)
    step = bench.make_step(kind, x, sets, K, N)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    kernel = lib.last_kernel()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(steps):
            step()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ramp_ms:
        g.replay()
        torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / steps)
    os.environ.pop("QUANTO_HIP_LARGE_CFG", None)
    del g, x, sets
    torch.cuda.empty_cache()
    med = float(np.median(ts))
    flops = 2.0 * M * K * N
    return {"M": M, "K": K, "N": N, "cfg": cfg, "kernel": kernel, "us_median": round(med, 2), "us_min": round(float(np.min(ts)), 2),
            "tflops": round(flops / med / 1e6, 1), "frac_of_2500": round(flops / med / 1e6 / 2500.0, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", type=int, nargs="*", default=[512, 1024, 2048, 4096, 8192, 16384])
    ap.add_argument("--ms", type=int, nargs="*", default=[4096, 8192, 16384])
    ap.add_argument("--cfgs", type=int, nargs="*", default=[0, 2, 3])
    ap.add_argument("--kind", default="qbytes_i8")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    import optimum_quanto_amd  # noqa: F401
    from optimum_quanto_amd.library.hip import quanto_hip

    lib = quanto_hip.lib
    pts = []
    for K in args.ks:
        r = time_shape(4096, K, 4096, 0, dev, lib, kind=args.kind)
        r["sweep"] = "k"
        pts.append(r)
        print(json.dumps(r), flush=True)
    ks = np.array([p["K"] / 64.0 for p in pts])
    ts = np.array([p["us_median"] for p in pts])
    if len(ks) >= 2:
        P, F = np.polyfit(ks, ts, 1)
        at = 64.0
        print(json.dumps({"fit": "t_us = F + P * (K / 64)", "F_us": round(float(F), 2), "P_us_per_k_tile": round(float(P), 4),
                          "residual_us_max": round(float(np.max(np.abs(F + P * ks - ts))), 2),
                          "k4096_loop_only_us": round(float(P * at), 2), "k4096_loop_only_frac_of_2500": round(2.0 * 4096 ** 3 / (P * at) / 1e6 / 2500.0, 4),
                          "k4096_half_of_F_hidden_frac": round(2.0 * 4096 ** 3 / (P * at + F / 2) / 1e6 / 2500.0, 4)}), flush=True)
    for M in args.ms:
        r = time_shape(M, 4096, 4096, 0, dev, lib, kind=args.kind)
        r["sweep"] = "m"
        r["tiles_per_cu"] = M // 4096
        r["us_per_round"] = round(r["us_median"] / (M // 4096), 2)
        print(json.dumps(r), flush=True)
    for cfg in args.cfgs:
        r = time_shape(4096, 4096, 4096, cfg, dev, lib, kind=args.kind)
        r["sweep"] = "cfg"
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
