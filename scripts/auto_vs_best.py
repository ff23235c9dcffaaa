#!/usr/bin/env python3
"""AUTO against every forced kernel, off the grid the dispatch thresholds were fitted on (K, N in {1024, 4096, 14336}).

    python scripts/auto_vs_best.py [--quick] [--out profiles/r04_auto_vs_best.jsonl]

For each shape (K, N from {2048, 5120, 8192, 11008} x M from {1, 8, 32, 96, 256, 1024}) and each weight format (int4 g128, int8) the
call is captured in a hipGraph per kernel choice (AUTO and every kernel that accepts the shape) inside ONE process, decode shapes
rotate over > 256 MB of weights, every graph is replayed five times and the best per-launch time kept.  Prints one JSON line per shape:
the times, the kernel AUTO took, the best forced kernel and AUTO / best.  tests/test_dispatch_auto_gpu.py asserts the ratio.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

QBITS_KERNELS = ("gemv", "mmv", "skinny", "mfma_fused4", "dequant_mfma")
QBYTES_KERNELS = ("gemv", "skinny", "mfma_large", "mfma")
SHAPES = [(m, k, n) for m in (1, 8, 32, 96, 256, 1024) for (k, n) in ((2048, 2048), (5120, 5120), (8192, 8192), (11008, 5120), (5120, 11008))]
QUICK = [(1, 5120, 5120), (8, 5120, 5120), (32, 8192, 8192), (32, 2048, 2048), (96, 5120, 5120), (96, 8192, 8192), (256, 5120, 5120),
         (256, 11008, 5120), (1024, 5120, 5120), (1024, 2048, 2048), (1, 11008, 5120), (8, 8192, 8192)]


def _time_graph(fn, reps, replays=5):
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def measure(fmt, M, K, N, device="cuda"):
    """{"auto": us, "<kernel>": us, ...}, the kernel AUTO ran, for one shape."""
    from optimum_quanto_amd.library.hip import QuantoHipError, quanto_hip

    lib = quanto_hip.lib
    g = torch.Generator(device=device).manual_seed(M * 131 + K + N)
    x = torch.randn((M, K), generator=g, device=device).to(torch.bfloat16)
    wbytes = N * K // 2 if fmt == "int4" else N * K
    nbuf = max(1, min(48, -(-(256 << 20) // wbytes))) if M <= 64 else 1
    reps = max(10, min(60, nbuf)) if M <= 64 else 10
    if fmt == "int4":
        ws = [(torch.randint(0, 256, (N // 2 * (K // 128), 128), generator=g, device=device, dtype=torch.uint8),
               (torch.rand((N * K // 128, 1), generator=g, device=device) * 0.01 + 0.001).to(torch.bfloat16),
               (torch.rand((N * K // 128, 1), generator=g, device=device) * 0.1).to(torch.bfloat16)) for _ in range(nbuf)]
        kernels = QBITS_KERNELS
    else:
        ws = [(torch.randint(-127, 128, (N, K), generator=g, device=device, dtype=torch.int8),
               (torch.rand((N, 1), generator=g, device=device) * 0.01 + 0.001).to(torch.bfloat16)) for _ in range(nbuf)]
        kernels = QBYTES_KERNELS
    out, state = {}, {"i": 0}

    def call(kernel):
        w = ws[state["i"] % nbuf]
        state["i"] += 1
        if fmt == "int4":
            return lib.qbits_mm(x, w[0], w[1], w[2], None, 4, 128, N, K, kernel=kernel)
        return lib.qbytes_mm(x, w[0], w[1], None, kernel=kernel)

    call("auto")
    auto_kernel = lib.last_kernel()
    for kernel in ("auto",) + kernels:
        try:
            call(kernel)
        except QuantoHipError:
            continue
        out[kernel] = round(_time_graph(lambda: call(kernel), reps), 3)
    del ws
    torch.cuda.empty_cache()
    return out, auto_kernel


def sweep(shapes, formats=("int4", "int8")):
    rows = []
    for fmt in formats:
        for (M, K, N) in shapes:
            times, auto_kernel = measure(fmt, M, K, N)
            forced = {k: v for k, v in times.items() if k != "auto"}
            best = min(forced, key=forced.get)
            rows.append({"fmt": fmt, "M": M, "K": K, "N": N, "auto_us": times["auto"], "auto_kernel": auto_kernel, "best": best,
                         "best_us": forced[best], "ratio": round(times["auto"] / forced[best], 3), "all": forced})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import optimum_quanto_amd  # noqa: F401

    rows = sweep(QUICK if args.quick else SHAPES)
    f = open(args.out, "w") if args.out else None
    for r in rows:
        line = json.dumps(r)
        print(line, flush=True)
        if f:
            f.write(line + "\n")
    worst = max(rows, key=lambda r: r["ratio"])
    print(f"# worst: {worst['fmt']} {worst['M']}x{worst['K']}x{worst['N']} auto={worst['auto_kernel']} {worst['auto_us']} us vs {worst['best']} "
          f"{worst['best_us']} us (x{worst['ratio']})", file=sys.stderr)


if __name__ == "__main__":
    main()
