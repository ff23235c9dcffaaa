#!/usr/bin/env python3
"""Within-process interleaved A/B of the library's experiment knobs (cdna_hip_programming.md rule 24: N variants x M rounds in
ONE process, report median and min).

    python scripts/ab.py --workloads northstar cfg3 --env QUANTO_HIP_GEMV_VARIANT=0,2,4 [--rounds 7] [--steps 200]

For every value of the variable a hipGraph of ``steps`` calls is captured (the kernel choice is baked in at capture), then
the graphs are replayed round-robin, each replay timed with device events.  One JSON line per (workload, value)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("QUANTO_HIP_EXPERIMENT", "1")  # the library reads its knobs only behind this switch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", nargs="+", required=True)
    ap.add_argument("--env", required=True, help="NAME=v1,v2,...  ('-' = variable unset)")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--ramp-ms", type=float, default=200.0)
    ap.add_argument("--sequential", action="store_true",
                    help="each variant alone: ramp on its own graph, then all its rounds (its own steady clock / power state); default is "
                         "interleaved rounds, where the variants share one clock state")
    args = ap.parse_args()
    name, values = args.env.split("=")
    values = values.split(",")
    device = torch.device("cuda", 0)
    import optimum_quanto_amd  # noqa: F401
    from optimum_quanto_amd.library.hip import quanto_hip

    for wl in args.workloads:
        kind, M, K, N, desc = bench.WORKLOADS[wl]
        flops, nbytes = bench.algorithmic_work(kind, M, K, N)
        Nt = sum(N) if isinstance(N, tuple) else N
        wbytes = Nt * K // 2 if kind.startswith("qbits") else Nt * K
        n_weights = max(1, -(-(512 << 20) // wbytes)) if M <= 64 else 1
        steps = args.steps or (200 if M <= 64 else 20)
        x, sets = bench.build_inputs(kind, M, K, N, device, n_weights, seed=1)
        graphs, kernels = {}, {}
        for v in values:
            if v == "-":
                os.environ.pop(name, None)
            else:
                os.environ[name] = v
            step = bench.make_step(kind, x, sets, K, N)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            kernels[v] = quanto_hip.lib.last_kernel()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(steps):
                        step()
            torch.cuda.current_stream().wait_stream(side)
            g.replay()
            torch.cuda.synchronize()
            graphs[v] = g
        os.environ.pop(name, None)
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < args.ramp_ms:
            for g in graphs.values():
                g.replay()
            torch.cuda.synchronize()
        times = {v: [] for v in values}
        if args.sequential:
            for v in values:
                t0 = time.perf_counter()
                while (time.perf_counter() - t0) * 1e3 < args.ramp_ms:
                    graphs[v].replay()
                    torch.cuda.synchronize()
                for _ in range(args.rounds):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    graphs[v].replay()
                    e1.record()
                    torch.cuda.synchronize()
                    times[v].append(e0.elapsed_time(e1) * 1e3 / steps)
        for _ in range(0 if args.sequential else args.rounds):
            for v in values:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graphs[v].replay()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) * 1e3 / steps)
        for v in values:
            med, mn = float(np.median(times[v])), float(np.min(times[v]))
            print(json.dumps({"workload": wl, name: v, "kernel": kernels[v], "us_median": round(med, 3), "us_min": round(mn, 3),
                              "tflops": round(flops / med / 1e6, 1), "gbps": round(nbytes / med / 1e3, 1), "rounds": args.rounds, "steps": steps, "mode": "sequential" if args.sequential else "interleaved"}), flush=True)
        del graphs, sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
