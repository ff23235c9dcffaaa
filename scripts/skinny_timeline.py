#!/usr/bin/env python3
"""In-kernel timeline of the streaming MFMA kernel (qbits_skinny): thread 0 of every block stamps s_memtime at the phase
boundaries into a buffer handed over through QUANTO_HIP_SKINNY_TIMELINE.  Prints, per phase, the median / p10 / p90 over the
blocks of the time since the first block started.

    python scripts/skinny_timeline.py --workload int4_decode32 [--launches 20]
"""
import argparse
import os
import sys

import numpy as np
import torch

os.environ.setdefault("QUANTO_HIP_EXPERIMENT", "1")  # the library reads its knobs only behind this switch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SLOTS = {0: "entry", 1: "prologue: DMA issued, scale/shift tables in LDS", 20: "loop done", 21: "partials stored + acked",
         22: "arrival counted", 23: "partials of all splits loaded + added (last block)", 24: "output stored"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="int4_decode32")
    ap.add_argument("--launches", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    import optimum_quanto_amd  # noqa: F401
    from optimum_quanto_amd.library.hip import quanto_hip

    kind, M, K, N, _ = bench.WORKLOADS[args.workload]
    wbytes = N * K // 2
    x, sets = bench.build_inputs(kind, M, K, N, dev, max(1, -(-(512 << 20) // wbytes)), seed=1)
    step = bench.make_step(kind, x, sets, K, N)
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    nblocks = 4096
    tl = torch.zeros(nblocks * 32, dtype=torch.int64, device=dev)
    os.environ["QUANTO_HIP_SKINNY_TIMELINE"] = hex(tl.data_ptr())
    rows = []
    for _ in range(args.launches):
        tl.zero_()
        for _ in range(8):  # keep the clocks up, the last launch is the one that stays in the buffer
            step()
        torch.cuda.synchronize()
        t = tl.cpu().numpy().reshape(nblocks, 32)
        used = t[:, 0] != 0
        rows.append(t[used])
    os.environ.pop("QUANTO_HIP_SKINNY_TIMELINE")
    print("kernel", quanto_hip.lib.last_kernel(), "blocks", rows[0].shape[0])
    # s_memtime counters are per XCD (not synchronised): every block is measured against its own entry stamp; entry skew
    # across blocks comes from the global 100 MHz clock (wall_clock64, 10 ns steps)
    r = rows[-1]
    done = r[:, 24] != 0
    ticks = (r[done, 24] - r[done, 0]).astype(np.float64)
    wall = (r[done, 31] - r[done, 30]).astype(np.float64) * 10.0
    ns_per_tick = float(np.median(wall / ticks))
    print(f"s_memtime: {1e3 / ns_per_tick:.1f} MHz (calibrated on {int(done.sum())} blocks against wall_clock64)")
    skew, acc = [], {}
    for r in rows[2:]:
        skew.append((r[:, 30] - r[:, 30].min()).astype(np.float64) * 10.0)
        span = (r[:, 31].max() - r[:, 30].min()) * 10.0
        for slot in list(SLOTS) + list(range(3, 19)):
            ok = r[:, slot] != 0
            if ok.any():
                acc.setdefault(slot, []).append((r[ok, slot] - r[ok, 0]).astype(np.float64) * ns_per_tick)
    skew = np.concatenate(skew)
    print(f"block entry skew (wall clock): p10 {np.percentile(skew, 10):.0f}  median {np.median(skew):.0f}  p90 {np.percentile(skew, 90):.0f}  max {skew.max():.0f} ns;"
          f"  first entry -> last exit of the last launch: {span:.0f} ns")
    print("time since the block's own entry:")
    for slot in sorted(acc):
        v = np.concatenate(acc[slot])
        name = SLOTS.get(slot, f"tiles {2 * (slot - 3)},{2 * (slot - 3) + 1} landed, barrier passed")
        print(f"{slot:2d} {name:52s} n={v.size // len(acc[slot]):5d}  p10 {np.percentile(v, 10):7.0f}  median {np.median(v):7.0f}  p90 {np.percentile(v, 90):7.0f}  max {v.max():7.0f} ns")


if __name__ == "__main__":
    main()
