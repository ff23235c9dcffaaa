#!/usr/bin/env python3
"""Stress of the split-K hand-over (sc0 sc1 partial sums + arrival counter): thousands of back-to-back launches of split problems that share one
workspace, eager and as hipGraph replays; every result must be bit-identical to the first one and equal to the unsplit kernel's
within fp32 summation-order noise."""
import os
import sys

import torch

os.environ.setdefault("QUANTO_HIP_EXPERIMENT", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimum_quanto_amd.library.hip import quanto_hip  # noqa: E402


def problem(M, K, N, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    packed = torch.randint(0, 256, (N // 2, K), device="cuda", dtype=torch.uint8, generator=g)
    scale = (torch.rand(N * K // 128, 1, device="cuda", generator=g) * 0.02 + 0.01).to(torch.bfloat16)
    shift = (torch.rand(N * K // 128, 1, device="cuda", generator=g) * 0.2).to(torch.bfloat16)
    return x, packed, scale, shift, N, K


def run(p, kernel="auto"):
    x, packed, scale, shift, N, K = p
    return quanto_hip.lib.qbits_mm(x, packed, scale, shift, None, 4, 128, N, K, kernel=kernel)


def main():
    shapes = [(32, 4096, 4096), (32, 4096, 14336), (64, 4096, 4096), (8, 4096, 1024), (24, 14336, 4096), (128, 4096, 4096)]
    probs = [problem(*s, seed=i) for i, s in enumerate(shapes)]
    first = [run(p).clone() for p in probs]
    kernels = []
    for p in probs:
        run(p)
        kernels.append(quanto_hip.lib.last_kernel())
    print("kernels:", kernels)
    bad = torch.zeros((), device="cuda", dtype=torch.int64)
    for it in range(1500):
        for p, f in zip(probs, first):
            bad += (run(p) != f).any().to(torch.int64)
    torch.cuda.synchronize()
    print("eager mismatches:", int(bad))
    # graph replays: the workspace of the capture is zeroed by a node of the graph
    gr = torch.cuda.CUDAGraph()
    outs = []
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for p in probs:
            run(p)
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for rep in range(20):
                outs.append([run(p) for p in probs])
    badg = 0
    for it in range(100):
        gr.replay()
        torch.cuda.synchronize()
        for o in outs:
            for y, f in zip(o, first):
                badg += int((y != f).any())
    print("graph mismatches:", badg)
    ok = int(bad) == 0 and badg == 0
    print("STRESS-OK" if ok else "STRESS-FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
