#!/usr/bin/env python3
"""W4A8 (quantized activation x int4 weight): the fused 8-bit kernel against the reference's own route (dequantize the activation, bf16 product) per token count.
One JSON line per (shape, activation dtype): us of each route (hipGraph replay of 20 calls)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optimum_quanto_amd as Q  # noqa: F401
from optimum_quanto_amd.library.hip import quanto_hip
from optimum_quanto_amd.library import ops as qops

dev = torch.device("cuda", 0)
lib = quanto_hip.lib

def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best

for (N, K) in ((4096, 4096), (14336, 4096), (4096, 14336)):
    packed = torch.randint(0, 256, (N // 2, K), dtype=torch.uint8, device=dev)
    scale = (torch.rand(N * K // 128, 1, device=dev) * 0.02 + 0.01).to(torch.bfloat16)
    shift = (torch.rand(N * K // 128, 1, device=dev) * 0.2).to(torch.bfloat16)
    for M in (96, 128, 256, 512, 768, 1024, 2048, 4096):
        for adt in (torch.int8, torch.float8_e4m3fn):
            a = torch.randint(-100, 100, (M, K), device=dev, dtype=torch.int8) if adt == torch.int8 else torch.randn(M, K, device=dev).to(adt)
            sx = torch.tensor([0.02], device=dev, dtype=torch.bfloat16)
            t_a8 = timed(lambda: lib.qbits_mm_a8(a, sx, packed, scale, shift, None, 4, 128, N, K))
            t_deq = timed(lambda: qops.qbits_mm_a8_default(a, sx, packed, scale, shift, None, 4, 128, N, K))
            print(json.dumps({"M": M, "N": N, "K": K, "act": str(adt).split(".")[-1], "a8_us": round(t_a8, 1), "dequantize_first_us": round(t_deq, 1)}), flush=True)
