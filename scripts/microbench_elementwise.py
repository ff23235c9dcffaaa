#!/usr/bin/env python3
"""hipGraph-timed elementwise kernels of the path: quantize_symmetric, quantize_affine, pack, unpack, dequantize_qbits."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimum_quanto_amd.library.hip import quanto_hip  # noqa: E402

lib = quanto_hip.lib
dev = torch.device("cuda", 0)


def timeit(name, fn, nbytes, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:44s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)


N = K = 4096
x = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
s0 = (x.abs().max() / 127).reshape(())
timeit("quantize_symmetric bf16->int8 per-tensor 4096^2", lambda: lib.quantize_symmetric(x, torch.int8, None, s0), N * K * 3)
timeit("quantize_symmetric bf16->e4m3 per-tensor 4096^2", lambda: lib.quantize_symmetric(x, torch.float8_e4m3fn, None, s0), N * K * 3)
sr = (x.abs().amax(dim=1, keepdim=True) / 127)
timeit("quantize_symmetric bf16->int8 per-row 4096^2", lambda: lib.quantize_symmetric(x, torch.int8, 0, sr), N * K * 3)
scale = (torch.rand(N * K // 128, 1, device=dev) * 0.01 + 0.005).to(torch.bfloat16)
shift = (torch.rand(N * K // 128, 1, device=dev) * 0.05 + 0.05).to(torch.bfloat16)
timeit("quantize_affine bf16->uint4 g128 4096^2", lambda: lib.quantize_affine(x, 4, 128, scale, shift), N * K * 3)
q = lib.quantize_affine(x, 4, 128, scale, shift)
timeit("pack 4-bit (65536x2, 128)", lambda: lib.pack(q, 4), N * K * 1.5)
p = lib.pack(q, 4)
timeit("unpack 4-bit", lambda: lib.unpack(p, 4), N * K * 1.5)
timeit("dequantize_qbits int4 g128 -> bf16 4096^2", lambda: lib.dequantize_qbits(p, scale, shift, 4, 128, N, K), N * K * 2.5)
