#!/usr/bin/env python3
"""One process that launches round 4's late additions a few dozen times each, for `rocprofv3 --kernel-trace --stats`: the streaming int4 kernel
with group size 96 ((32,4800,4096), (64,4800,4096)) and the implicit-GEMM convolution with int4 / int8 weights ((8,128,56,56) -> 128, 3x3)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optimum_quanto_amd as Q  # noqa: E402
from optimum_quanto_amd.library.hip import quanto_hip  # noqa: E402

lib = quanto_hip.lib
g = torch.Generator(device="cuda").manual_seed(1)
K, N = 4800, 4096
ws = [(torch.randint(0, 256, (N // 2 * (K // 96), 96), generator=g, device="cuda", dtype=torch.uint8),
       (torch.rand((N * K // 96, 1), generator=g, device="cuda") * 0.01 + 0.001).to(torch.bfloat16),
       (torch.rand((N * K // 96, 1), generator=g, device="cuda") * 0.1).to(torch.bfloat16)) for _ in range(16)]
for M in (32, 64):
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    for i in range(48):
        w = ws[i % 16]
        lib.qbits_mm(x, w[0], w[1], w[2], None, 4, 96, N, K)
    assert lib.last_kernel() == "skinny"
for wq in ("qint4", "qint8"):
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(128, 128, 3, padding=1).to(torch.bfloat16)
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, wq))
    Q.freeze(q)
    q = q.cuda()
    xc = torch.randn(8, 128, 56, 56, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        for _ in range(30):
            q(xc)
    assert lib.last_kernel() in ("conv2d_mfma", "conv2d_mfma_int4")
torch.cuda.synchronize()
