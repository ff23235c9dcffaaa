#!/usr/bin/env python3
"""Per-call device time of quanto::qbits_mm kernels on a few (M,N,K) shapes (hipGraph replay of 20 calls)."""
import os, sys, torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
from optimum_quanto_amd.library.hip import quanto_hip
lib = quanto_hip.lib
shapes = [tuple(map(int, a.split("x"))) for a in sys.argv[1:]] or [(32,4096,4096),(32,14336,4096),(32,4096,14336),(16,4096,4096),(64,4096,4096),(8,4096,4096),(1,4096,4096)]
for (M,N,K) in shapes:
    g = torch.Generator().manual_seed(0)
    nw = max(1, (512 << 20) // (N*K//2))
    sets = []
    for _ in range(min(nw, 16)):
        packed = torch.randint(0,256,(N*K//256,128),dtype=torch.uint8,generator=g).cuda()
        sc = (torch.rand((N*K//128,1),generator=g)*0.01+0.005).to(torch.bfloat16).cuda(); sh = (torch.rand((N*K//128,1),generator=g)*0.05).to(torch.bfloat16).cuda()
        sets.append((packed, sc, sh))
    x = torch.randn((M,K),generator=g).to(torch.bfloat16).cuda()
    for kern in ("skinny","gemv","mfma","dequant_mfma"):
        i = [0]
        def f():
            p, s, z = sets[i[0] % len(sets)]; i[0] += 1
            return lib.qbits_mm(x, p, s, z, None, 4, 128, N, K, kernel=kern)
        try:
            for _ in range(3): f()
        except Exception as e:
            continue
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(32): f()
        import time
        t_ramp = time.perf_counter()  # clock ramp: an idle device needs ~100 ms of load to reach its steady state
        while time.perf_counter() - t_ramp < 0.1:
            gr.replay(); torch.cuda.synchronize()
        e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1)/32*1000
        print(f"M={M} N={N} K={K} {kern:12s} {us:8.1f} us  {N*K/2/us/1e6:6.2f} TB/s(weights) {2*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
