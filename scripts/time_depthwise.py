#!/usr/bin/env python3
"""Depthwise QConv2d (int8 weight): the stencil kernel against the reference's sequence (dequantize the weight, float grouped convolution) - us per call, hipGraph replay."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optimum_quanto_amd as Q
from optimum_quanto_amd.library.hip import quanto_hip

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
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(7):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best

for (B, C, H, k, s) in ((8, 32, 112, 3, 1), (8, 96, 112, 3, 2), (8, 144, 56, 3, 1), (8, 192, 28, 3, 1), (8, 384, 14, 3, 1), (8, 960, 7, 3, 1), (8, 240, 28, 5, 1), (32, 144, 56, 3, 1)):
    conv = torch.nn.Conv2d(C, C, k, stride=s, padding=k // 2, groups=C).to(torch.bfloat16)
    q = Q.QConv2d.from_module(conv, weights=Q.qint8)
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(B, C, H, H, dtype=torch.bfloat16, device="cuda")
    with torch.no_grad():
        t_lib = timed(lambda: q(x))
        kern = quanto_hip.lib.last_kernel()
        wq = q.weight
        t_ref = timed(lambda: torch.nn.functional.conv2d(x, wq.dequantize(), q.bias, s, k // 2, 1, C))
        wd = wq.dequantize()
        t_conv = timed(lambda: torch.nn.functional.conv2d(x, wd, q.bias, s, k // 2, 1, C))
    oh = (H + 2 * (k // 2) - k) // s + 1
    nbytes = (B * C * H * H + B * C * oh * oh) * 2
    print(json.dumps({"B": B, "C": C, "H": H, "k": k, "stride": s, "kernel": kern, "lib_us": round(t_lib, 2), "reference_sequence_us": round(t_ref, 2),
                      "float_conv_alone_us": round(t_conv, 2), "algorithmic_MB": round(nbytes / 1e6, 2), "lib_TBps": round(nbytes / t_lib / 1e6, 3)}), flush=True)
