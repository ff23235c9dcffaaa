#!/usr/bin/env python3
"""Host-side cost per QLinear call (decode shape, eager): where the Python time goes between F.linear and the kernel launch."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optimum_quanto_amd as Q  # noqa: E402
from optimum_quanto_amd.library.hip import quanto_hip  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
lin = torch.nn.Linear(4096, 4096, bias=False).to(torch.bfloat16)
results = {}
for wname in ("qint4", "qint8"):
    q = Q.QLinear.from_module(lin, weights=Q.qtypes[wname])
    Q.freeze(q)
    q.to(dev)
    x = torch.randn(1, 1, 4096, dtype=torch.bfloat16, device=dev)
    w = q.weight

    def timeit(fn, n=2000):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    with torch.no_grad():
        t_module = timeit(lambda: q(x))
        t_flinear = timeit(lambda: torch.nn.functional.linear(x, w))
        if wname == "qint4":
            t_op = timeit(lambda: torch.ops.quanto.qbits_mm(x, w._data._data, w._scale, w._shift, None, 4, 128, 4096, 4096))
            t_lib = timeit(lambda: quanto_hip.lib.qbits_mm(x, w._data._data, w._scale, w._shift, None, 4, 128, 4096, 4096))
        else:
            x2 = x.reshape(-1, 4096)
            t_op = timeit(lambda: torch.ops.quanto.qbytes_mm_bias(x2, w._data, w._scale, None))
            t_lib = timeit(lambda: quanto_hip.lib.qbytes_mm(x2, w._data, w._scale))
        t_dense = timeit(lambda: torch.nn.functional.linear(x, lin.weight.to(dev))) if False else None
    print(f"{wname}: module {t_module:.1f} us | F.linear(x, qweight) {t_flinear:.1f} us | torch.ops.quanto.* {t_op:.1f} us | ctypes binding {t_lib:.1f} us", flush=True)
wd = lin.weight.to(dev)
with torch.no_grad():
    t = timeit(lambda: torch.nn.functional.linear(x, wd))
print(f"dense bf16 F.linear (hipBLASLt): {t:.1f} us per call")
