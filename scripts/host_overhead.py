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
            # the C entry alone, arguments marshalled once: what ctypes + the launch cost with nothing of the wrapper around them
            import ctypes
            c = quanto_hip.lib._c
            y = torch.empty((1, 4096), dtype=torch.bfloat16, device=dev)
            vp = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
            a = (vp(x), vp(w._data._data), vp(w._scale), vp(w._shift), ctypes.c_void_p(0), vp(y), 1, 4096, 4096, 4, 128, 2, 2, 2, ctypes.c_void_p(0), 0,
                 ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            t_raw = timeit(lambda: c.quanto_hip_qbits_mm(*a))
        else:
            x2 = x.reshape(-1, 4096)
            t_op = timeit(lambda: torch.ops.quanto.qbytes_mm_bias(x2, w._data, w._scale, None))
            t_lib = timeit(lambda: quanto_hip.lib.qbytes_mm(x2, w._data, w._scale))
            t_raw = None
        t_dense = timeit(lambda: torch.nn.functional.linear(x, lin.weight.to(dev))) if False else None
    import json
    print(json.dumps({"weights": wname, "shape": "(1,4096,4096) eager, us per call (host-bound: 2000 back-to-back calls)", "QLinear_module": round(t_module, 1),
                      "F_linear_qweight": round(t_flinear, 1), "torch_ops_quanto": round(t_op, 1), "python_binding": round(t_lib, 1),
                      "raw_ctypes_call": None if t_raw is None else round(t_raw, 1)}), flush=True)
wd = lin.weight.to(dev)
with torch.no_grad():
    t = timeit(lambda: torch.nn.functional.linear(x, wd))
print(f"dense bf16 F.linear (hipBLASLt): {t:.1f} us per call")
