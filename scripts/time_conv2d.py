#!/usr/bin/env python3
"""QConv2d on the device: the implicit-GEMM kernel (quanto::qbytes_conv2d; `qint4` as first argument: quanto::qbits_conv2d) against the materialised
im2col + quanto::qbytes_mm / qbits_mm lowering and the reference's behaviour (dequantize + float convolution).  One JSON line per shape;
hipGraph-timed, best of five replays."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import optimum_quanto_amd as Q  # noqa: E402
from auto_vs_best import _time_graph  # noqa: E402
from optimum_quanto_amd.library.hip import quanto_hip  # noqa: E402
from optimum_quanto_amd.tensor.weights import conv2d_as_gemm  # noqa: E402

WEIGHTS = sys.argv[1] if len(sys.argv) > 1 else "qint8"
SHAPES = [(8, 256, 56, 256, 3, 1, 1), (8, 128, 56, 128, 3, 1, 1), (32, 512, 14, 512, 3, 1, 1), (8, 64, 112, 128, 3, 2, 1), (8, 512, 28, 128, 1, 1, 0)]
if len(sys.argv) > 2 and sys.argv[2] == "stems":  # r5: ragged K (RGB stems) and wide windows - shapes that fell back to F.unfold + GEMM until r5
    SHAPES = [(8, 3, 224, 64, 7, 2, 3), (32, 3, 224, 64, 7, 2, 3), (8, 3, 224, 32, 3, 2, 1), (8, 3, 32, 64, 3, 1, 1), (8, 16, 64, 64, 9, 1, 4), (8, 3, 227, 96, 11, 4, 0)]
if len(sys.argv) > 2 and sys.argv[2] == "strided":  # r5: three-tap windows the pair form cannot take (stride 2, 7 x 7 maps): the one-pixel row form
    SHAPES = [(8, 64, 112, 128, 3, 2, 1), (8, 128, 56, 256, 3, 2, 1), (8, 512, 7, 512, 3, 1, 1), (32, 512, 7, 512, 3, 1, 1)]
if len(sys.argv) > 2 and sys.argv[2] == "grid":  # the sweep behind the dispatch rule of tensor/weights.py (_implicit_conv2d_wins)
    SHAPES = [(8, 64, 56, 64, 3, 1, 1), (32, 64, 56, 64, 3, 1, 1), (8, 128, 28, 128, 3, 1, 1), (32, 128, 28, 128, 3, 1, 1), (8, 192, 28, 192, 3, 1, 1),
              (8, 256, 14, 256, 3, 1, 1), (32, 256, 14, 256, 3, 1, 1), (8, 256, 28, 256, 3, 1, 1), (8, 320, 32, 320, 3, 1, 1), (8, 512, 7, 512, 3, 1, 1),
              (32, 512, 7, 512, 3, 1, 1), (1, 128, 56, 128, 3, 1, 1), (1, 256, 28, 256, 3, 1, 1), (8, 64, 56, 256, 1, 1, 0), (8, 256, 56, 64, 1, 1, 0),
              (8, 128, 28, 128, 5, 1, 2), (8, 64, 64, 64, 7, 1, 3), (8, 128, 56, 256, 3, 2, 1), (8, 512, 28, 128, 1, 1, 0), (8, 256, 14, 1024, 1, 1, 0),
              (8, 1024, 14, 256, 1, 1, 0), (1, 512, 28, 128, 1, 1, 0)]
for (B, C, H, OC, k, s, p) in SHAPES:
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(C, OC, k, stride=s, padding=p).to(torch.bfloat16)
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, WEIGHTS))
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(B, C, H, H, device="cuda").to(torch.bfloat16)
    w = q.weight
    if WEIGHTS == "qint8":
        scale = w._scale.reshape(-1, 1).expand(OC, 1).contiguous()
        data2d = w._data.reshape(OC, -1)
        gemm = lambda a: torch.ops.quanto.qbytes_mm_bias(a, data2d, scale, q.bias)  # noqa: E731
    else:
        gemm = lambda a: torch.ops.quanto.qbits_mm(a, w._data._data, w._scale, w._shift, q.bias, 4, w._group_size, OC, C * k * k)  # noqa: E731
    wdq = w.dequantize()
    if WEIGHTS == "qint8":  # the convolution kernel called directly (the module routes pointwise convolutions to permute + GEMM)
        direct = lambda: torch.ops.quanto.qbytes_conv2d(x, w._data, w._scale, q.bias, [s, s], [p, p], [1, 1])  # noqa: E731
    else:
        direct = lambda: torch.ops.quanto.qbits_conv2d(x, w._data._data, w._scale, w._shift, q.bias, 4, w._group_size, list(w.shape), [s, s], [p, p], [1, 1])  # noqa: E731
    if os.environ.get("TIME_CONV2D_DIRECT_ONLY"):  # A/B of kernel knobs: the convolution kernel alone
        with torch.no_grad():
            t_dir = _time_graph(direct, 5)
        print(json.dumps({"weights": WEIGHTS, "B": B, "C": C, "H": H, "OC": OC, "k": k, "stride": s, "K": C * k * k, "conv_kernel_direct_us": round(t_dir, 1),
                          "pair": os.environ.get("QUANTO_HIP_CONV_PAIR", "auto"), "split": os.environ.get("QUANTO_HIP_CONV_SPLIT", "auto"),
                          "rows": os.environ.get("QUANTO_HIP_CONV_ROWS", "auto"), "kernel": quanto_hip.lib.last_kernel()}), flush=True)
        continue
    with torch.no_grad():
        t_dir = _time_graph(direct, 5)
        t_imp = _time_graph(lambda: q(x), 5)
        t_unf = _time_graph(lambda: conv2d_as_gemm(x, w, q.bias, (s, s), (p, p), (1, 1), 1, gemm), 5)
        t_ref = _time_graph(lambda: torch.nn.functional.conv2d(x, w.dequantize(), q.bias, s, p), 5)
        t_dense = _time_graph(lambda: torch.nn.functional.conv2d(x, wdq, q.bias, s, p), 5)
    OHW = (H + 2 * p - k) // s + 1
    print(json.dumps({"weights": WEIGHTS, "group_size": getattr(w, "_group_size", None), "B": B, "C": C, "H": H, "OC": OC, "k": k, "stride": s, "M": B * OHW * OHW, "K": C * k * k, "implicit_gemm_us": round(t_imp, 1), "conv_kernel_direct_us": None if t_dir is None else round(t_dir, 1),
                      "unfold_plus_gemm_us": round(t_unf, 1), "reference_dequantize_plus_conv_us": round(t_ref, 1), "float_conv_only_us": round(t_dense, 1),
                      "im2col_bytes_not_written": B * OHW * OHW * C * k * k * 2}), flush=True)
