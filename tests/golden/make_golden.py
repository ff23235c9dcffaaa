#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

The reference (optimum-quanto, /root/reference) is copied to a scratch directory
first because importing it writes __pycache__ and JIT-build artefacts into its own
tree.  Every tensor is produced by reference code paths:

* packing:       optimum.quanto.tensor.packed.pack_weights / torch.ops.quanto.unpack
* quantization:  AbsmaxOptimizer / MaxOptimizer + quantize_weight(..., optimized=False)
* dequantize:    QBytesTensor.dequantize / QBitsTensor.dequantize
* forward:       torch.nn.functional.linear(x, qweight[, bias])  (QLinear.forward is exactly this,
                 nn/qlinear.py:49-50) and torch.ops.quanto.qbytes_mm

Low-precision tensors are stored as float32 (values exactly representable in the
source dtype); fp8 tensors as their uint8 codes.  Seeds are fixed; re-running
reproduces the files bit for bit on the same torch build.
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("QUANTO_REFERENCE", "/root/reference")


def _import_reference():
    scratch = tempfile.mkdtemp(prefix="quanto_ref_")
    dst = os.path.join(scratch, "ref")
    shutil.copytree(REF, dst, ignore=shutil.ignore_patterns(".git", "*.png"))
    sys.path.insert(0, dst)
    import optimum.quanto  # noqa: F401

    return scratch


def f32(t):
    import torch

    if t.dtype in (torch.float8_e4m3fn, torch.float8_e4m3fnuz, torch.float8_e5m2):
        return t.view(torch.uint8).numpy().copy()
    if t.dtype.is_floating_point:
        return t.to(torch.float32).numpy().copy()
    return t.numpy().copy()


DT = {"fp32": "float32", "fp16": "float16", "bf16": "bfloat16"}


def main():
    scratch = _import_reference()
    import torch
    from optimum.quanto import (
        AbsmaxOptimizer,
        MaxOptimizer,
        qfloat8_e4m3fn,
        qfloat8_e4m3fnuz,
        qfloat8_e5m2,
        qint2,
        qint4,
        qint8,
        quantize_weight,
    )
    from optimum.quanto.tensor.packed import pack_weights

    out = {}

    # ---- 1. pack / unpack (tests/library/test_unpack.py:22-30, tests/tensor/test_packed_tensor.py:24-36)
    g = torch.Generator().manual_seed(1234)
    for bits in (2, 4):
        for shape in [(10,), (12,), (10, 10), (12, 10), (32, 32), (7, 5), (1, 3), (256, 128)]:
            a = torch.randint(0, 2**bits, shape, dtype=torch.uint8, generator=g)
            p = pack_weights(a, bits)
            u = torch.ops.quanto.unpack(p, bits)
            key = f"pack/b{bits}/" + "x".join(map(str, shape))
            out[key + "/a"] = f32(a)
            out[key + "/packed"] = f32(p)
            out[key + "/unpacked"] = f32(u)

    # ---- 2. integer ramp known-answer test (tests/library/test_quantize.py:105-119)
    for name, tdt in (("fp32", torch.float32), ("fp16", torch.float16)):
        for qt in (qint2, qint4):
            bits = qt.bits
            qmin, qmax = -(2 ** (bits - 1)), 2 ** (bits - 1) - 1
            a = torch.tensor(range(qmin, qmax + 1), dtype=tdt)
            scale, shift = MaxOptimizer()(a, qtype=qt, axis=0, group_size=None)
            zp = torch.round(shift / scale)
            data = torch.ops.quanto.quantize_affine(a, bits, 0, None, scale, zp)
            key = f"ramp/{name}/b{bits}"
            out[key + "/a"] = f32(a)
            out[key + "/scale"] = f32(scale)
            out[key + "/shift"] = f32(shift)
            out[key + "/zeropoint"] = f32(zp)
            out[key + "/data"] = f32(data)

    # ---- 3. qbits weights: quantize -> pack -> dequantize -> linear
    def qbits_case(tag, N, K, qt, dtname, group_size, Ms, zeropoint=False, bias=False, seed=0, wscale=1.0):
        tdt = getattr(torch, DT[dtname])
        gg = torch.Generator().manual_seed(seed)
        w = ((torch.rand((N, K), generator=gg) * 2 - 1) * wscale).to(tdt)
        scale, shift = MaxOptimizer()(w, qtype=qt, axis=0, group_size=group_size, zeropoint=zeropoint)
        qw = quantize_weight(w, qtype=qt, axis=0, scale=scale, shift=shift, group_size=group_size, optimized=False)
        key = f"qbits/{tag}"
        out[key + "/w"] = f32(w)
        out[key + "/scale"] = f32(qw._scale)
        out[key + "/shift"] = f32(qw._shift)
        out[key + "/packed"] = f32(qw._data._data)
        out[key + "/unpacked"] = f32(qw._data.unpack())
        out[key + "/dequantized"] = f32(qw.dequantize())
        out[key + "/meta"] = np.array([N, K, qt.bits, group_size or 0, int(zeropoint)], dtype=np.int64)
        b = None
        if bias:
            b = (torch.rand((N,), generator=gg) * 2 - 1).to(tdt)
            out[key + "/bias"] = f32(b)
        for M in Ms:
            x = torch.randn((M, K), generator=gg).to(tdt)
            with torch.no_grad():
                y = torch.nn.functional.linear(x, qw, b)
            out[key + f"/x{M}"] = f32(x)
            out[key + f"/y{M}"] = f32(y)

    for dtname in ("fp32", "fp16", "bf16"):
        qbits_case(f"int4_g128_{dtname}", 64, 256, qint4, dtname, 128, (1, 5, 32), seed=10)
    qbits_case("int4_g128_bf16_bias", 96, 384, qint4, "bf16", 128, (1, 3), bias=True, seed=11)
    qbits_case("int4_g128_fp16_zp", 64, 256, qint4, "fp16", 128, (1, 4), zeropoint=True, seed=12)
    qbits_case("int4_g64_fp32", 48, 192, qint4, "fp32", 64, (2,), seed=13)
    qbits_case("int4_perchannel_fp32", 33, 40, qint4, "fp32", None, (3,), seed=14)
    qbits_case("int4_oddrows_fp32", 5, 128, qint4, "fp32", 128, (2,), seed=15)
    qbits_case("int2_g128_fp32", 64, 256, qint2, "fp32", 128, (1, 4), seed=16)
    qbits_case("int2_g128_bf16", 64, 256, qint2, "bf16", 128, (2,), seed=17)
    qbits_case("int4_g128_bf16_small_w", 128, 512, qint4, "bf16", 128, (1, 17), seed=18, wscale=0.02)

    # ---- 4. qbytes weights (int8 / fp8): quantize -> dequantize -> linear, and the raw qbytes_mm op
    def qbytes_case(tag, N, K, qt, dtname, Ms, bias=False, seed=0, wscale=1.0, store_w=True):
        tdt = getattr(torch, DT[dtname])
        gg = torch.Generator().manual_seed(seed)
        w = ((torch.rand((N, K), generator=gg) * 2 - 1) * wscale).to(tdt)
        scale = AbsmaxOptimizer()(w, qtype=qt, axis=0)
        qw = quantize_weight(w, qtype=qt, axis=0, scale=scale, optimized=False)
        key = f"qbytes/{tag}"
        if store_w:
            out[key + "/w"] = f32(w)
        out[key + "/scale"] = f32(qw._scale)
        out[key + "/data"] = f32(qw._data)
        if N * K <= 1 << 16:
            out[key + "/dequantized"] = f32(qw.dequantize())
        out[key + "/meta"] = np.array([N, K], dtype=np.int64)
        b = None
        if bias:
            b = (torch.rand((N,), generator=gg) * 2 - 1).to(tdt)
            out[key + "/bias"] = f32(b)
        for M in Ms:
            x = torch.randn((M, K), generator=gg).to(tdt)
            with torch.no_grad():
                y = torch.nn.functional.linear(x, qw, b)
                y_op = torch.ops.quanto.qbytes_mm(x, qw._data, qw._scale)
            out[key + f"/x{M}"] = f32(x)
            out[key + f"/y{M}"] = f32(y)
            out[key + f"/yop{M}"] = f32(y_op)

    for dtname in ("fp32", "fp16", "bf16"):
        qbytes_case(f"int8_{dtname}", 48, 64, qint8, dtname, (1, 5, 32), seed=20)
        qbytes_case(f"e4m3fn_{dtname}", 48, 64, qfloat8_e4m3fn, dtname, (1, 5, 32), seed=21)
    qbytes_case("int8_bf16_bias", 50, 50, qint8, "bf16", (1, 10), bias=True, seed=22)
    qbytes_case("e4m3fnuz_fp16", 48, 64, qfloat8_e4m3fnuz, "fp16", (2,), seed=23)
    qbytes_case("e5m2_fp16", 48, 64, qfloat8_e5m2, "fp16", (2,), seed=24)
    # BASELINE.json configs[0]: QLinear int8 weights, fp32 activations, (M,K,N)=(1,1024,1024), reference CPU path
    qbytes_case("cfg1_int8_fp32_1x1024x1024", 1024, 1024, qint8, "fp32", (1,), seed=25, wscale=1.0 / 32, store_w=False)

    # ---- 5. int8 x int8 (quantized activations) qbytes_mm: library/qbytes_mm.py:36-50
    gg = torch.Generator().manual_seed(30)
    a = torch.randint(-127, 127, (32, 64), dtype=torch.int8, generator=gg)
    b = torch.randint(-127, 127, (48, 64), dtype=torch.int8, generator=gg)
    for dtname in ("fp16", "bf16", "fp32"):
        s = ((torch.rand((48, 1), generator=gg) * 2 - 1) / 1e3).to(getattr(torch, DT[dtname]))
        y = torch.ops.quanto.qbytes_mm(a, b, s)
        out[f"qbytes_i8i8/{dtname}/a"] = f32(a)
        out[f"qbytes_i8i8/{dtname}/b"] = f32(b)
        out[f"qbytes_i8i8/{dtname}/scales"] = f32(s)
        out[f"qbytes_i8i8/{dtname}/y"] = f32(y)

    # ---- 6. quantized activations (tensor/activations/quantization.py:24-39, calibrate.py:38-64) and the
    #         F.linear(qinput, qweight) dispatch they feed (tensor/weights/qbytes.py:68-82; tests/tensor/ops/test_linear_dispatch.py:22-42)
    from optimum.quanto import Calibration, QLinear, absmax_scale, freeze, quantize_activation

    for qname, qt in (("int8", qint8), ("e4m3fn", qfloat8_e4m3fn), ("e5m2", qfloat8_e5m2)):
        for dtname in ("fp32", "fp16", "bf16"):
            tdt = getattr(torch, DT[dtname])
            gg = torch.Generator().manual_seed(40)
            x = (torch.randn((4, 24, 64), generator=gg) * 3).to(tdt)
            scale = absmax_scale(x, qt)
            qx = quantize_activation(x, qt, scale)
            key = f"qact/{qname}_{dtname}"
            out[key + "/x"] = f32(x)
            out[key + "/scale"] = f32(scale)
            out[key + "/data"] = f32(qx._data)
            out[key + "/dequantized"] = f32(qx.dequantize())

    for tag, aqt, wqt in (("a8w8_int8", qint8, qint8), ("a8w8_e4m3fn", qfloat8_e4m3fn, qfloat8_e4m3fn), ("aint8_we4m3fn", qint8, qfloat8_e4m3fn)):
        for dtname in ("fp32", "fp16", "bf16"):
            tdt = getattr(torch, DT[dtname])
            gg = torch.Generator().manual_seed(41)
            N, K = 192, 128
            w = (torch.rand((N, K), generator=gg) * 2 - 1).to(tdt)
            bias = (torch.rand((N,), generator=gg) * 2 - 1).to(tdt)
            x = torch.randn((2, 40, K), generator=gg).to(tdt)
            wscale = AbsmaxOptimizer()(w, qtype=wqt, axis=0)
            qw = quantize_weight(w, qtype=wqt, axis=0, scale=wscale, activation_qtype=aqt, optimized=False)
            xscale = absmax_scale(x, aqt)
            qx = quantize_activation(x, aqt, xscale)
            with torch.no_grad():
                y = torch.nn.functional.linear(qx, qw, bias)
                y_nobias = torch.nn.functional.linear(qx, qw)
            key = f"qact_linear/{tag}_{dtname}"
            out[key + "/w"] = f32(w)
            out[key + "/bias"] = f32(bias)
            out[key + "/x"] = f32(x)
            out[key + "/wdata"] = f32(qw._data)
            out[key + "/wscale"] = f32(qw._scale)
            out[key + "/xdata"] = f32(qx._data)
            out[key + "/xscale"] = f32(qx._scale)
            out[key + "/y"] = f32(y)
            out[key + "/y_nobias"] = f32(y_nobias)

    # QLinear(weights=qint8, activations=qint8): calibrate on two batches, freeze, run (nn/qmodule.py:131-134,281-299)
    for dtname in ("fp32", "bf16"):
        tdt = getattr(torch, DT[dtname])
        torch.manual_seed(42)
        lin = torch.nn.Linear(128, 192).to(tdt)
        q = QLinear.from_module(lin, weights=qint8, activations=qint8)
        gg = torch.Generator().manual_seed(43)
        batches = [torch.randn((3, 20, 128), generator=gg).to(tdt) for _ in range(2)]
        with torch.no_grad(), Calibration():
            for b in batches:
                q(b)
        freeze(q)
        with torch.no_grad():
            yq = q(batches[0])
        key = f"qlinear_a8w8/{dtname}"
        out[key + "/w"] = f32(lin.weight.detach())
        out[key + "/bias"] = f32(lin.bias.detach())
        out[key + "/x0"] = f32(batches[0])
        out[key + "/x1"] = f32(batches[1])
        out[key + "/input_scale"] = f32(q.input_scale)
        out[key + "/output_scale"] = f32(q.output_scale)
        out[key + "/y_data"] = f32(yq._data)
        out[key + "/y_scale"] = f32(yq._scale)

    # ---- 7. QConv2d (nn/qconv2d.py:26-55; tests/nn/test_qconv2d.py): quantize -> freeze -> forward.  in_features = C*kh*kw:
    # 144 -> per-channel int4 (no group size divides it), 288 -> groups of 96
    from optimum.quanto import QConv2d

    for dtname in ("fp32", "bf16"):
        tdt = getattr(torch, DT[dtname])
        for tag, wq in (("int8", qint8), ("int4", qint4), ("e4m3fn", qfloat8_e4m3fn)):
            for cname, (cin, cout, ksz, stride, pad) in (("c16k3", (16, 32, 3, 1, 1)), ("c32k3s2", (32, 24, 3, 2, 0)),
                                                          ("c64k1", (64, 48, 1, 1, 0))):
                torch.manual_seed(77 + cin)
                conv = torch.nn.Conv2d(cin, cout, ksz, stride=stride, padding=pad).to(tdt)
                q = QConv2d.from_module(conv, weights=wq)
                freeze(q)
                gg = torch.Generator().manual_seed(78 + cout)
                x = torch.randn((2, cin, 12, 10), generator=gg).to(tdt)
                with torch.no_grad():
                    y = q(x)
                qw = q.weight.detach()
                key = f"qconv2d/{tag}_{cname}_{dtname}"
                out[key + "/w"] = f32(conv.weight.detach())
                out[key + "/bias"] = f32(conv.bias.detach())
                out[key + "/x"] = f32(x)
                out[key + "/y"] = f32(y)
                out[key + "/wscale"] = f32(qw._scale)
                if wq is qint4:
                    out[key + "/wpacked"] = f32(qw._data._data)
                    out[key + "/wshift"] = f32(qw._shift)
                    out[key + "/group_size"] = np.array(-1 if qw._group_size is None else qw._group_size)
                else:
                    out[key + "/wdata"] = f32(qw._data)
                out[key + "/wdq"] = f32(qw.dequantize())

    path = os.path.join(HERE, "quanto_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")
    with open(os.path.join(HERE, "PROVENANCE.txt"), "w") as f:
        f.write(
            "quanto_golden.npz generated by tests/golden/make_golden.py\n"
            f"reference: huggingface/optimum-quanto @ /root/reference (version {__import__('optimum.quanto').quanto.__version__})\n"
            f"torch: {torch.__version__}\nnumpy: {np.__version__}\n"
        )
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
