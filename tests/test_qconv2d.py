"""QConv2d: the host mirror against the reference's golden vectors (tests/golden section 7, produced
by the real QConv2d: quantize -> freeze -> forward), and on the GPU the im2col + fused-GEMM lowering
(tensor/weights.py conv2d_as_gemm) against exact math on the reference's integers and against the reference's outputs with
the tolerance the reference's own test uses (tests/nn/test_qconv2d.py: assert_similar, atol 1e-2 class)."""
import io
import os

import numpy as np
import pytest
import torch

import optimum_quanto_amd as Q
from optimum_quanto_amd.library.hip import quanto_hip
from optimum_quanto_amd.tensor.weights import conv2d_patches

from oracle import quanto_oracle as O

from helpers import TORCH_DT, assert_close_to_exact, assert_close_with_bias, assert_similar, observed_activation_scales, to_numpy, to_torch

QTYPES = {"int8": "qint8", "int4": "qint4", "e4m3fn": "qfloat8_e4m3fn"}
CONVS = {"c16k3": (16, 32, 3, 1, 1), "c32k3s2": (32, 24, 3, 2, 0), "c64k1": (64, 48, 1, 1, 0)}
CASES = [(t, c, d) for t in QTYPES for c in CONVS for d in ("fp32", "bf16")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(golden, tag, cname, dt, device="cpu"):
    key = f"qconv2d/{tag}_{cname}_{dt}"
    cin, cout, ksz, stride, pad = CONVS[cname]
    conv = torch.nn.Conv2d(cin, cout, ksz, stride=stride, padding=pad).to(TORCH_DT[dt])
    with torch.no_grad():
        conv.weight.copy_(to_torch(golden[key + "/w"], dt))
        conv.bias.copy_(to_torch(golden[key + "/bias"], dt))
    # quantize on the CPU (bit-identical to the reference; a device computes absmax / 127 through a reciprocal and may land
    # one ulp away), then move the frozen module: what a user does when loading a quantized checkpoint onto the GPU
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, QTYPES[tag]))
    Q.freeze(q)
    return key, q.to(device), to_torch(golden[key + "/x"], dt, device)


def _check_inner_tensors(golden, key, tag, qw):
    np.testing.assert_array_equal(to_numpy(qw._scale), golden[key + "/wscale"])
    if tag == "int4":
        gs = int(golden[key + "/group_size"])
        assert qw._group_size == (None if gs < 0 else gs)
        np.testing.assert_array_equal(to_numpy(qw._data._data), golden[key + "/wpacked"])
        np.testing.assert_array_equal(to_numpy(qw._shift), golden[key + "/wshift"])
    else:
        np.testing.assert_array_equal(to_numpy(qw._data), golden[key + "/wdata"])
    np.testing.assert_array_equal(to_numpy(qw.dequantize()), golden[key + "/wdq"])


def _row_form(cin, k, s, d, w_in, p, wq="qint8"):
    """True where the convolution takes the ROW form (r5): three taps wide, dilation 1 along the width, W >= 4, KH * 3 <= 31, cin KH a multiple
    of 8 (csrc/qconv_mfma.hip::rows_eligible)."""
    kh, kw = (k, k) if isinstance(k, int) else k
    sw = s if isinstance(s, int) else s[1]
    dw = d if isinstance(d, int) else d[1]
    pw = p if isinstance(p, int) else p[1]
    ow = (w_in + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    del ow, sw  # (stride 1 with an even OW: pixel pairs; any other stride / an odd OW: one pixel per thread - the row form either way)
    return kw == 3 and dw == 1 and w_in >= 4 and kh * kw <= 31 and (cin * kh) % 8 == 0


def _conv_kernel_name(cin, k, s, d, w_in, p, wq="qint8", pixels=None, dense_min_tiles=8):
    """8-bit weights: the row form or the tap gather; int4 / int2: dequantize once + the row form on the dense weight - from `dense_min_tiles`
    128-pixel tiles on (`pixels` = B OH OW of the call; QUANTO_HIP_CONV_DENSE_MIN_TILES) -, or the tap gather that dequantizes while staging
    (csrc/c_api.hip: quanto_hip_qbits_conv2d)."""
    rows = _row_form(cin, k, s, d, w_in, p, wq)
    if wq in ("qint4", "qint2"):
        dense = rows and pixels is not None and (pixels + 127) // 128 >= dense_min_tiles
        return ("conv2d_rows_dequant_" if dense else "conv2d_mfma_") + wq[1:]
    return "conv2d_mfma_rows" if rows else "conv2d_mfma"


def _oracle_dequantized(w, dt):
    """The dense weight the reference would materialise for the packed int4 weight ``w`` - computed by the numpy oracle on the HOST from the
    packed bytes, scale and shift, so that the convolution gate does not lean on this library's own device dequantize kernel."""
    from oracle import quanto_oracle as O

    packed = w._data._data.cpu().numpy()
    shift = w._shift.cpu()
    shift = shift.numpy() if shift.dtype == torch.uint8 else to_numpy(shift)
    k = w.numel() // w.shape[0]
    dq = O.dequantize_qbits_ref(packed, w._data.bits, to_numpy(w._scale.cpu()), shift, 0, w._group_size, (w.shape[0], k), dt)
    return torch.from_numpy(np.ascontiguousarray(dq, dtype=np.float64)).reshape(tuple(w.shape))


@pytest.mark.parametrize("shape", [(16, 32, 3, 1, 1, 1), (8, 24, 3, 2, 0, 1), (4, 8, (3, 5), (2, 1), (1, 2), (1, 2)), (16, 8, 1, 1, 0, 1),
                                   (3, 5, 2, 3, 2, 1)])
def test_conv2d_patches_times_flat_weight_is_conv2d(shape):
    """im2col rows are in the weight's (c, i, j) order: patches @ W.view(N, -1).T, folded back, equals F.conv2d."""
    cin, cout, k, s, p, d = shape
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=p, dilation=d).double()
    x = torch.randn(2, cin, 13, 11, dtype=torch.float64)
    a, oh, ow = conv2d_patches(x, conv.kernel_size, conv.stride, conv.padding, conv.dilation)
    y = (a @ conv.weight.reshape(cout, -1).t() + conv.bias).view(2, oh * ow, cout).permute(0, 2, 1).reshape(2, cout, oh, ow)
    torch.testing.assert_close(y, conv(x), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("tag,cname,dt", CASES)
def test_qconv2d_matches_reference_golden_cpu(golden, tag, cname, dt):
    """Quantized weight bit-identical to the reference's; CPU forward (reference behaviour: dequantize + float convolution)
    equal to the reference's output."""
    key, q, x = _build(golden, tag, cname, dt)
    assert isinstance(q, Q.QConv2d) and q.frozen
    _check_inner_tensors(golden, key, tag, q.weight)
    with torch.no_grad():
        y = q(x)
    assert y.dtype == TORCH_DT[dt]
    np.testing.assert_allclose(to_numpy(y), golden[key + "/y"], rtol=0, atol=0 if dt == "fp32" else 1e-6)


def test_qconv2d_quantize_freeze_state_dict_roundtrip():
    torch.manual_seed(5)
    model = torch.nn.Sequential(torch.nn.Conv2d(16, 32, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(32, 8, 3, stride=2))
    x = torch.randn(2, 16, 12, 12)
    ref = model(x)
    Q.quantize(model, weights=Q.qint4)
    assert isinstance(model[0], Q.QConv2d) and model[0].weight_group_size is None and model[2].weight_group_size == 96
    y_dyn = model(x)
    Q.freeze(model)
    y = model(x)
    torch.testing.assert_close(y, y_dyn, rtol=0, atol=0)
    assert (y - ref).abs().max() < 0.1 * ref.abs().max()
    buf = io.BytesIO()
    torch.save(model.state_dict(), buf)
    buf.seek(0)
    fresh = torch.nn.Sequential(torch.nn.Conv2d(16, 32, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(32, 8, 3, stride=2))
    Q.quantize(fresh, weights=Q.qint4)
    Q.freeze(fresh)
    fresh.load_state_dict(torch.load(buf, weights_only=False))
    torch.testing.assert_close(fresh(x), y, rtol=0, atol=0)


@pytest.mark.parametrize("wq,gs,zp", [("qint8", None, False), ("qint4", 96, False), ("qint4", 64, True), ("qint4", None, False)])
def test_conv2d_ops_default_implementations_are_the_reference_behaviour(wq, gs, zp):
    """quanto::qbytes_conv2d / quanto::qbits_conv2d off the device (CompositeExplicitAutograd): dequantize the weight, float convolution -
    bit for bit what the reference's QConv2d.forward computes through qfallback (nn/qconv2d.py:54-55)."""
    torch.manual_seed(11)
    conv = torch.nn.Conv2d(64, 96, 3, stride=2, padding=1).to(torch.bfloat16)
    x = torch.randn(2, 64, 9, 7).to(torch.bfloat16)
    if wq == "qint8":
        q = Q.QConv2d.from_module(conv, weights=Q.qint8)
        Q.freeze(q)
        w = q.weight
        y = torch.ops.quanto.qbytes_conv2d(x, w._data, w._scale, conv.bias, [2, 2], [1, 1], [1, 1])
    else:
        scale, shift = Q.MaxOptimizer()(conv.weight.detach(), Q.qint4, 0, gs, zeropoint=zp)
        w = Q.quantize_weight(conv.weight.detach(), Q.qint4, 0, scale, shift, group_size=gs, optimized=False)
        y = torch.ops.quanto.qbits_conv2d(x, w._data._data, w._scale, w._shift, conv.bias, 4, gs, list(w.shape), [2, 2], [1, 1], [1, 1])
    want = torch.nn.functional.conv2d(x, w.dequantize(), conv.bias, 2, 1)
    assert y.dtype == torch.bfloat16 and torch.equal(y, want)


# ------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag,cname,dt", CASES)
def test_qconv2d_fused_gemm_gpu(golden, tag, cname, dt):
    """On the device the convolution is im2col + quanto::qbytes_mm / quanto::qbits_mm.  Gate: exact float64 convolution
    with the reference's dequantized weight (the parity gate of the Linear kernels), and the reference's own output with
    the reference test's similarity criterion."""
    dev = torch.device("cuda")
    key, q, x = _build(golden, tag, cname, dt, dev)
    _check_inner_tensors(golden, key, tag, q.weight)
    with torch.no_grad():
        y = q(x)
    kernel = quanto_hip.lib.last_kernel()
    assert kernel in ("gemv", "skinny", "mfma", "mfma_large", "dequant_mfma", "naive", "conv2d_mfma", "conv2d_mfma_rows", "conv2d_mfma_int4", "conv2d_rows_dequant_int4",
                      "mfma_f32", "gemv_f32"), kernel  # (r6: fp32 convolutions lowered to im2col reach the fp32 kernels of csrc/qmm_f32.hip)
    assert y.is_cuda and y.dtype == TORCH_DT[dt] and tuple(y.shape) == golden[key + "/y"].shape
    # exact math on the reference's integers: float64 convolution, product rounded, bias added, rounded again
    cin, cout, ksz, stride, pad = CONVS[cname]
    x64 = torch.from_numpy(golden[key + "/x"]).double()
    a64, _, _ = conv2d_patches(x64, (ksz, ksz), stride, pad, 1)
    a64 = a64.numpy()
    if tag == "int4" and kernel not in ("dequant_mfma", "conv2d_mfma_int4", "conv2d_rows_dequant_int4"):
        # the fused int4 kernels use q * scale - shift unrounded (exact math on the reference's integers and scales)
        gs = int(golden[key + "/group_size"])
        w64 = O.dequantize_qbits_exact(golden[key + "/wpacked"], 4, golden[key + "/wscale"], golden[key + "/wshift"], 0,
                                       None if gs < 0 else gs, (cout, cin * ksz * ksz))
        prod = a64 @ np.asarray(w64, np.float64).reshape(cout, -1).T
    elif tag == "int4":
        # the flat path multiplies by the weight rounded exactly as the reference rounds it, which is what wdq holds
        prod = a64 @ golden[key + "/wdq"].astype(np.float64).reshape(cout, -1).T
    else:
        # 8-bit: exact integer / fp8 products, the per-channel scale applied to the accumulator (library/qbytes_mm.py:25-33)
        prod = O.qbytes_mm_exact(a64, golden[key + "/wdata"].reshape(cout, -1), golden[key + "/wscale"].reshape(cout, 1),
                                 "e4m3fn" if tag == "e4m3fn" else None)
    bias = golden[key + "/bias"]
    yn = to_numpy(y).transpose(0, 2, 3, 1).reshape(-1, cout)
    if dt == "fp32":  # fp32 accumulation order is visible at fp32 output precision: the north-star 1e-3 gate (expect ~1e-6)
        assert_close_to_exact(yn, prod + bias.astype(np.float64), dt, f"qconv2d {key} ({kernel})")
    else:
        assert_close_with_bias(yn, prod, bias, dt, f"qconv2d {key} ({kernel})")
    assert_similar(torch.from_numpy(golden[key + "/y"]), y.float().cpu(), atol=1e-2 if dt == "bf16" else 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("wq", ["qint8", "qfloat8_e4m3fn", "qfloat8_e5m2"])
@pytest.mark.parametrize("cin,cout,k,s,p,d", [(64, 96, 3, 1, 1, 1), (64, 40, 3, 2, 0, 1), (128, 64, (3, 5), (2, 1), (1, 2), (1, 2)), (64, 32, 1, 2, 0, 1),
                                              (192, 130, 3, 1, 2, 2), (128, 72, 1, 1, 0, 1), (64, 96, 1, 1, 0, 1)])  # (the last: pointwise with ONE K-tile - the kernel's since r5)
def test_qconv2d_implicit_gemm_gpu(dt, wq, cin, cout, k, s, p, d):
    """quanto::qbytes_conv2d (r4): the convolution as an IMPLICIT GEMM - the im2col operand is gathered inside the kernel's staging loads,
    the output is written NCHW.  Strides, paddings, dilations, rectangular windows, ragged M (3 x 13 x 11 pixels) and ragged output channels;
    gate: float64 convolution on the stored integers / fp8 values, per-channel scale on the accumulator, the reference's bias order."""
    torch.manual_seed(cin + cout)
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=p, dilation=d).to(TORCH_DT[dt])
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, wq))
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(3, cin, 13, 11).to(TORCH_DT[dt])
    with torch.no_grad():
        y = q(x.cuda())
        assert quanto_hip.lib.last_kernel() == _conv_kernel_name(cin, k, s, d, 11, p, wq)
        w64 = q.weight._data.cpu().double() if wq == "qint8" else q.weight._data.cpu().float().double()
        prod = torch.nn.functional.conv2d(x.double(), w64, None, conv.stride, conv.padding, conv.dilation)
        prod = prod * q.weight._scale.cpu().double().reshape(1, -1, 1, 1)
    assert y.shape == prod.shape and y.dtype == TORCH_DT[dt]
    bias = to_numpy(q.bias).astype(np.float64).reshape(1, -1, 1, 1)
    assert_close_with_bias(to_numpy(y), prod.numpy(), np.broadcast_to(bias, prod.shape), dt, f"implicit conv {cin}->{cout} k{k}")
    # and without a bias: the plain exact-math gate
    q.bias = None
    with torch.no_grad():
        y0 = q(x.cuda())
    assert_close_to_exact(to_numpy(y0), prod.numpy(), dt, f"implicit conv {cin}->{cout} k{k}, no bias")


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("wq", ["qint8", "qfloat8_e4m3fn", "qfloat8_e5m2"])
@pytest.mark.parametrize("c,mult,k,s,p,d,hw", [(32, 1, 3, 1, 1, 1, (14, 14)), (48, 1, 3, 2, 1, 1, (15, 13)), (24, 2, 5, 1, 2, 1, (12, 17)), (16, 1, 7, 2, 3, 1, (20, 9)),
                                                (40, 1, (3, 5), (2, 1), (1, 2), (1, 2), (13, 11)), (8, 3, 3, 1, 0, 2, (11, 10)), (64, 1, 1, 1, 0, 1, (7, 7)),
                                                (20, 1, (2, 4), 1, (1, 0), 1, (9, 6)), (96, 1, 3, 1, 1, 1, (56, 56))])
def test_qconv2d_depthwise_gpu(dt, wq, c, mult, k, s, p, d, hw):
    _depthwise_case(dt, wq, c, mult, k, s, p, d, hw, ("conv2d_depthwise", "conv2d_depthwise_strip"))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("wq", ["qint8", "qfloat8_e4m3fn", "qfloat8_e5m2"])
@pytest.mark.parametrize("c,mult,k,s,p,hw", [(16, 1, 3, 1, 1, (9, 16)), (8, 2, 3, 2, 1, (13, 24)), (12, 1, 5, 1, 2, (10, 32)), (8, 1, 5, 2, 2, (11, 40)),
                                              (8, 1, 3, 1, 0, (7, 16)), (8, 3, 3, 2, 0, (9, 24)), (4, 1, 3, 1, 1, (5, 8)), (6, 1, 3, 2, 1, (56, 56)),
                                              (5, 1, 5, 1, 2, (3, 8)), (3, 1, 3, 1, 1, (1, 64)),
                                              (10, 1, 3, 1, 1, (28, 28)), (6, 1, 5, 1, 2, (9, 12)), (6, 2, 3, 2, 1, (11, 20)), (4, 1, 5, 2, 2, (8, 4)), (4, 1, 3, 1, 0, (6, 12))])
def test_qconv2d_depthwise_strip_gpu(dt, wq, c, mult, k, s, p, hw):
    """r6: the strip form of the depthwise kernel (rows of whole 16-byte chunks: W % 8 == 0; 3 x 3 / 5 x 5 windows, stride 1 or 2, "same" padding or none) - a
    single chunk per row (both neighbours out of range), ragged groups of output rows, output widths that are not a multiple of the eight columns a thread owns
    (stride 2), channel multipliers, planes shorter than the window."""
    _depthwise_case(dt, wq, c, mult, k, s, p, 1, hw, ("conv2d_depthwise_strip",))


def _depthwise_case(dt, wq, c, mult, k, s, p, d, hw, kernels):
    """r6: depthwise layers (groups = in_channels; channel multipliers 1, 2, 3) with an int8 / fp8 weight on the stencil kernel of csrc/qconv_depthwise.hip -
    the register windows (3, 5, 7), rectangular and even windows (the generic tap loop), strides, paddings, dilations, widths that are not a multiple of the four
    columns a thread owns, a 56 x 56 MobileNet plane.  Gate: float64 grouped convolution on the stored integers / fp8 values, per-channel scale on the sum, the
    reference's bias order (nn/qconv2d.py:54-55 dequantizes the weight and calls the float convolution)."""
    torch.manual_seed(c + mult)
    conv = torch.nn.Conv2d(c, c * mult, k, stride=s, padding=p, dilation=d, groups=c).to(TORCH_DT[dt])
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, wq))
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(3, c, *hw).to(TORCH_DT[dt])
    with torch.no_grad():
        y = q(x.cuda())
        assert quanto_hip.lib.last_kernel() in kernels
        w64 = q.weight._data.cpu().double() if wq == "qint8" else q.weight._data.cpu().float().double()
        prod = torch.nn.functional.conv2d(x.double(), w64, None, conv.stride, conv.padding, conv.dilation, groups=c)
        prod = prod * q.weight._scale.cpu().double().reshape(1, -1, 1, 1)
    assert y.shape == prod.shape and y.dtype == TORCH_DT[dt]
    bias = to_numpy(q.bias).astype(np.float64).reshape(1, -1, 1, 1)
    assert_close_with_bias(to_numpy(y), prod.numpy(), np.broadcast_to(bias, prod.shape), dt, f"depthwise conv {c}x{mult} k{k}")
    q.bias = None
    with torch.no_grad():
        y0 = q(x.cuda())
    assert_close_to_exact(to_numpy(y0), prod.numpy(), dt, f"depthwise conv {c}x{mult} k{k}, no bias")
    # the reference's own tolerance against its dequantize-first result (tests/nn/test_qconv2d.py: assert_similar)
    with torch.no_grad():
        ref = torch.nn.functional.conv2d(x.cuda(), q.weight.dequantize(), None, conv.stride, conv.padding, conv.dilation, groups=c)
    assert_similar(ref.float().cpu(), y0.float().cpu(), atol=1e-2 if dt == "bf16" else 1e-3)


@pytest.mark.gpu
def test_qconv2d_other_groupings_keep_the_reference_behaviour_gpu():
    """groups = 2 on 8 channels is not a depthwise layer: the dispatch keeps dequantize + float convolution (no library kernel runs)."""
    torch.manual_seed(1)
    conv = torch.nn.Conv2d(8, 16, 3, padding=1, groups=2).to(torch.bfloat16)
    q = Q.QConv2d.from_module(conv, weights=Q.qint8)
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(2, 8, 9, 9, dtype=torch.bfloat16, device="cuda")
    before = quanto_hip.lib.last_kernel()
    with torch.no_grad():
        y = q(x)
        ref = torch.nn.functional.conv2d(x, q.weight.dequantize(), q.bias, 1, 1, 1, 2)
    assert quanto_hip.lib.last_kernel() == before
    torch.testing.assert_close(y, ref, rtol=0, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("zp", [False, True])
@pytest.mark.parametrize("cin,cout,k,s,p,d,gs", [(128, 96, 3, 1, 1, 1, 128), (64, 40, 3, 2, 0, 1, 64), (128, 64, (3, 5), (2, 1), (1, 2), (1, 2), 128),
                                                 (192, 130, 3, 1, 2, 2, 32), (64, 256, 3, 1, 1, 1, None), (32, 34, (2, 3), 1, 0, 1, 96),
                                                 (64, 96, 3, 1, 1, 1, 96), (192, 66, 1, 1, 0, 1, 64)])
def test_qconv2d_int4_implicit_gemm_gpu(dt, zp, cin, cout, k, s, p, d, gs):
    """quanto::qbits_conv2d (r4): the implicit GEMM with a packed int4 weight, dequantized while it is staged with the reference's roundings.
    Gate: float64 convolution with the weight the reference dequantizes (bit for bit `q.weight.dequantize()`), the reference's bias order;
    group sizes 128 / 96 / 64 / 32 and per-channel, float shifts and integer zero-points, ragged M and ragged output channels."""
    from optimum_quanto_amd.tensor.weights import WeightQBitsTensor

    torch.manual_seed(cin + cout)
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=p, dilation=d).to(TORCH_DT[dt])
    q = Q.QConv2d.from_module(conv, weights=Q.qint4)
    Q.freeze(q)
    # the module picks its own group size (nn/qmodule.py:121-129); the weight under test is rebuilt with the requested one / with zero-points
    scale, shift = Q.MaxOptimizer()(conv.weight.detach(), Q.qint4, 0, gs, zeropoint=zp)
    w = Q.quantize_weight(conv.weight.detach(), Q.qint4, 0, scale, shift, group_size=gs, optimized=False)
    assert isinstance(w, WeightQBitsTensor) and w._group_size == gs and (w._shift.dtype == torch.uint8) == zp
    q.weight = torch.nn.Parameter(w, requires_grad=False)
    q = q.cuda()
    x = torch.randn(3, cin, 13, 11).to(TORCH_DT[dt])
    with torch.no_grad():
        y = q(x.cuda())
        assert quanto_hip.lib.last_kernel() == _conv_kernel_name(cin, k, s, d, 11, p, "qint4", pixels=y.numel() // y.shape[1])
        wdq = _oracle_dequantized(q.weight, dt)
        assert torch.equal(q.weight.dequantize().cpu().double(), wdq)  # (and the device dequantize kernel agrees with the oracle bit for bit)
        prod = torch.nn.functional.conv2d(x.double(), wdq, None, conv.stride, conv.padding, conv.dilation)
    assert y.shape == prod.shape and y.dtype == TORCH_DT[dt]
    bias = to_numpy(q.bias).astype(np.float64).reshape(1, -1, 1, 1)
    assert_close_with_bias(to_numpy(y), prod.numpy(), np.broadcast_to(bias, prod.shape), dt, f"int4 implicit conv {cin}->{cout} k{k} g{gs}")
    q.bias = None
    with torch.no_grad():
        y0 = q(x.cuda())
    assert_close_to_exact(to_numpy(y0), prod.numpy(), dt, f"int4 implicit conv {cin}->{cout} k{k} g{gs}, no bias")


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("wq", ["qint8", "qfloat8_e4m3fn"])
@pytest.mark.parametrize("cin,cout,k,s,p,d,hw", [(3, 64, 7, 2, 3, 1, (37, 29)),      # the ResNet stem: K = 147, three K-tiles, the last one 19 wide
                                                 (3, 32, 3, 1, 1, 1, (20, 17)),      # K = 27: a single ragged K-tile
                                                 (5, 24, (3, 2), 1, 0, 1, (9, 8)),   # K = 30
                                                 (16, 48, 9, 1, 4, 1, (15, 14)),     # 81 taps: two mask words
                                                 (8, 40, 11, 4, 2, 1, (40, 33)),     # 121 taps (AlexNet's first layer window), K = 968: ragged as well
                                                 (64, 40, 8, 1, 3, 1, (12, 12)),     # exactly 64 taps
                                                 (130, 36, 3, 1, 1, 1, (7, 9))])     # K = 1170: eighteen full K-tiles and one of 18
def test_qconv2d_implicit_gemm_ragged_k_and_wide_windows_gpu(dt, wq, cin, cout, k, s, p, d, hw):
    """r5: any K = cin kh kw (a ragged last K-tile: no activation is gathered for k >= K and the weight bytes behind the row end are never
    read - fp8 garbage there could be a NaN) and windows of up to 127 taps; no ungrouped convolution is left to the materialised F.unfold."""
    torch.manual_seed(cin * 7 + cout)
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=p, dilation=d).to(TORCH_DT[dt])
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, wq))
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(2, cin, *hw).to(TORCH_DT[dt])
    with torch.no_grad():
        y = q(x.cuda())
        assert quanto_hip.lib.last_kernel() == "conv2d_mfma"
        w64 = q.weight._data.cpu().double() if wq == "qint8" else q.weight._data.cpu().float().double()
        prod = torch.nn.functional.conv2d(x.double(), w64, None, conv.stride, conv.padding, conv.dilation)
        prod = prod * q.weight._scale.cpu().double().reshape(1, -1, 1, 1)
    assert y.shape == prod.shape and y.dtype == TORCH_DT[dt] and torch.isfinite(y.float()).all()
    bias = to_numpy(q.bias).astype(np.float64).reshape(1, -1, 1, 1)
    assert_close_with_bias(to_numpy(y), prod.numpy(), np.broadcast_to(bias, prod.shape), dt, f"implicit conv {cin}->{cout} k{k}")


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("bits,zp", [(4, False), (4, True), (2, False), (2, True)])
@pytest.mark.parametrize("cin,cout,k,s,p,d,gs", [(3, 64, 7, 2, 3, 1, None),          # the stem with a sub-byte weight: K = 147, per-channel scales
                                                 (3, 32, 3, 1, 1, 1, None),          # K = 27
                                                 (20, 44, 3, 1, 1, 1, 36),           # K = 180 = 5 groups of 36: no multiple of 8 -> not eligible, falls back
                                                 (16, 48, 9, 1, 4, 1, 144),          # 81 taps, K = 1296 = 9 groups of 144
                                                 (128, 96, 3, 1, 1, 1, 128), (64, 40, 3, 2, 0, 1, 64), (192, 132, 3, 1, 2, 2, 32), (64, 256, 3, 1, 1, 1, None)])
def test_qconv2d_subbyte_implicit_gemm_ragged_k_and_qint2_gpu(dt, bits, zp, cin, cout, k, s, p, d, gs):
    """r5: qint2 weights (four planes per packed byte) and ragged K / wide windows on the sub-byte implicit GEMM.  Gate: float64 convolution with
    the weight the numpy oracle dequantizes on the host from the packed bytes (the reference's roundings)."""
    from optimum_quanto_amd.tensor.weights import WeightQBitsTensor

    qt = Q.qint4 if bits == 4 else Q.qint2
    torch.manual_seed(cin + cout + bits)
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=p, dilation=d).to(TORCH_DT[dt])
    q = Q.QConv2d.from_module(conv, weights=qt)
    Q.freeze(q)
    scale, shift = Q.MaxOptimizer()(conv.weight.detach(), qt, 0, gs, zeropoint=zp)
    w = Q.quantize_weight(conv.weight.detach(), qt, 0, scale, shift, group_size=gs, optimized=False)
    assert isinstance(w, WeightQBitsTensor) and w._group_size == gs
    q.weight = torch.nn.Parameter(w, requires_grad=False)
    q = q.cuda()
    x = torch.randn(2, cin, 15, 13).to(TORCH_DT[dt])
    with torch.no_grad():
        y = q(x.cuda())
        eligible = gs is None or gs % 8 == 0
        assert (quanto_hip.lib.last_kernel() == _conv_kernel_name(cin, k, s, d, 13, p, f"qint{bits}", pixels=y.numel() // y.shape[1])) == eligible, quanto_hip.lib.last_kernel()
        prod = torch.nn.functional.conv2d(x.double(), _oracle_dequantized(q.weight, dt), None, conv.stride, conv.padding, conv.dilation)
    assert y.shape == prod.shape and y.dtype == TORCH_DT[dt]
    bias = to_numpy(q.bias).astype(np.float64).reshape(1, -1, 1, 1)
    if eligible:
        assert_close_with_bias(to_numpy(y), prod.numpy(), np.broadcast_to(bias, prod.shape), dt, f"int{bits} implicit conv {cin}->{cout} k{k} g{gs}")
    else:  # the fused GEMM kernels fold scale / shift in fp32 instead of rounding the weight first: the reference-similarity gate
        assert_similar(torch.from_numpy(prod.numpy() + bias).float(), y.float().cpu(), atol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("wq", ["qint8", "qint4"])
@pytest.mark.parametrize("split", [1, 2, 7])
def test_qconv2d_implicit_gemm_k_split_gpu(monkeypatch, wq, split):
    """The convolution kernel's K split (blockIdx.z + a reduce kernel that adds the fp32 partial tiles in split order): forced to 1 (no
    workspace use), 2 and 7 (ragged: 18 K-tiles), every result against the float64 gate; and a call WITHOUT a workspace runs unsplit."""
    monkeypatch.setenv("QUANTO_HIP_CONV_SPLIT", str(split))
    monkeypatch.setenv("QUANTO_HIP_CONV_ROWS", "0")  # (the tap kernel's split; the row form has its own test)
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(128, 200, 3, padding=1).to(torch.bfloat16)
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, wq))
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(2, 128, 17, 9).to(torch.bfloat16)
    with torch.no_grad():
        y = q(x.cuda())
        assert quanto_hip.lib.last_kernel() == ("conv2d_mfma" if wq == "qint8" else "conv2d_mfma_int4")
        if wq == "qint8":  # exact integers, the per-channel scale on the accumulator
            prod = torch.nn.functional.conv2d(x.double(), q.weight._data.cpu().double(), None, 1, 1) * q.weight._scale.cpu().double().reshape(1, -1, 1, 1)
        else:              # the weight the reference dequantizes
            prod = torch.nn.functional.conv2d(x.double(), _oracle_dequantized(q.weight, "bf16"), None, 1, 1)
    bias = to_numpy(q.bias).astype(np.float64).reshape(1, -1, 1, 1)
    assert_close_with_bias(to_numpy(y), prod.numpy(), np.broadcast_to(bias, prod.shape), "bf16", f"conv K split {split} {wq}")
    if split == 7 and wq == "qint8":  # the C entry without a workspace: same problem, unsplit, same gate
        lib = quanto_hip.lib
        w = q.weight
        yy = torch.empty_like(y)
        s = w._scale.reshape(-1).to(torch.bfloat16).contiguous()
        xc = x.cuda()
        st = lib._c.quanto_hip_qbytes_conv2d(xc.data_ptr(), w._data.data_ptr(), s.data_ptr(), q.bias.data_ptr(), yy.data_ptr(), 2, 128, 17, 9, 200, 3, 3, 17, 9,
                                             1, 1, 1, 1, 1, 1, 2, 3, 2, None, 0, None)
        torch.cuda.synchronize()
        assert st == 0
        assert_close_with_bias(to_numpy(yy), prod.numpy(), np.broadcast_to(bias, prod.shape), "bf16", "conv without a workspace")


@pytest.mark.parametrize("geom", [(2, 2, 6, 8, 3, 3, 1, 1, 1, 1, 1), (1, 3, 5, 10, 3, 3, 2, 0, 2, 1, 1), (2, 1, 8, 14, 3, 5, 1, 1, 2, 1, 2),
                                  (3, 2, 5, 2, 3, 3, 1, 1, 1, 1, 1), (1, 1, 10, 12, 7, 7, 1, 3, 3, 1, 1), (1, 1, 4, 4, 3, 3, 1, 2, 3, 2, 1)])
def test_conv_pair_gather_address_model(geom):
    """CPU model of the kernel's two-pixels-per-load gather (scripts/models/conv_pair_gather_model.py): every load inside the tensor, every
    extracted element the im2col value (0 over the padding) - borders on both sides, paddings wider than one element, W = 2, dilation."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("conv_pair_gather_model", os.path.join(ROOT, "scripts", "models", "conv_pair_gather_model.py"))
    model = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(model)
    assert model.check(*geom) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("wq", ["qint8", "qfloat8_e4m3fn", "qint4", "qint2"])
@pytest.mark.parametrize("cin,cout,k,s,p,d,hw", [(64, 96, 3, 1, 1, 1, (13, 12)),          # both borders of every row: the +2 / -2 byte realignments
                                                 (64, 40, 3, (2, 1), (0, 2), 1, (9, 10)),    # padding 2: taps with NEITHER pixel inside the row; stride 2 down
                                                 (32, 48, (3, 5), 1, (1, 2), (1, 2), (8, 14)),  # dilation 2 along the width, rectangular window
                                                 (16, 24, 3, 1, 1, 1, (5, 2)),              # W = OW = 2: a pair is a whole row
                                                 (128, 72, 1, 1, 0, 1, (6, 6)),             # pointwise: no border at all
                                                 (8, 40, 7, 1, 3, 1, (10, 12)),             # 49 taps: two 64-bit validity words per pixel
                                                 (3, 32, 3, 1, 1, 1, (20, 16)),             # K = 27: a ragged K-tile
                                                 (64, 64, (1, 2), 1, 0, 1, (4, 9)),         # OW = 8 from W = 9 (even window along the width)
                                                 (64, 32, 3, 1, 1, 1, (1, 2))])             # planes of TWO pixels: a lane's four output pixels span two images
def test_qconv2d_pair_gather_gpu(monkeypatch, dt, wq, cin, cout, k, s, p, d, hw):
    """r5: two neighbouring output pixels per 4-byte load (stride 1 along the width, even OW).  Same staged operand as the one-pixel gather, so
    the two kernels' outputs must be IDENTICAL bit for bit (the one-pixel form is forced through QUANTO_HIP_CONV_PAIR=0), and each passes the
    float64 gate; borders on both sides, paddings wider than one element, dilation, W = 2, ragged M, every weight format."""
    torch.manual_seed(cin * 3 + cout)
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=p, dilation=d).to(TORCH_DT[dt])
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, wq))
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(3, cin, *hw).to(TORCH_DT[dt])
    sub = wq in ("qint4", "qint2")
    monkeypatch.setenv("QUANTO_HIP_CONV_ROWS", "0")  # the tap gather's two forms (three-tap windows otherwise take the row form)
    with torch.no_grad():
        y = q(x.cuda())
        assert quanto_hip.lib.last_kernel() == {"qint4": "conv2d_mfma_int4", "qint2": "conv2d_mfma_int2"}.get(wq, "conv2d_mfma")
        assert y.shape[-1] % 2 == 0
        monkeypatch.setenv("QUANTO_HIP_CONV_PAIR", "0")
        y1 = q(x.cuda())
        monkeypatch.delenv("QUANTO_HIP_CONV_PAIR")
        if sub:
            prod = torch.nn.functional.conv2d(x.double(), _oracle_dequantized(q.weight, dt), None, conv.stride, conv.padding, conv.dilation)
        else:
            w64 = q.weight._data.cpu().double() if wq == "qint8" else q.weight._data.cpu().float().double()
            prod = torch.nn.functional.conv2d(x.double(), w64, None, conv.stride, conv.padding, conv.dilation) * q.weight._scale.cpu().double().reshape(1, -1, 1, 1)
    assert torch.equal(y, y1), f"pair gather differs from the one-pixel gather: {(y != y1).sum().item()} of {y.numel()} elements"
    bias = to_numpy(q.bias).astype(np.float64).reshape(1, -1, 1, 1)
    assert_close_with_bias(to_numpy(y), prod.numpy(), np.broadcast_to(bias, prod.shape), dt, f"pair gather {wq} {cin}->{cout} k{k}")


@pytest.mark.parametrize("case", range(12))
def test_conv_row_form_model(case):
    """CPU model of the row form (scripts/models/conv_rows_model.py): clamped 8-byte windows, selectors, scalar row arithmetic, the 4 x 4 v_perm
    transposition, the LDS image and the fragment reads, the weight bytes' regrouping, the K split - index for index against a direct convolution
    in exact integer arithmetic; and the LDS bank model (conflict-free fragment reads, two-way staging stores)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("conv_rows_model", os.path.join(ROOT, "scripts", "models", "conv_rows_model.py"))
    model = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(model)
    if case < 7:
        assert model.run_case(model.CASES[case], seed=case)
    else:  # one pixel per thread: strides 2 / 3 along the width, odd OW, a pair-eligible geometry forced onto it
        assert model.run_case(model.SINGLE_CASES[case - 7], seed=case, single=True)
    if case == 0:
        assert model.lds_bank_model() == (1, 2, 2) and model.lds_bank_model_single() == 1


@pytest.mark.parametrize("seed", range(4))
def test_conv_row_form_model_random_geometries(seed):
    """The row form's CPU model over random corner geometries: paddings up to 3 columns / rows, strides and height dilations up to 3, 1 .. 10 tap
    rows, ragged pixel / channel / K tiles, up to 7 splits - both thread mappings wherever the geometry allows pixel pairs.  Besides the result the
    model asserts that every in-range window lies inside the tensor and every selector picks a byte of the window."""
    import importlib.util
    import random

    spec = importlib.util.spec_from_file_location("conv_rows_model", os.path.join(ROOT, "scripts", "models", "conv_rows_model.py"))
    model = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(model)
    rnd = random.Random(1000 + seed)
    done = 0
    while done < 6:
        KH = rnd.choice([1, 2, 3, 3, 4, 5, 7, 10])
        cin = rnd.choice([c for c in range(1, 41) if (c * KH) % 8 == 0][:4])
        B, H, W, OC = rnd.choice([1, 2, 3]), rnd.randint(max(1, KH - 2), 8), rnd.randint(4, 12), rnd.choice([4, 12, 132])
        sh, sw, ph, pw, dh, S = rnd.choice([1, 2, 3]), rnd.choice([1, 1, 2, 3]), rnd.randint(0, 3), rnd.randint(0, 3), rnd.choice([1, 2, 3]), rnd.choice([1, 2, 7])
        OH, OW = (H + 2 * ph - dh * (KH - 1) - 1) // sh + 1, (W + 2 * pw - 3) // sw + 1
        if OH < 1 or OW < 1:
            continue
        case = (B, cin, H, W, OC, KH, sh, ph, pw, dh, S, sw)
        for single in ([False, True] if model.rows_pairs(OW, sw) else [True]):
            assert model.run_case(case, seed=seed, single=single), (case, single)
        done += 1


ROW_FORM_GEOMETRIES = [(64, 96, 3, 1, 1, 1, (13, 12)),              # "same": both borders of every row, ragged M (3 x 13 x 12 pixels)
                       (64, 40, 3, (2, 1), (0, 2), 1, (9, 10)),        # two columns of padding (windows with two elements outside), stride 2 down
                       (32, 48, (5, 3), 1, (2, 1), (2, 1), (12, 8)),   # five tap rows, dilation 2 along the height
                       (16, 24, 3, 1, 0, 1, (6, 6)),                  # "valid": no padding at all, OW = 4
                       (8, 40, (1, 3), 1, (0, 1), 1, (5, 4)),          # 1 x 3 window, W = 4: the clamped window is the whole row
                       (24, 136, 3, 1, 1, 1, (9, 10)),                # 72 window rows: a ragged last K-tile; two channel tiles, the second ragged
                       (128, 128, 3, 1, 1, 1, (28, 28))]              # a ResNet layer: 12 K-tiles, 19 pixel tiles


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("wq", ["qint8", "qfloat8_e4m3fn", "qfloat8_e5m2"])
@pytest.mark.parametrize("cin,cout,k,s,p,d,hw", ROW_FORM_GEOMETRIES)
@pytest.mark.parametrize("loads", ["buffer", "global"])
def test_qconv2d_row_form_gpu(monkeypatch, loads, dt, wq, cin, cout, k, s, p, d, hw):
    """r5: windows three taps wide at stride 1 along the width take the ROW form - one 8-byte range-checked load per (pixel pair, window row),
    K ordered [tap][row] inside a tile.  Gate: the float64 convolution on the stored integers / fp8 values (with and without bias); the variant
    with plain global loads (QUANTO_HIP_CONV_ROWS=2) stages the same operands, so it must agree bit for bit."""
    torch.manual_seed(cin * 5 + cout)
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=p, dilation=d).to(TORCH_DT[dt])
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, wq))
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(3, cin, *hw).to(TORCH_DT[dt])
    assert _row_form(cin, k, s, d, hw[1], p, wq)
    with torch.no_grad():
        y2 = q(x.cuda())  # the product's form: range-checked buffer loads
        assert quanto_hip.lib.last_kernel() == "conv2d_mfma_rows"
        if loads == "global":
            monkeypatch.setenv("QUANTO_HIP_CONV_ROWS", "2")
        y = q(x.cuda())
        assert quanto_hip.lib.last_kernel() == "conv2d_mfma_rows"
        w64 = q.weight._data.cpu().double() if wq == "qint8" else q.weight._data.cpu().float().double()
        prod = torch.nn.functional.conv2d(x.double(), w64, None, conv.stride, conv.padding, conv.dilation) * q.weight._scale.cpu().double().reshape(1, -1, 1, 1)
    assert y.shape == prod.shape and y.dtype == TORCH_DT[dt] and torch.isfinite(y.float()).all()
    bias = to_numpy(q.bias).astype(np.float64).reshape(1, -1, 1, 1)
    assert_close_with_bias(to_numpy(y), prod.numpy(), np.broadcast_to(bias, prod.shape), dt, f"row form {wq} {cin}->{cout} k{k}")
    if loads == "global":
        assert torch.equal(y, y2), f"buffer-load and global-load variants differ: {(y != y2).sum().item()} of {y.numel()} elements"
    q.bias = None
    with torch.no_grad():
        y0 = q(x.cuda())
    assert_close_to_exact(to_numpy(y0), prod.numpy(), dt, f"row form {wq} {cin}->{cout} k{k}, no bias")


SINGLE_ROW_FORM_GEOMETRIES = [(64, 96, 3, 2, 1, 1, (13, 12)),                   # the downsampling 3 x 3: stride 2 both ways
                              (64, 40, 3, 1, 1, 1, (9, 7)),                      # stride 1, OW = 7: no pixel pairs
                              (32, 48, (5, 3), (1, 3), (2, 2), (2, 1), (12, 11)),  # stride 3 along the width, two columns of padding, five dilated tap rows
                              (24, 136, 3, 2, 1, 1, (9, 9)),                     # 72 window rows (ragged last K-tile), two channel tiles
                              (128, 128, 3, 2, 1, 1, (28, 28)),                  # a ResNet stage's first layer
                              (16, 24, (1, 3), 1, 0, 1, (5, 5))]                 # 1 x 3 "valid", OW = 3


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("wq", ["qint8", "qfloat8_e4m3fn", "qint4", "qint2"])
@pytest.mark.parametrize("cin,cout,k,s,p,d,hw", SINGLE_ROW_FORM_GEOMETRIES)
def test_qconv2d_row_form_one_pixel_per_thread_gpu(monkeypatch, dt, wq, cin, cout, k, s, p, d, hw):
    """r5: three-tap-wide windows at ANY stride along the width / with an odd OW - the row form with one output pixel per thread and eight window
    rows (int4 / int2: on the weight dequantized once - taken from 8 pixel tiles on; forced here for these small images).  Float64 gate as for the pair form."""
    monkeypatch.setenv("QUANTO_HIP_CONV_DENSE_MIN_TILES", "1")
    torch.manual_seed(cin * 11 + cout)
    cout -= cout % 4
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=p, dilation=d).to(TORCH_DT[dt])
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, wq))
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(3, cin, *hw).to(TORCH_DT[dt])
    sub = wq in ("qint4", "qint2")
    with torch.no_grad():
        y = q(x.cuda())
        assert quanto_hip.lib.last_kernel() == _conv_kernel_name(cin, k, s, d, hw[1], p, wq, pixels=y.numel() // y.shape[1], dense_min_tiles=1) and "rows" in quanto_hip.lib.last_kernel()
        if sub:
            prod = torch.nn.functional.conv2d(x.double(), _oracle_dequantized(q.weight, dt), None, conv.stride, conv.padding, conv.dilation)
        else:
            w64 = q.weight._data.cpu().double() if wq == "qint8" else q.weight._data.cpu().float().double()
            prod = torch.nn.functional.conv2d(x.double(), w64, None, conv.stride, conv.padding, conv.dilation) * q.weight._scale.cpu().double().reshape(1, -1, 1, 1)
    assert y.shape == prod.shape and y.dtype == TORCH_DT[dt] and torch.isfinite(y.float()).all()
    bias = to_numpy(q.bias).astype(np.float64).reshape(1, -1, 1, 1)
    assert_close_with_bias(to_numpy(y), prod.numpy(), np.broadcast_to(bias, prod.shape), dt, f"row form, one pixel per thread, {wq} {cin}->{cout} k{k} s{s}")


@pytest.mark.gpu
@pytest.mark.parametrize("wq", ["qint8", "qint4"])
@pytest.mark.parametrize("cin,cout,k,s,p,d,hw", [ROW_FORM_GEOMETRIES[i] for i in (0, 1, 2, 5)])
def test_qconv2d_row_form_pairs_and_single_pixels_agree_gpu(monkeypatch, wq, cin, cout, k, s, p, d, hw):
    """The two thread mappings of the row form stage the same LDS image and run the same MFMA sequence: on a geometry both can take
    (QUANTO_HIP_CONV_ROWS=3 forces one pixel per thread) the outputs must be identical bit for bit."""
    monkeypatch.setenv("QUANTO_HIP_CONV_DENSE_MIN_TILES", "1")
    torch.manual_seed(cin + cout)
    cout -= cout % 4
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=p, dilation=d).to(torch.bfloat16)
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, wq))
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(3, cin, *hw).to(torch.bfloat16)
    with torch.no_grad():
        y_pairs = q(x.cuda())
        name = quanto_hip.lib.last_kernel()
        monkeypatch.setenv("QUANTO_HIP_CONV_ROWS", "3")
        y_single = q(x.cuda())
        assert quanto_hip.lib.last_kernel() == name and "rows" in name
    assert torch.equal(y_pairs, y_single), f"{(y_pairs != y_single).sum().item()} of {y_pairs.numel()} elements differ"


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("bits,zp,gs", [(4, False, 64), (4, True, 128), (4, False, None), (2, False, 32), (2, True, None)])
@pytest.mark.parametrize("cin,cout,k,s,p,d,hw", [ROW_FORM_GEOMETRIES[i] for i in (0, 1, 2, 5, 6)])
def test_qconv2d_subbyte_row_form_gpu(monkeypatch, dt, bits, zp, gs, cin, cout, k, s, p, d, hw):
    """r5: int4 / int2 weights on three-tap-wide windows - the weight is dequantized ONCE into the scratch buffer (the reference's dense weight,
    bit for bit) and the row form multiplies by it; every pixel tile of the tap kernel dequantizes the whole weight again.  Gate: the float64
    convolution with the weight the reference dequantizes; the tap kernel on the same call stays within an ulp of it."""
    kh = k if isinstance(k, int) else k[0]
    if gs is not None and (cin * kh * 3) % gs:
        pytest.skip("group size does not divide K")
    monkeypatch.setenv("QUANTO_HIP_CONV_DENSE_MIN_TILES", "1")  # (the product takes this route from 8 pixel tiles on; these images have 2 .. 19)
    qt = Q.qint4 if bits == 4 else Q.qint2
    cout -= cout % 4
    torch.manual_seed(cin + cout + bits)
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=p, dilation=d).to(TORCH_DT[dt])
    q = Q.QConv2d.from_module(conv, weights=qt)
    Q.freeze(q)
    scale, shift = Q.MaxOptimizer()(conv.weight.detach(), qt, 0, gs, zeropoint=zp)
    w = Q.quantize_weight(conv.weight.detach(), qt, 0, scale, shift, group_size=gs, optimized=False)
    assert w._group_size == gs and (w._shift.dtype == torch.uint8) == zp
    q.weight = torch.nn.Parameter(w, requires_grad=False)
    q = q.cuda()
    x = torch.randn(3, cin, *hw).to(TORCH_DT[dt])
    with torch.no_grad():
        y = q(x.cuda())
        assert quanto_hip.lib.last_kernel() == f"conv2d_rows_dequant_int{bits}"
        monkeypatch.setenv("QUANTO_HIP_CONV_ROWS", "0")
        y_taps = q(x.cuda())
        assert quanto_hip.lib.last_kernel() == f"conv2d_mfma_int{bits}"
        prod = torch.nn.functional.conv2d(x.double(), _oracle_dequantized(q.weight, dt), None, conv.stride, conv.padding, conv.dilation)
    bias = to_numpy(q.bias).astype(np.float64).reshape(1, -1, 1, 1)
    assert_close_with_bias(to_numpy(y), prod.numpy(), np.broadcast_to(bias, prod.shape), dt, f"int{bits} row form {cin}->{cout} k{k} g{gs}")
    eps = {"fp16": 2.0 ** -10, "bf16": 2.0 ** -7}[dt]
    diff = (y.float() - y_taps.float()).abs().cpu().double()
    assert (diff <= eps * (prod.abs() + y_taps.cpu().double().abs()) * 1.01 + 1e-9).all()


@pytest.mark.gpu
@pytest.mark.parametrize("wq", ["qint8", "qfloat8_e4m3fn", "qint4"])
@pytest.mark.parametrize("cin,cout,k,s,p,hw,split", [(128, 128, 3, 1, 1, (28, 28), 1), (128, 200, 3, 1, 1, (17, 10), 5), (24, 136, 3, 1, 1, (9, 10), 1),
                                                      (128, 128, 3, 2, 1, (28, 28), 1), (64, 40, 3, 1, 1, (9, 7), 2), (8, 40, (1, 3), 1, (0, 1), (5, 4), 1)])
def test_qconv2d_row_form_two_lds_buffers_bit_equal_gpu(monkeypatch, wq, cin, cout, k, s, p, hw, split):
    """r6: the row form with TWO LDS buffers and one barrier per K-tile (grids that leave every workgroup a CU of its own) stages and multiplies the same
    operands in the same order as the one-buffer form: forced on (2) and off (0) the outputs are bit-identical - pair and one-pixel mappings, K split, a
    single K-tile (no second buffer ever filled), ragged last tiles, sub-byte weights through the dense route."""
    monkeypatch.setenv("QUANTO_HIP_CONV_SPLIT", str(split))
    monkeypatch.setenv("QUANTO_HIP_CONV_DENSE_MIN_TILES", "1")
    torch.manual_seed(cin + cout)
    conv = torch.nn.Conv2d(cin, cout, k, stride=s, padding=p).to(torch.bfloat16)
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, wq))
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(3, cin, *hw).to(torch.bfloat16).cuda()
    ys = {}
    with torch.no_grad():
        for db in ("0", "2"):
            monkeypatch.setenv("QUANTO_HIP_CONV_ROWS_DB", db)
            ys[db] = q(x)
            assert quanto_hip.lib.last_kernel() in ("conv2d_mfma_rows", "qbits_conv2d_rows", "conv2d_dense_rows") or "rows" in quanto_hip.lib.last_kernel()
    assert torch.isfinite(ys["0"].float()).all()
    assert torch.equal(ys["0"], ys["2"]), f"{(ys['0'] != ys['2']).sum().item()} of {ys['0'].numel()} elements differ"


@pytest.mark.gpu
@pytest.mark.parametrize("split", [1, 2, 5, 64])
def test_qconv2d_row_form_k_split_gpu(monkeypatch, split):
    """The row form under the K split (12 K-tiles of 32 window rows: 5 is ragged, 64 is clamped to 12) and against the tap gather on the same call."""
    monkeypatch.setenv("QUANTO_HIP_CONV_SPLIT", str(split))
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(128, 200, 3, padding=1).to(torch.bfloat16)
    q = Q.QConv2d.from_module(conv, weights=Q.qint8)
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(2, 128, 17, 10).to(torch.bfloat16)
    with torch.no_grad():
        y = q(x.cuda())
        assert quanto_hip.lib.last_kernel() == "conv2d_mfma_rows"
        y_again = q(x.cuda())
        monkeypatch.setenv("QUANTO_HIP_CONV_ROWS", "0")
        y_taps = q(x.cuda())
        assert quanto_hip.lib.last_kernel() == "conv2d_mfma"
        prod = torch.nn.functional.conv2d(x.double(), q.weight._data.cpu().double(), None, 1, 1) * q.weight._scale.cpu().double().reshape(1, -1, 1, 1)
    bias = to_numpy(q.bias).astype(np.float64).reshape(1, -1, 1, 1)
    assert_close_with_bias(to_numpy(y), prod.numpy(), np.broadcast_to(bias, prod.shape), "bf16", f"row form, K split {split}")
    assert torch.equal(y, y_again)  # (the reduce kernel adds in split order)
    # two summation orders of the same exact products: at most an ulp apart after the rounding to bf16
    diff = (y.float() - y_taps.float()).abs().cpu().double()
    assert (diff <= 2.0 ** -7 * (prod.abs() + y_taps.cpu().double().abs()) + 1e-9).all()


@pytest.mark.gpu
def test_qconv2d_row_form_at_bench_size_gpu():
    """(8, 128, 56, 56) -> 128, the size the timings are quoted on: whole-output float64 gate."""
    torch.manual_seed(11)
    conv = torch.nn.Conv2d(128, 128, 3, padding=1).to(torch.bfloat16)
    q = Q.QConv2d.from_module(conv, weights=Q.qint8)
    Q.freeze(q)
    q = q.cuda()
    x = torch.randn(8, 128, 56, 56).to(torch.bfloat16)
    with torch.no_grad():
        y = q(x.cuda())
        assert quanto_hip.lib.last_kernel() == "conv2d_mfma_rows"
        prod = torch.nn.functional.conv2d(x.double(), q.weight._data.cpu().double(), None, 1, 1) * q.weight._scale.cpu().double().reshape(1, -1, 1, 1)
    bias = to_numpy(q.bias).astype(np.float64).reshape(1, -1, 1, 1)
    assert_close_with_bias(to_numpy(y), prod.numpy(), np.broadcast_to(bias, prod.shape), "bf16", "row form (8,128,56,56)->128")


@pytest.mark.gpu
def test_qconv2d_grouped_convolution_keeps_reference_behaviour_gpu():
    """groups != 1 is not lowered to a GEMM: dequantize + float convolution on the device, as the reference does."""
    torch.manual_seed(9)
    conv = torch.nn.Conv2d(16, 32, 3, padding=1, groups=4).to(torch.bfloat16).cuda()
    q = Q.QConv2d.from_module(conv, weights=Q.qint8)
    Q.freeze(q)
    x = torch.randn(2, 16, 9, 9, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        y = q(x)
        want = torch.nn.functional.conv2d(x, q.weight.dequantize(), q.bias, padding=1, groups=4)
    torch.testing.assert_close(y, want, rtol=0, atol=0)


def test_conv_model_with_quantized_activations_calibrates_and_runs():
    """quantize(weights=qint8, activations=qint8) on a small conv net: QConv2d / QLinear are created (LayerNorm stays a float
    module: its quantized twin is outside this backend's scope), activation scales are set from one observed batch, the frozen
    model stays close to the float model."""
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(16, 4, 3), torch.nn.Flatten(),
                                torch.nn.LayerNorm(4 * 6 * 6), torch.nn.Linear(4 * 6 * 6, 10))
    x = torch.randn(2, 8, 8, 8)
    ref = model(x)
    Q.quantize(model, weights=Q.qint8, activations=Q.qint8)
    assert [type(layer).__name__ for layer in model] == ["QConv2d", "ReLU", "QConv2d", "Flatten", "LayerNorm", "QLinear"]
    with torch.no_grad(), observed_activation_scales():
        model(x)
    assert float(model[0].output_scale) != 1.0 and float(model[5].input_scale) != 1.0
    Q.freeze(model)
    with torch.no_grad():
        y = model(x)
    y = y.dequantize() if isinstance(y, Q.QTensor) else y
    assert (y - ref).abs().max() < 0.05 * ref.abs().max()
