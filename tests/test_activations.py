"""Quantized activations: oracle and host mirror against the reference's golden vectors
on CPU; the one-pass quantize kernel and the int8 x int8 / fp8 x fp8 MFMA product against the oracle on the GPU.

Reference tests mirrored: tests/tensor/activations/test_activations_quantize.py, tests/tensor/ops/test_linear_dispatch.py:22-42,
tests/nn/test_qlinear.py (quantize_linear_*_activations), tests/nn/test_calibrate.py.
"""
import numpy as np
import pytest
import torch

import optimum_quanto_amd as Q
from optimum_quanto_amd.library.hip import quanto_hip
from oracle import quanto_oracle as O

from helpers import FP8_TORCH, TORCH_DT, assert_similar, fp8_tensor, observed_activation_scales, to_numpy, to_torch

QTYPES = {"int8": Q.qint8, "e4m3fn": Q.qfloat8_e4m3fn, "e5m2": Q.qfloat8_e5m2}
GPU = torch.cuda.is_available()
DEV = torch.device("cuda", 0) if GPU else None


def _oracle_quantize(x, scale, qname, dt):
    if qname == "int8":
        return O.quantize_symmetric_int8(x, scale, dt)
    return O.quantize_symmetric_fp8(x, scale, qname, dt)


def _qmax(qname):
    return 127.0 if qname == "int8" else O.FP8_MAX[qname]


# ---------------------------------------------------------------------------------------------------------------------
# CPU: oracle and host mirror vs the reference
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("qname", ["int8", "e4m3fn", "e5m2"])
def test_oracle_quantize_activation_matches_reference(golden, qname, dt):
    k = f"qact/{qname}_{dt}"
    x = golden[k + "/x"]
    scale = O.absmax_scale(x, _qmax(qname), None, dt)
    np.testing.assert_array_equal(scale, golden[k + "/scale"])
    data = _oracle_quantize(x, scale, qname, dt)
    np.testing.assert_array_equal(data.view(np.uint8) if qname != "int8" else data, golden[k + "/data"])
    deq = O.dequantize_qbytes_ref(data, scale, dt, None if qname == "int8" else qname)
    np.testing.assert_array_equal(deq, golden[k + "/dequantized"])


@pytest.mark.parametrize("dt", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("qname", ["int8", "e4m3fn", "e5m2"])
def test_host_quantize_activation_matches_reference(golden, qname, dt):
    k = f"qact/{qname}_{dt}"
    x = to_torch(golden[k + "/x"], dt)
    scale = Q.absmax_scale(x, QTYPES[qname])
    np.testing.assert_array_equal(to_numpy(scale), golden[k + "/scale"])
    qx = Q.quantize_activation(x, QTYPES[qname], scale)
    assert isinstance(qx, Q.ActivationQBytesTensor) and qx.qtype == QTYPES[qname] and qx.axis is None
    assert qx.dtype == TORCH_DT[dt] and qx.shape == x.shape
    np.testing.assert_array_equal(to_numpy(qx._data), golden[k + "/data"])
    np.testing.assert_array_equal(to_numpy(qx.dequantize()), golden[k + "/dequantized"])


def test_quantize_activation_rejects_non_scalar_scale():
    with pytest.raises(ValueError):
        Q.quantize_activation(torch.randn(4, 8), Q.qint8, torch.ones(4, 1))


@pytest.mark.parametrize("dt", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("tag", ["a8w8_int8", "a8w8_e4m3fn", "aint8_we4m3fn"])
def test_oracle_qact_linear_matches_reference(golden, tag, dt):
    """y = qbytes_mm(x_data, w_data, x_scale * w_scale) (+ bias), tensor/weights/qbytes.py:68-82."""
    k = f"qact_linear/{tag}_{dt}"
    xd, wd = golden[k + "/xdata"], golden[k + "/wdata"]
    s = O.round_to(golden[k + "/xscale"] * golden[k + "/wscale"], dt)
    akind = None if tag != "a8w8_e4m3fn" else "e4m3fn"
    wkind = None if tag == "a8w8_int8" else "e4m3fn"
    a = xd.astype(np.float64) if akind is None else O.fp8_decode(xd, akind).astype(np.float64)
    w = wd.astype(np.float64) if wkind is None else O.fp8_decode(wd, wkind).astype(np.float64)
    exact = np.matmul(a.reshape(-1, a.shape[-1]), w.T) * s.astype(np.float64).reshape(1, -1)
    want = golden[k + "/y_nobias"].reshape(exact.shape)
    # the reference's CPU summation order / promotion is not part of its contract: north-star tolerance on the result
    assert O.rel_fro(want, exact) < (1e-3 if dt != "bf16" else 4e-3)


@pytest.mark.parametrize("dt", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("tag", ["a8w8_int8", "a8w8_e4m3fn", "aint8_we4m3fn"])
def test_host_qact_linear_matches_reference(golden, tag, dt):
    k = f"qact_linear/{tag}_{dt}"
    aq = Q.qint8 if tag != "a8w8_e4m3fn" else Q.qfloat8_e4m3fn
    wq = Q.qint8 if tag == "a8w8_int8" else Q.qfloat8_e4m3fn
    w, x, bias = (to_torch(golden[k + n], dt) for n in ("/w", "/x", "/bias"))
    qw = Q.quantize_weight(w, qtype=wq, axis=0, scale=Q.AbsmaxOptimizer()(w, qtype=wq, axis=0), activation_qtype=aq)
    qx = Q.quantize_activation(x, aq, Q.absmax_scale(x, aq))
    np.testing.assert_array_equal(to_numpy(qw._data), golden[k + "/wdata"])
    np.testing.assert_array_equal(to_numpy(qx._data), golden[k + "/xdata"])
    with torch.no_grad():
        y = torch.nn.functional.linear(qx, qw, bias)
    assert type(y) is torch.Tensor and y.dtype == TORCH_DT[dt] and y.shape == (2, 40, 192)
    assert_similar(to_torch(golden[k + "/y"], dt), y)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_host_qlinear_with_calibrated_scales_matches_reference(golden, dt):
    """QLinear(weights=qint8, activations=qint8) with the activation scales the reference's Calibration pass produced (golden):
    the frozen module's quantized output matches the reference's (nn/qmodule.py:131-134,281-299)."""
    k = f"qlinear_a8w8/{dt}"
    lin = torch.nn.Linear(128, 192).to(TORCH_DT[dt])
    with torch.no_grad():
        lin.weight.copy_(to_torch(golden[k + "/w"], dt))
        lin.bias.copy_(to_torch(golden[k + "/bias"], dt))
    q = Q.QLinear.from_module(lin, weights=Q.qint8, activations=Q.qint8)
    x0 = to_torch(golden[k + "/x0"], dt)
    q.input_scale = to_torch(golden[k + "/input_scale"], dt).reshape(())
    q.output_scale = to_torch(golden[k + "/output_scale"], dt).reshape(())
    Q.freeze(q)
    with torch.no_grad():
        y = q(x0)
    assert isinstance(y, Q.ActivationQBytesTensor)
    got, want = to_numpy(y._data).astype(np.int32), golden[k + "/y_data"].astype(np.int32)
    assert np.abs(got - want).max() <= (2 if dt == "bf16" else 1)
    assert (got != want).mean() < (0.2 if dt == "bf16" else 0.01)


def test_activation_tensor_ops_keep_quantization():
    x = torch.randn(4, 6, 16)
    qx = Q.quantize_activation(x, Q.qint8, Q.absmax_scale(x))
    for out in (qx.view(24, 16), qx.transpose(0, 1), qx.permute(2, 0, 1), qx[1], qx.unsqueeze(0), qx * 2.0, qx / 4, torch.relu(qx),
                -qx, qx.detach(), qx.clone(), qx.to(torch.float16), torch.softmax(qx, -1), torch.cat([qx, qx]), torch.stack([qx, qx])):
        assert isinstance(out, Q.ActivationQBytesTensor), type(out)
    assert torch.equal((qx * 2.0).dequantize(), qx.dequantize() * 2.0)
    assert torch.equal(qx.transpose(0, 1).dequantize(), qx.dequantize().transpose(0, 1))
    assert type(qx + 1.0) is torch.Tensor  # no quantized add: dequantizes
    assert type(torch.nn.functional.silu(qx)) is torch.Tensor
    # serialisation round trip through the flatten protocol
    names, meta = qx.__tensor_flatten__()
    back = Q.ActivationQBytesTensor.__tensor_unflatten__({n: getattr(qx, n) for n in names}, meta, None, None)
    assert back.equal(qx)


def test_qlinear_heterogeneous_activation_qtypes_rejected():
    q = Q.QLinear.from_module(torch.nn.Linear(16, 16), weights=Q.qint8, activations=Q.qint8)
    x = torch.randn(2, 16)
    with pytest.raises(ValueError):
        q(Q.quantize_activation(x, Q.qfloat8_e4m3fn, Q.absmax_scale(x, Q.qfloat8_e4m3fn)))


# ---------------------------------------------------------------------------------------------------------------------
# GPU: kernels vs oracle
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("qname", ["int8", "e4m3fn", "e5m2"])
@pytest.mark.parametrize("shape", [(4, 24, 64), (1, 7), (3, 1001), (256, 4096), (5,)])
def test_hip_quantize_symmetric_per_tensor_bit_exact(shape, qname, dt):
    from optimum_quanto_amd.library.hip import quanto_hip

    rng = np.random.default_rng(sum(shape))
    x = O.round_to((rng.standard_normal(shape) * 3).astype(np.float32), dt)
    x.flat[0] = 0.0
    scale = O.absmax_scale(x, _qmax(qname), None, dt) * np.float32(0.8)  # < absmax/qmax: exercises the clamp
    scale = O.round_to(np.asarray(scale, np.float32), dt)
    want = _oracle_quantize(x, scale, qname, dt)
    tdt = torch.int8 if qname == "int8" else FP8_TORCH[qname]
    tscale = to_torch(scale, dt, DEV).reshape(())
    got = torch.ops.quanto.quantize_symmetric(to_torch(x, dt, DEV), dtype=tdt, axis=None, scale=tscale)
    assert got.dtype == tdt and got.shape == x.shape
    np.testing.assert_array_equal(to_numpy(got).view(np.uint8), want.view(np.uint8))
    # and through the C ABI directly
    got2 = quanto_hip.lib.quantize_symmetric(to_torch(x, dt, DEV), tdt, None, tscale)
    np.testing.assert_array_equal(to_numpy(got2).view(np.uint8), want.view(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("qname", ["int8", "e4m3fn"])
@pytest.mark.parametrize("axis", [0, -1])
@pytest.mark.parametrize("shape", [(48, 64), (33, 7), (5, 3, 4), (256, 1024)])
def test_hip_quantize_symmetric_per_axis_bit_exact(shape, axis, qname, dt):
    rng = np.random.default_rng(sum(shape) + axis)
    x = O.round_to(rng.standard_normal(shape).astype(np.float32), dt)
    scale = O.absmax_scale(x, _qmax(qname), axis, dt)
    want = _oracle_quantize(x, scale, qname, dt)
    tdt = torch.int8 if qname == "int8" else FP8_TORCH[qname]
    got = torch.ops.quanto.quantize_symmetric(to_torch(x, dt, DEV), dtype=tdt, axis=axis, scale=to_torch(scale, dt, DEV))
    np.testing.assert_array_equal(to_numpy(got).view(np.uint8), want.view(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("tag", ["a8w8_int8", "a8w8_e4m3fn", "aint8_we4m3fn"])
def test_hip_qact_linear_golden(golden, tag, dt):
    """F.linear(quantized activation, quantized weight) on the device vs the reference's output and vs the oracle."""
    from optimum_quanto_amd.library.hip import quanto_hip

    k = f"qact_linear/{tag}_{dt}"
    aq = Q.qint8 if tag != "a8w8_e4m3fn" else Q.qfloat8_e4m3fn
    wq = Q.qint8 if tag == "a8w8_int8" else Q.qfloat8_e4m3fn
    w, x, bias = (to_torch(golden[k + n], dt, DEV) for n in ("/w", "/x", "/bias"))
    # scales are taken from the fixture: torch's device `max / 127` multiplies by a reciprocal (1 ulp off the CPU's divide)
    qw = Q.quantize_weight(w, qtype=wq, axis=0, scale=to_torch(golden[k + "/wscale"], dt, DEV), activation_qtype=aq)
    qx = Q.quantize_activation(x, aq, to_torch(golden[k + "/xscale"], dt, DEV).reshape(()))
    np.testing.assert_array_equal(to_numpy(qw._data), golden[k + "/wdata"])  # device quantization == reference CPU, bit for bit
    np.testing.assert_array_equal(to_numpy(qx._data), golden[k + "/xdata"])
    with torch.no_grad():
        y = torch.nn.functional.linear(qx, qw, bias)
        y_nobias = torch.nn.functional.linear(qx, qw)
    assert quanto_hip.lib.last_kernel() == ("naive" if tag == "aint8_we4m3fn" else "mfma_native8")
    assert_similar(to_torch(golden[k + "/y"], dt), y.cpu())
    if tag == "a8w8_int8":
        s = O.round_to(golden[k + "/xscale"] * golden[k + "/wscale"], dt)
        want = O.qbytes_int_mm_ref(golden[k + "/xdata"].reshape(-1, 128), golden[k + "/wdata"], s, dt)
        np.testing.assert_array_equal(to_numpy(y_nobias).reshape(want.shape), want)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_hip_qlinear_a8w8_calibrated(golden, dt):
    k = f"qlinear_a8w8/{dt}"
    lin = torch.nn.Linear(128, 192).to(TORCH_DT[dt])
    with torch.no_grad():
        lin.weight.copy_(to_torch(golden[k + "/w"], dt))
        lin.bias.copy_(to_torch(golden[k + "/bias"], dt))
    q = Q.QLinear.from_module(lin.to(DEV), weights=Q.qint8, activations=Q.qint8)
    batches = [to_torch(golden[k + "/x0"], dt, DEV), to_torch(golden[k + "/x1"], dt, DEV)]
    q.input_scale = to_torch(golden[k + "/input_scale"], dt, DEV).reshape(())    # the reference's calibrated scales (golden)
    q.output_scale = to_torch(golden[k + "/output_scale"], dt, DEV).reshape(())
    Q.freeze(q)
    with torch.no_grad():
        y = q(batches[0])
    assert isinstance(y, Q.ActivationQBytesTensor) and y._data.is_cuda
    got, want = to_numpy(y._data).astype(np.int32), golden[k + "/y_data"].astype(np.int32)
    assert np.abs(got - want).max() <= (2 if dt == "bf16" else 1)
    assert (got != want).mean() < (0.2 if dt == "bf16" else 0.02)


@pytest.mark.gpu
@pytest.mark.parametrize("batch_size", [1, 10])
@pytest.mark.parametrize("tokens, embeddings", [(5, 5), (32, 32), (10, 32), (32, 128)])
@pytest.mark.parametrize("use_bias", [True, False], ids=["bias", "no-bias"])
@pytest.mark.parametrize("dt", ["fp32", "fp16"])
@pytest.mark.parametrize("wname", ["qint2", "qint4", "qint8"])
def test_qactivation_qweight_linear_reference_grid(batch_size, tokens, embeddings, use_bias, dt, wname):
    """tests/tensor/ops/test_linear_dispatch.py:22-42 with a-qint8, on the device."""
    torch.manual_seed(batch_size + tokens + embeddings)
    tdt = TORCH_DT[dt]
    x = (torch.rand((batch_size, tokens, embeddings), dtype=torch.float32) * 2 - 1).to(tdt).to(DEV)
    qx = Q.quantize_activation(x, Q.qint8, Q.absmax_scale(x, Q.qint8))
    w = (torch.rand((embeddings, embeddings), dtype=torch.float32) * 2 - 1).to(tdt).to(DEV)
    wq = Q.qtypes[wname]
    if wname == "qint8":
        qw = Q.quantize_weight(w, qtype=wq, axis=0, scale=Q.AbsmaxOptimizer()(w, qtype=wq, axis=0), activation_qtype=Q.qint8)
    else:
        scale, shift = Q.MaxOptimizer()(w, qtype=wq, axis=0, group_size=None)
        qw = Q.quantize_weight(w, qtype=wq, axis=0, scale=scale, shift=shift, group_size=None)
    bias = (torch.rand((embeddings,), dtype=torch.float32) * 2 - 1).to(tdt).to(DEV) if use_bias else None
    qout = torch.nn.functional.linear(qx, qw, bias)
    out = torch.nn.functional.linear(qx.dequantize(), qw.dequantize(), bias)
    assert_similar(out, qout)


# ---------------------------------------------------------------------------------------------------------------------
# GPU: fused quantize_affine / pack (freeze-time half of SURVEY.md 8f rank 2) vs the reference-pinned oracle, bit-exact
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("bits", [2, 4])
@pytest.mark.parametrize("zeropoint", [False, True])
@pytest.mark.parametrize("N,K,gs", [(48, 256, 128), (33, 96, 32), (256, 4096, 128), (10, 50, None), (64, 64, 64), (7, 128, 64)])
def test_hip_quantize_affine_and_pack_bit_exact(N, K, gs, zeropoint, bits, dt):
    rng = np.random.default_rng(N + K + bits)
    w = O.round_to((rng.standard_normal((N, K)) * 0.05).astype(np.float32), dt)
    scale, shift = O.max_scale_shift(w, bits, 0, gs, dt)
    if zeropoint:
        shift = np.clip(np.rint(O.round_to(shift / scale, dt)), 0, 2**bits - 1).astype(np.uint8)
    want = O.quantize_affine(w, bits, 0, gs, scale, shift, dt)
    tshift = torch.from_numpy(shift).to(DEV) if zeropoint else to_torch(shift, dt, DEV)
    got = torch.ops.quanto.quantize_affine(to_torch(w, dt, DEV), bits, 0, gs, to_torch(scale, dt, DEV), tshift)
    assert got.dtype == torch.uint8 and tuple(got.shape) == want.shape
    np.testing.assert_array_equal(to_numpy(got), want)
    from optimum_quanto_amd.tensor.packing import PackedTensor, pack_weights

    packed = pack_weights(got, bits)
    np.testing.assert_array_equal(to_numpy(packed), O.pack_weights(want, bits))
    pt = PackedTensor.pack(got, bits)
    np.testing.assert_array_equal(to_numpy(pt.unpack()), want)
    # the one-pass kernel (quantize + pack, what quantize_weight / freeze use on the device) writes the same bytes
    fused = quanto_hip.lib.quantize_affine_packed(to_torch(w, dt, DEV), bits, gs, to_torch(scale, dt, DEV), tshift)
    np.testing.assert_array_equal(to_numpy(fused), O.pack_weights(want, bits))


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [2, 4])
@pytest.mark.parametrize("shape", [(10,), (12,), (10, 10), (12, 10), (32, 32), (7, 5), (1, 3), (256, 128)])
def test_hip_pack_golden(golden, bits, shape):
    """tests/tensor/test_packed_tensor.py:24-36 shapes (incl. first dims that are not a multiple of 8/bits), on the device."""
    from optimum_quanto_amd.tensor.packing import pack_weights

    key = f"pack/b{bits}/" + "x".join(map(str, shape))
    a = torch.from_numpy(golden[key + "/a"]).to(DEV)
    np.testing.assert_array_equal(to_numpy(pack_weights(a, bits)), golden[key + "/packed"])


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["fp16", "bf16"])
@pytest.mark.parametrize("wname", ["qint4", "qint2"])
def test_device_side_freeze_matches_cpu_freeze(wname, dt):
    """quantize_weight on the device (fused affine quantize + pack) produces the very bytes the CPU path produces."""
    torch.manual_seed(3)
    w = (torch.randn(96, 256) * 0.05).to(TORCH_DT[dt])
    qt = Q.qtypes[wname]
    scale, shift = Q.MaxOptimizer()(w, qtype=qt, axis=0, group_size=128)
    cpu = Q.quantize_weight(w, qtype=qt, axis=0, scale=scale, shift=shift, group_size=128)
    dev = Q.quantize_weight(w.to(DEV), qtype=qt, axis=0, scale=scale.to(DEV), shift=shift.to(DEV), group_size=128)
    assert torch.equal(dev._data._data.cpu(), cpu._data._data)
    assert torch.equal(dev.dequantize().cpu(), cpu.dequantize())


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("kind", ["int8", "e4m3fn", "e5m2"])
@pytest.mark.parametrize("shape", [(4, 37), (1, 4096), (33, 1000), (5,), (0, 16)])
def test_dequantize_symmetric_one_pass_is_bit_identical_gpu(dt, kind, shape):
    """r6: QBytesTensor.dequantize() with a per-tensor scale on the device = one kernel (csrc/quantize.hip: dequantize_symmetric) - bit-identical to the
    reference's `scale * data.to(dtype)` (tensor/qbytes.py:23-36) computed on the CPU, for every 8-bit format, every float dtype, ragged sizes and an empty tensor;
    all 256 byte values appear."""
    from optimum_quanto_amd.library.hip import quanto_hip

    tdt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[dt]
    n = int(np.prod(shape))
    codes = (np.arange(n, dtype=np.int64) * 7 % 256).astype(np.uint8).reshape(shape)
    if kind != "int8":  # NaN patterns compare unequal: keep the finite codes
        nan = (codes & 0x7F) == 0x7F if kind == "e4m3fn" else (codes & 0x7F) > 0x7C
        codes = np.where(nan, 0x38, codes).astype(np.uint8)
    qdt = {"int8": torch.int8, "e4m3fn": torch.float8_e4m3fn, "e5m2": torch.float8_e5m2}[kind]
    data = torch.from_numpy(codes).view(qdt)
    scale = torch.tensor([0.0371], dtype=tdt)
    want = scale * data.to(tdt)
    got = quanto_hip.lib.dequantize_symmetric(data.cuda(), scale.cuda())
    assert got is not None and got.dtype == tdt and got.shape == want.shape
    assert torch.equal(got.cpu(), want)
    # through the tensor class: a quantized activation on the device
    if n:
        qt = Q.ActivationQBytesTensor(Q.qint8 if kind == "int8" else getattr(Q, "qfloat8_" + kind), data.shape, data.stride(), data.cuda(), scale.cuda())
        assert torch.equal(qt.dequantize().cpu(), want)
