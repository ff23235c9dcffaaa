"""The reference's own device tests for this path, re-stated against this package (run with -m gpu).

Mirrors tests/tensor/weights/test_weight_qbits_tensor_dispatch.py:62-94, tests/tensor/weights/weight_helpers.py:19-37,
tests/nn/test_qlinear.py:116-135,179-225 and tests/library/test_unpack.py:22-30 of the reference: same construction
(random_qweight with Absmax/Max optimizers on the device), same assertions (assert_similar + relative max error < 2e-2 on
"cuda").  The fused kernels are compared with F.linear on the dequantized weight, exactly as upstream does.
"""
import io

import pytest
import torch

import optimum_quanto_amd as Q
from optimum_quanto_amd.library.hip import quanto_hip

from helpers import assert_similar

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def random_tensor(shape, dtype=torch.float32, device="cpu"):
    rand_dtype = dtype if dtype.itemsize > 1 else torch.float16
    return (torch.rand(shape, dtype=rand_dtype, device=device) * 2 - 1).to(dtype)


def random_qweight(shape, qtype, dtype=torch.float32, axis=0, group_size=None, device="cpu"):
    t = random_tensor(shape, dtype, device=device)
    if qtype.bits == 8:
        scale, shift = Q.AbsmaxOptimizer()(t, qtype=qtype, axis=axis), None
    else:
        scale, shift = Q.MaxOptimizer()(t, qtype=qtype, axis=axis, group_size=group_size)
    return Q.quantize_weight(t, qtype=qtype, axis=axis, scale=scale, shift=shift, group_size=group_size, optimized=False)


def check_weight_qtensor_linear(qweight, batch_size, tokens, use_bias):
    dtype, device = qweight.dtype, qweight.device
    out_features, in_features = qweight.shape
    inputs = torch.rand((batch_size, tokens, in_features), dtype=dtype, device=device)
    bias = random_tensor((out_features,), dtype=dtype, device=device) if use_bias else None
    qout = torch.nn.functional.linear(inputs, qweight, bias)
    out = torch.nn.functional.linear(inputs, qweight.dequantize(), bias)
    assert_similar(out, qout)
    rel_max_err = (out - qout).abs().max() / out.abs().max()
    assert rel_max_err < 2e-2, f"relative max error {float(rel_max_err):.4f}"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("batch_size,tokens", [(1, 16), (2, 32), (1, 48), (2, 64), (1, 1), (4, 512)])
@pytest.mark.parametrize("in_features", [1024, 4096, 16384])
@pytest.mark.parametrize("out_features", [1024, 4096])
@pytest.mark.parametrize("use_bias", [True, False], ids=["bias", "no-bias"])
def test_weight_qbits_tensor_linear_gpu(dtype, batch_size, tokens, in_features, out_features, use_bias):
    qbt = random_qweight((out_features, in_features), Q.qint4, dtype, group_size=128, device=DEV)
    check_weight_qtensor_linear(qbt, batch_size, tokens, use_bias)
    assert quanto_hip.lib.last_kernel() in ("gemv", "mmv", "skinny", "mfma", "mfma_fused4", "dequant_mfma")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("qtype", ["qint8", "qfloat8_e4m3fn", "qfloat8_e5m2", "qfloat8_e4m3fnuz"])
@pytest.mark.parametrize("batch_size,tokens", [(1, 1), (1, 16), (2, 64), (8, 256)])
@pytest.mark.parametrize("use_bias", [True, False], ids=["bias", "no-bias"])
def test_weight_qbytes_tensor_linear_gpu(dtype, qtype, batch_size, tokens, use_bias):
    qbt = random_qweight((2048, 1024), Q.qtypes[qtype], dtype, device=DEV)
    check_weight_qtensor_linear(qbt, batch_size, tokens, use_bias)


@pytest.mark.parametrize("weights", ["qint4", "qint8", "qfloat8"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_move_and_serialize_qlinear(weights, dtype):
    """tests/nn/test_qlinear.py:179-225: quantize on CPU, move to the device, state_dict round trip on the device."""
    linear = torch.nn.Linear(1024, 1024).to(dtype)
    qlinear = Q.QLinear.from_module(linear, weights=Q.qtypes[weights])
    qlinear.freeze()
    qlinear.to(DEV)
    inner = ["_data", "_scale"] + (["_shift"] if weights == "qint4" else [])
    for name in inner:
        assert getattr(qlinear.weight, name).device.type == "cuda"
    x = random_tensor((2, 8, 1024), dtype=dtype, device=DEV)
    with torch.no_grad():
        y = qlinear(x)
    buf = io.BytesIO()
    torch.save(qlinear.state_dict(), buf)
    buf.seek(0)
    state = torch.load(buf, weights_only=False)
    again = Q.QLinear(1024, 1024, weights=Q.qtypes[weights], dtype=dtype, device=DEV)
    again.load_state_dict(state)
    assert again.frozen and again.weight.qtype == qlinear.weight.qtype and again.weight.device.type == "cuda"
    with torch.no_grad():
        assert torch.equal(again(x), y)


@pytest.mark.parametrize("group_size", [None, 128], ids=["channel-wise", "group-wise"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16], ids=["fp32", "fp16", "bf16"])
def test_qbitstensor_to_device_dequantize_equal(dtype, group_size):
    """tests/tensor/weights/test_weight_qbits_tensor_dispatch.py:23-42: dequantize() is bit-identical across devices."""
    qa = random_qweight((256, 512), Q.qint4, dtype, group_size=group_size, device="cpu")
    dqa = qa.dequantize()
    moved = qa.to(DEV)
    assert isinstance(moved, Q.QBitsTensor) and moved._data.device.type == "cuda"
    assert torch.equal(moved.dequantize().cpu(), dqa)
