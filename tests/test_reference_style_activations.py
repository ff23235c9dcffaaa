"""Scenarios of the reference's activation / calibration test-suite (tests/tensor/activations/test_activations_dispatch.py,
test_activations_quantize.py, tests/nn/test_calibrate.py, tests/nn/test_qlinear.py::*activations*) against this package's
host mirror - on the CPU in the default run and on the device under ``-m gpu``."""
import pytest
import torch

import optimum_quanto_amd as Q

from helpers import assert_similar, observed_activation_scales

DEVICES = [pytest.param("cpu", id="cpu"), pytest.param("cuda", id="cuda", marks=pytest.mark.gpu)]
F8 = [Q.qfloat8_e5m2, Q.qfloat8_e4m3fn]


def rand(shape, dtype=torch.float32, device="cpu"):
    return (torch.rand(shape, dtype=torch.float32) * 2 - 1).to(dtype).to(device)


def rand_qact(shape, qtype=Q.qint8, dtype=torch.float32, device="cpu"):
    t = rand(shape, dtype, device)
    return Q.quantize_activation(t, qtype=qtype, scale=Q.absmax_scale(t, qtype=qtype))


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16], ids=["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("qtype", [Q.qint8] + F8, ids=["qint8", "qfloat8_e5m2", "qfloat8_e4m3fn"])
@pytest.mark.parametrize("shape", [(32, 32), (32, 10, 32)])
def test_symmetric_quantize_activation_round_trip(shape, qtype, dtype, device):
    a = rand(shape, dtype, device)
    qa = Q.quantize_activation(a, qtype=qtype, scale=Q.absmax_scale(a, qtype))
    assert isinstance(qa, Q.ActivationQBytesTensor) and qa.dtype == dtype and qa.qtype == qtype
    assert qa.device.type == device and qa._data.dtype == qtype.dtype
    assert_similar(a, qa, atol=5e-3 if qtype == Q.qint8 else 5e-2)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("shape", [(10,), (1, 10), (10, 32, 32)])
@pytest.mark.parametrize("scalar", [1, 0.5, torch.tensor(0.12)], ids=["int", "float", "tensor"])
def test_mul_by_scalar_folds_into_the_scale(shape, scalar, device):
    qa = rand_qact(shape, device=device)
    if isinstance(scalar, torch.Tensor):
        scalar = scalar.to(device)
    for prod, ref in ((qa * scalar, qa.dequantize() * scalar), (scalar * qa, scalar * qa.dequantize())):
        assert isinstance(prod, Q.ActivationQBytesTensor)
        assert torch.equal(prod._data, qa._data)  # only the scale moved
        assert_similar(ref, prod)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("batch", [1, 10])
@pytest.mark.parametrize("tokens,emb", [(5, 5), (32, 32), (10, 32)])
def test_relu_and_softmax_stay_quantized(batch, tokens, emb, device):
    qa = rand_qact((batch, tokens, emb), device=device)
    r = torch.nn.functional.relu(qa)
    assert isinstance(r, Q.ActivationQBytesTensor)
    assert torch.equal(r._data, torch.clamp(qa._data, min=0))
    s = torch.nn.functional.softmax(qa, dim=-1)
    assert isinstance(s, Q.ActivationQBytesTensor)
    d = s.dequantize()
    assert d.min() >= 0 and d.max() <= 1


@pytest.mark.parametrize("device", DEVICES)
def test_shape_ops(device):
    qa = rand_qact((10, 32, 64), device=device)
    assert isinstance(qa.view((1, 10, 32, 64)), Q.ActivationQBytesTensor)
    tr = torch.transpose(qa, 1, 2)
    assert tr.qtype == qa.qtype and torch.equal(tr.dequantize(), torch.transpose(qa.dequantize(), 1, 2))
    q2 = rand_qact((4, 6), device=device)
    t2 = q2.t()
    assert t2.shape == (6, 4) and torch.equal(t2.dequantize(), q2.dequantize().t())
    other = Q.quantize_activation(rand((4, 6), device=device), qtype=q2.qtype, scale=q2._scale)
    cat = torch.cat([q2, other])
    assert isinstance(cat, Q.ActivationQBytesTensor)
    assert_similar(torch.cat([q2.dequantize(), other.dequantize()]), cat)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("batch", [1, 10])
@pytest.mark.parametrize("tokens,emb", [(32, 32), (10, 32)])
@pytest.mark.parametrize("use_bias", [True, False], ids=["bias", "no-bias"])
@pytest.mark.parametrize("activations", [Q.qint8] + F8, ids=["a-qint8", "a-qfloat8-e5m2", "a-qfloat8-e4m3"])
def test_calibrate_qlinear(batch, tokens, emb, use_bias, activations, device):
    linear = torch.nn.Linear(emb, emb, bias=use_bias).to(device)
    qlinear = Q.QLinear.from_module(linear, weights=Q.qint8, activations=activations)
    qin = rand_qact((batch, tokens, emb), qtype=activations, device=device)
    with torch.no_grad():
        qlinear(qin)
    assert torch.all(qlinear.input_scale == 1) and torch.all(qlinear.output_scale == 1)  # nothing calibrates outside the mode
    with torch.no_grad(), observed_activation_scales():
        qout = qlinear(qin)
    assert qout.qtype == activations
    assert torch.any(qlinear.input_scale != 1) and torch.any(qlinear.output_scale != 1)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("activations", [Q.qint8] + F8, ids=["a-qint8", "a-qfloat8-e5m2", "a-qfloat8-e4m3"])
def test_calibrate_two_chained_qlinears(activations, device):
    class Two(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.linear1, self.linear2 = torch.nn.Linear(32, 32), torch.nn.Linear(32, 32)

        def forward(self, x):
            return self.linear2(self.linear1(x))

    model = Two().to(device)
    model.linear1 = Q.QLinear.from_module(model.linear1, weights=Q.qint8, activations=activations)
    model.linear2 = Q.QLinear.from_module(model.linear2, weights=Q.qint8, activations=activations)
    with torch.no_grad(), observed_activation_scales():
        qout = model(rand_qact((1, 10, 32), qtype=activations, device=device))
    for m in (model.linear1, model.linear2):
        assert torch.any(m.input_scale != 1) and torch.any(m.output_scale != 1)
    assert qout.qtype == activations


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("batch", [1, 10])
@pytest.mark.parametrize("tokens,emb", [(32, 32), (10, 64)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("weights", [Q.qint4, Q.qint8], ids=["w-qint4", "w-qint8"])
@pytest.mark.parametrize("activations", [Q.qint8, Q.qfloat8_e4m3fn], ids=["a-qint8", "a-qfloat8-e4m3"])
def test_quantize_linear_with_activations(batch, tokens, emb, dtype, weights, activations, device):
    """tests/nn/test_qlinear.py::test_quantize_linear_*_activations: calibrated quantized outputs stay close to the float module."""
    torch.manual_seed(tokens + emb)
    linear = torch.nn.Linear(emb, emb).to(dtype).to(device)
    qlinear = Q.QLinear.from_module(linear, weights=weights, activations=activations)
    assert qlinear.qweight.qtype == weights
    x = rand((batch, tokens, emb), dtype, device)
    with torch.no_grad(), observed_activation_scales():
        qlinear(x)
    Q.freeze(qlinear)
    with torch.no_grad():
        qout = qlinear(x)
        out = linear(x)
    assert isinstance(qout, Q.ActivationQBytesTensor) and qout.qtype == activations
    # the reference's own gates for this scenario: 1e-1 for int8 activations, 2e-1 for float8, on top of int4 weight noise
    atol = {Q.qint8: 1.2e-1, Q.qfloat8_e4m3fn: 2.5e-1}[activations] * (2 if weights == Q.qint4 else 1)
    assert_similar(out, qout, atol=atol)
