"""fp32 activations on the fast kernels (r6, csrc/qmm_f32.hip): a model quantized without a dtype cast - what the reference's own
tests/nn/test_qlinear.py:116-135 runs - computes in float32.  Until r6 such calls took the one-thread-per-output kernels.

Gate: relative Frobenius AND relative max error <= 1e-5 against exact (float64) math on the same integers / scales (the r5 review's bound); the
tile kernel multiplies the reference's own fp32 dequantized weight (tensor/qbits.py:27-49), checked against the oracle's dequantize as well.
"""
import numpy as np
import pytest
import torch

import optimum_quanto_amd as Q
from optimum_quanto_amd.library.hip import quanto_hip
from oracle import quanto_oracle as O

from helpers import fp8_tensor, make_qbits_problem, make_qbytes_problem, to_numpy, to_torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-5


def _close(y, want, what):
    y = np.asarray(y, np.float64)
    fro, mx = O.rel_fro(y, want), O.rel_max(y, want)
    assert fro <= TOL and mx <= TOL, f"{what}: rel_fro={fro:.3e} rel_max={mx:.3e}"


def _qbits(p, kernel="auto", bias=None):
    shift = torch.from_numpy(p["shift"]).to(DEV) if p["shift"].dtype == np.uint8 else to_torch(p["shift"], "fp32", DEV)
    return to_numpy(quanto_hip.lib.qbits_mm(to_torch(p["x"], "fp32", DEV), torch.from_numpy(p["packed"]).to(DEV), to_torch(p["scale"], "fp32", DEV), shift,
                                            None if bias is None else to_torch(bias, "fp32", DEV), p["bits"], p["group_size"], p["N"], p["K"], kernel=kernel))


def _exact(p, bias=None):
    return O.qbits_mm_exact(p["x"], p["packed"], p["bits"], p["scale"], p["shift"], p["group_size"], p["N"], p["K"], bias)


@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("N,K", [(256, 1024), (512, 4096), (64, 2048), (34, 11008), (96, 14336), (256, 128), (4096, 4096)])
def test_qbits_gemv_f32(M, N, K):
    p = make_qbits_problem(M, N, K, "fp32", seed=M * 7 + N)
    y = _qbits(p)
    assert quanto_hip.lib.last_kernel() == "gemv_f32"
    _close(y, _exact(p), f"fp32 gemv {M}x{K}x{N}")


@pytest.mark.parametrize("bits,gs,zp", [(4, 64, False), (4, 32, True), (4, 96, False), (4, None, False), (2, 128, False), (2, 64, True), (2, None, False), (4, 128, True)])
@pytest.mark.parametrize("M", [1, 3, 4, 7])
def test_qbits_gemv_f32_formats(bits, gs, zp, M):
    """Every format nn/qmodule.py:121-129 can select (group sizes 128 / 96 / 64 / 32, per-channel), qint2, float shifts and integer zero-points."""
    N, K = 192, 1152 if gs == 96 else 2048
    p = make_qbits_problem(M, N, K, "fp32", bits=bits, group_size=gs, zeropoint=zp, seed=bits + M + (gs or 0))
    bias = np.random.default_rng(1).standard_normal(N).astype(np.float32)
    y = _qbits(p, bias=bias)
    assert quanto_hip.lib.last_kernel() == "gemv_f32"
    _close(y, _exact(p) + bias.astype(np.float64), f"fp32 gemv int{bits} g{gs} zp={zp} M={M}")


@pytest.mark.parametrize("bits,gs,zp", [(4, 128, False), (4, 64, True), (4, 32, False), (4, None, False), (2, 128, False), (2, 64, True)])
@pytest.mark.parametrize("M,N,K", [(9, 256, 1024), (100, 512, 4096), (300, 200, 2048), (129, 68, 1152), (1024, 1024, 1024)])
def test_qbits_mm_f32_tiles(bits, gs, zp, M, N, K):
    """The fp32 MFMA tile kernel (v_mfma_f32_16x16x4_f32): ragged M / N, int4 and int2 planes, every group size it takes - against exact math and
    against the float64 product with the reference's own fp32 dequantized weight (the operand the kernel builds while staging)."""
    if gs is not None and K % gs:
        pytest.skip("group size does not divide K")
    N = N - N % (8 // bits)
    p = make_qbits_problem(M, N, K, "fp32", bits=bits, group_size=gs, zeropoint=zp, seed=M + N + bits)
    bias = np.random.default_rng(2).standard_normal(N).astype(np.float32)
    y = _qbits(p, bias=bias)
    assert quanto_hip.lib.last_kernel() == "mfma_f32"
    _close(y, _exact(p) + bias.astype(np.float64), f"fp32 tiles int{bits} g{gs} zp={zp} {M}x{K}x{N}")
    w = O.dequantize_qbits_ref(p["packed"], bits, p["scale"], p["shift"], 0, gs, (N, K), "fp32").astype(np.float64)
    _close(y, np.matmul(p["x"].astype(np.float64), w.T) + bias.astype(np.float64), "fp32 tiles vs the reference's dequantized weight")


def test_qbits_f32_odd_shapes_keep_the_general_kernel():
    """K that is not a multiple of 16 (or a group size that is not) stays on the one-thread-per-output kernel - correct, just not fast."""
    p = make_qbits_problem(3, 64, 200, "fp32", group_size=40, seed=5)
    y = _qbits(p)
    assert quanto_hip.lib.last_kernel() == "naive"
    _close(y, _exact(p), "fp32 odd K")


@pytest.mark.parametrize("kind", [None, "e4m3fn", "e5m2", "e4m3fnuz"])
@pytest.mark.parametrize("M,N,K", [(1, 1024, 1024), (2, 300, 4096), (8, 64, 11008), (9, 256, 1024), (100, 520, 4096), (300, 129, 2048)])
def test_qbytes_f32(kind, M, N, K):
    """BASELINE configs[0] ((1,1024,1024) int8 weights, fp32 activations) and its siblings on the device: the fp32 weight stream up to 8 rows, fp32
    MFMA tiles beyond."""
    mk = "e4m3fn" if kind == "e4m3fnuz" else kind
    q = make_qbytes_problem(M, N, K, "fp32", mk, seed=M + N)
    data = q["data"]
    if kind == "e4m3fnuz":  # every byte value, NaN pattern excluded
        data = np.where(data == 0x80, 0, data).astype(np.uint8)
    tb = torch.from_numpy(data).to(DEV) if kind is None else fp8_tensor(data, kind, DEV)
    bias = np.random.default_rng(3).standard_normal(N).astype(np.float32)
    y = to_numpy(quanto_hip.lib.qbytes_mm(to_torch(q["x"], "fp32", DEV), tb, to_torch(q["scale"], "fp32", DEV), to_torch(bias, "fp32", DEV)))
    assert quanto_hip.lib.last_kernel() == ("gemv_f32" if M <= 8 else "mfma_f32")
    _close(y, O.qbytes_mm_exact(q["x"], data, q["scale"], kind) + bias.astype(np.float64), f"fp32 qbytes {kind} {M}x{K}x{N}")


@pytest.mark.parametrize("weights", ["qint4", "qint8", "qfloat8"])
@pytest.mark.parametrize("tokens", [1, 33])
def test_default_dtype_qlinear_on_device(weights, tokens):
    """tests/nn/test_qlinear.py:116-135 in the reference: a Linear in PyTorch's default dtype, quantized and frozen, run on the device - the call now
    reaches the fp32 kernels instead of the one-thread-per-output fallback, and agrees with the float product of the dequantized weight."""
    torch.manual_seed(0)
    lin = torch.nn.Linear(1024, 512).to(DEV)
    q = Q.QLinear.from_module(lin, weights=getattr(Q, weights))
    q.freeze()
    x = torch.randn(tokens, 1024, device=DEV)
    y = q(x)
    assert y.dtype == torch.float32
    assert quanto_hip.lib.last_kernel() == ("gemv_f32" if tokens <= 8 else "mfma_f32")
    want = torch.nn.functional.linear(x.double(), q.weight.dequantize().double(), q.bias.double())
    assert ((y.double() - want).norm() / want.norm()).item() < 1e-5
