"""W4A8 (r6, csrc/qbits_a8_fused.hip): F.linear(quantized activation, int4 weight) on the 8-bit matrix instructions - the activation x weight combination
of the reference's tests/tensor/ops/test_linear_dispatch.py:22-42 that every backend of the reference serves by dequantizing the activation first.

Gates: int8 activations, unsplit form: BIT-EXACT against the oracle's restatement of the kernel's arithmetic (exact integer group sums, fp32 fma chain:
oracle.qbits_mm_a8_chain); every form (split-K, fp8 activations): the library's exact-math gate (helpers.assert_close_to_exact) against the float64 product of
the stored values; module level: the reference's own tolerance against its dequantize-first result.
"""
import numpy as np
import pytest
import torch

import optimum_quanto_amd as Q
from optimum_quanto_amd.library.hip import QuantoHipError, quanto_hip
from oracle import quanto_oracle as O

from helpers import assert_close_to_exact, assert_close_with_bias, fp8_tensor, make_qbits_problem, to_numpy, to_torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _act_int8(M, K, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    sx = O.round_to(np.array([0.0173 + 0.001 * (seed % 7)], np.float32), "bf16")
    return a, sx


def _run(p, a_t, sx, dt, bias=None):
    shift = torch.from_numpy(p["shift"]).to(DEV) if p["shift"].dtype == np.uint8 else to_torch(p["shift"], dt, DEV)
    y = quanto_hip.lib.qbits_mm_a8(a_t, to_torch(sx, dt, DEV), torch.from_numpy(p["packed"]).to(DEV), to_torch(p["scale"], dt, DEV), shift,
                                   None if bias is None else to_torch(bias, dt, DEV), 4, 128, p["N"], p["K"])
    return to_numpy(y)


@pytest.mark.parametrize("bm", ["64", "128"])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("zp", [False, True])
@pytest.mark.parametrize("M,N,K", [(65, 128, 256), (200, 264, 1024), (1, 8, 128), (300, 520, 384), (128, 4096, 4096)])
def test_w4a8_int8_bit_exact(monkeypatch, bm, dt, zp, M, N, K):
    """Both token tiles, ragged M and N, 1 .. 32 groups, float shifts and integer zero-points, with a bias: every output element identical to the fp32
    fma chain over the exact integer group sums."""
    monkeypatch.setenv("QUANTO_HIP_A8_BM", bm)
    monkeypatch.setenv("QUANTO_HIP_A8_SPLIT", "1")
    p = make_qbits_problem(2, N, K, dt, zeropoint=zp, seed=M + N + K)
    a, sx = _act_int8(M, K, seed=M + K)
    sx = O.round_to(sx, dt)
    bias = O.round_to(np.random.default_rng(3).standard_normal(N).astype(np.float32), dt)
    y = _run(p, torch.from_numpy(a).to(DEV), sx, dt, bias)
    assert quanto_hip.lib.last_kernel() == "a8_fused_int8"
    want = O.qbits_mm_a8_chain(a, sx, p["packed"], 4, p["scale"], p["shift"], 128, N, K, dt, bias)
    np.testing.assert_array_equal(y, want)


@pytest.mark.parametrize("split", ["0", "2", "4"])
@pytest.mark.parametrize("M,N,K", [(96, 256, 2048), (130, 1024, 4096), (512, 4096, 4096)])
def test_w4a8_int8_split_k_exact_math_and_deterministic(monkeypatch, split, M, N, K):
    """The K split (partial tiles through the workspace, last arriver adds in split order): exact-math gate, and two runs give the same bits."""
    monkeypatch.setenv("QUANTO_HIP_A8_SPLIT", split)
    p = make_qbits_problem(2, N, K, "bf16", seed=N + K)
    a, sx = _act_int8(M, K, seed=M)
    ta = torch.from_numpy(a).to(DEV)
    y = _run(p, ta, sx, "bf16")
    assert_close_to_exact(y, O.qbits_mm_a8_exact(a, sx, p["packed"], 4, p["scale"], p["shift"], 128, N, K), "bf16", f"w4a8 int8 split {split} {M}x{K}x{N}")
    for _ in range(3):
        np.testing.assert_array_equal(_run(p, ta, sx, "bf16"), y)


@pytest.mark.parametrize("bm", ["64", "128"])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("zp", [False, True])
@pytest.mark.parametrize("M,N,K", [(65, 128, 256), (200, 264, 1024), (300, 520, 384), (512, 1024, 4096)])
def test_w4a8_fp8_activations(monkeypatch, bm, dt, zp, M, N, K):
    """e4m3 activations x nibbles on the K = 128 MX-format matrix instruction (the nibbles become e4m3 codes through a byte table): every e4m3 value
    times an integer below 16 is exact - float64 gate on the stored values."""
    monkeypatch.setenv("QUANTO_HIP_A8_BM", bm)
    p = make_qbits_problem(2, N, K, dt, zeropoint=zp, seed=M + N)
    rng = np.random.default_rng(M + K)
    codes = O.fp8_encode((rng.standard_normal((M, K)) * 40).astype(np.float32), "e4m3fn")
    sx = O.round_to(np.array([0.021], np.float32), dt)
    y = _run(p, fp8_tensor(codes, "e4m3fn", DEV), sx, dt)
    assert quanto_hip.lib.last_kernel() == "a8_fused_fp8"
    want = O.qbits_mm_a8_exact(O.fp8_decode(codes, "e4m3fn"), sx, p["packed"], 4, p["scale"], p["shift"], 128, N, K)
    assert_close_to_exact(y, want, dt, f"w4a8 fp8 {M}x{K}x{N}")


def test_w4a8_every_nibble_and_every_e4m3_code():
    """The byte table: all 16 nibble values against all 254 finite e4m3 codes, one group - exact products, so the float64 gate is an identity check of the
    table (a wrong entry is off by at least one e4m3 ulp of a weight = 12 %)."""
    N, K, M = 16, 128, 256
    q = ((np.arange(K, dtype=np.int64)[None, :] + np.arange(N, dtype=np.int64)[:, None]) % 16).astype(np.uint8)  # every nibble value in every feature row
    packed = O.pack_weights(O.group(q, 0, 128), 4)
    scale = O.round_to(np.full((N, 1), 2.0**-6, np.float32), "fp16")
    shift = O.round_to(np.full((N, 1), 2.0**-7, np.float32), "fp16")
    codes = np.array([c for c in range(256) if (c & 0x7F) != 0x7F], dtype=np.uint8)
    a = np.resize(codes, (M, K)).astype(np.uint8)
    sx = np.array([1.0], np.float32)
    y = quanto_hip.lib.qbits_mm_a8(fp8_tensor(a, "e4m3fn", DEV), to_torch(sx, "fp16", DEV), torch.from_numpy(packed).to(DEV), to_torch(scale, "fp16", DEV),
                                   to_torch(shift, "fp16", DEV), None, 4, 128, N, K)
    want = O.qbits_mm_a8_exact(O.fp8_decode(a, "e4m3fn"), sx, packed, 4, scale, shift, 128, N, K)
    y = to_numpy(y).astype(np.float64)
    assert np.isfinite(y).all()
    assert np.abs(y - want).max() <= np.abs(want).max() * 2.0**-10


def test_w4a8_4096_cubed_bit_exact():
    """The bench shape, whole output, int8 activations: bit-exact (1024 tiles: the plan does not split K)."""
    M = N = K = 4096
    p = make_qbits_problem(2, N, K, "bf16", seed=7)
    a, sx = _act_int8(M, K, seed=11)
    y = _run(p, torch.from_numpy(a).to(DEV), sx, "bf16")
    np.testing.assert_array_equal(y, O.qbits_mm_a8_chain(a, sx, p["packed"], 4, p["scale"], p["shift"], 128, N, K, "bf16"))


def test_w4a8_formats_outside_the_kernel_are_refused_by_the_c_entry():
    """int2, other group sizes, per-channel scales: ENOTSUP from the C entry (the op keeps the reference's dequantize-first sequence for them)."""
    p = make_qbits_problem(2, 64, 256, "bf16", group_size=64, seed=1)
    a, sx = _act_int8(70, 256, seed=2)
    with pytest.raises(QuantoHipError):
        quanto_hip.lib.qbits_mm_a8(torch.from_numpy(a).to(DEV), to_torch(sx, "bf16", DEV), torch.from_numpy(p["packed"]).to(DEV), to_torch(p["scale"], "bf16", DEV),
                                   to_torch(p["shift"], "bf16", DEV), None, 4, 64, 64, 256)


@pytest.mark.parametrize("act", ["qint8", "qfloat8"])
@pytest.mark.parametrize("tokens,kernel", [(300, "a8"), (8, "other")])
def test_linear_dispatch_quantized_activation_int4_weight(act, tokens, kernel):
    """tests/tensor/ops/test_linear_dispatch.py:22-42 in the reference (activation qint8 / qfloat8 x weight qint4): F.linear keeps the activation quantized
    from 64 rows on and agrees with the dequantize-first product within the reference's tolerance."""
    torch.manual_seed(0)
    w = (torch.randn(512, 1024, device=DEV) * 0.02).to(torch.bfloat16)
    scale, shift = Q.MaxOptimizer()(w, Q.qint4, 0, 128)
    qw = Q.quantize_weight(w, Q.qint4, 0, scale, shift, group_size=128)
    x = torch.randn(tokens, 1024, device=DEV, dtype=torch.bfloat16)
    aq = getattr(Q, "qfloat8_e4m3fn" if act == "qfloat8" else "qint8")
    from optimum_quanto_amd.tensor.activations import absmax_scale, quantize_activation
    xq = quantize_activation(x, aq, absmax_scale(x, aq))
    y = torch.nn.functional.linear(xq, qw)
    name = quanto_hip.lib.last_kernel()
    assert name.startswith("a8_fused") if kernel == "a8" else not name.startswith("a8_fused")
    want = torch.matmul(xq.dequantize().float(), qw.dequantize().float().t())
    err = ((y.float() - want).abs().max() / want.abs().max()).item()
    assert err < 2e-2, err  # helpers.assert_similar's bound for cuda in the reference
