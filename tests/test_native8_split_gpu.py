"""K split of the quantized-activation GEMM (r6, csrc/qmm_native8.hip): S workgroups per output tile, partial accumulator tiles through the workspace, the
last arriver adds them in split order.  int8 x int8: int32 partial sums - the split result is the unsplit one, bit for bit (library/qbytes_mm.py:36-50 through
_int_mm); fp8 x fp8: fp32 partials in a fixed order - exact-math gate and run-to-run identical bits.  The workspace contract (counters zero on entry and on
exit) is checked by running different shapes back to back on the same scratch buffer.
"""
import numpy as np
import pytest
import torch

from optimum_quanto_amd.library.hip import quanto_hip
from oracle import quanto_oracle as O

from helpers import assert_close_to_exact, fp8_tensor, to_numpy, to_torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _int8_problem(M, N, K, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    b = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
    s = O.round_to(((rng.random((N, 1)) + 0.5) / 1e4).astype(np.float32), "bf16")
    return a, b, s


@pytest.mark.parametrize("small", ["0", "1"])
@pytest.mark.parametrize("split", ["2", "4", "8"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 3072), (300, 700, 6144), (129, 257, 1536), (520, 512, 12288)])
def test_native8_split_k_int8_bit_exact(monkeypatch, small, split, M, N, K):
    """Forced splits of 2 .. 8 over 256- and 128-tiles, ragged M / N, 2 .. 48 pairs per K range, with a bias: identical to the exact integer reference."""
    a, b, s = _int8_problem(M, N, K, M + N + K)
    bias = O.round_to(np.random.default_rng(5).standard_normal(N).astype(np.float32), "bf16")
    ta, tb, ts, tbias = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), to_torch(s, "bf16", DEV), to_torch(bias, "bf16", DEV)
    monkeypatch.setenv("QUANTO_HIP_NATIVE8_SMALL", small)
    monkeypatch.setenv("QUANTO_HIP_NATIVE8_SPLIT", split)
    y = quanto_hip.lib.qbytes_mm(ta, tb, ts, tbias, kernel="mfma_native8")
    assert quanto_hip.lib.last_kernel() == "mfma_native8"
    monkeypatch.setenv("QUANTO_HIP_NATIVE8_SPLIT", "1")
    assert torch.equal(y, quanto_hip.lib.qbytes_mm(ta, tb, ts, tbias, kernel="mfma_native8"))
    # library/qbytes_mm.py:47-50 + the reference's `+ bias` on the rounded product
    np.testing.assert_array_equal(to_numpy(y), O.round_to(O.qbytes_int_mm_ref(a, b, s, "bf16").astype(np.float32) + bias.reshape(1, -1), "bf16"))


@pytest.mark.parametrize("kind", ["e4m3fn", "e5m2"])
@pytest.mark.parametrize("small,split", [("0", "2"), ("0", "4"), ("1", "4"), ("1", "8")])
def test_native8_split_k_fp8_exact_math_and_deterministic(monkeypatch, kind, small, split):
    M, N, K = 300, 700, 4096
    rng = np.random.default_rng(17)
    a = O.fp8_encode(rng.standard_normal((M, K)).astype(np.float32), kind)
    b = O.fp8_encode(rng.standard_normal((N, K)).astype(np.float32), kind)
    s = O.round_to(((rng.random((N, 1)) + 0.5) / 1e2).astype(np.float32), "bf16")
    ta, tb, ts = fp8_tensor(a, kind, DEV), fp8_tensor(b, kind, DEV), to_torch(s, "bf16", DEV)
    monkeypatch.setenv("QUANTO_HIP_NATIVE8_SMALL", small)
    monkeypatch.setenv("QUANTO_HIP_NATIVE8_SPLIT", split)
    y = quanto_hip.lib.qbytes_mm(ta, tb, ts, kernel="mfma_native8")
    want = np.matmul(O.fp8_decode(a, kind).astype(np.float64), O.fp8_decode(b, kind).astype(np.float64).T) * s.astype(np.float64).reshape(1, -1)
    assert_close_to_exact(to_numpy(y), want, "bf16", f"fp8 x fp8 split {split} small {small}")
    for _ in range(3):
        assert torch.equal(y, quanto_hip.lib.qbytes_mm(ta, tb, ts, kernel="mfma_native8"))


@pytest.mark.parametrize("ticks", ["0", "50"])
@pytest.mark.parametrize("small,split", [("0", "4"), ("1", "8"), ("0", "2")])
def test_native8_split_k_abandoned_slices_are_finished_by_the_last_arriver(monkeypatch, ticks, small, split):
    """The tail's fallback: with a poll limit of 0 (or 0.5 us) workgroups do not wait for their partners - every slice whose owner saw fewer than S arrivals is marked
    abandoned and reduced by the tile's last arriver.  Same bits as the unsplit kernel, state words left zero (the next call on the same scratch works)."""
    M, N, K = 520, 1030, 4096
    a, b, s = _int8_problem(M, N, K, 77)
    ta, tb, ts = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), to_torch(s, "bf16", DEV)
    want = O.qbytes_int_mm_ref(a, b, s, "bf16")
    monkeypatch.setenv("QUANTO_HIP_NATIVE8_SMALL", small)
    monkeypatch.setenv("QUANTO_HIP_NATIVE8_SPLIT", split)
    monkeypatch.setenv("QUANTO_HIP_NATIVE8_POLL_TICKS", ticks)
    for _ in range(3):
        np.testing.assert_array_equal(to_numpy(quanto_hip.lib.qbytes_mm(ta, tb, ts, kernel="mfma_native8")), want)
    monkeypatch.setenv("QUANTO_HIP_NATIVE8_POLL_TICKS", "20000")
    np.testing.assert_array_equal(to_numpy(quanto_hip.lib.qbytes_mm(ta, tb, ts, kernel="mfma_native8")), want)


def test_native8_split_k_auto_plan_long_k_int8_bit_exact_and_workspace_left_clean():
    """A Llama-3 down-projection at 512 tokens, (512,4096,14336), with int8 operands: AUTO splits the 128 128-tiles four ways (tests/test_host_cpu.py states the
    plan); whole output bit-exact; then other shapes reuse the same scratch buffer (its state words must have been left zero)."""
    M, N, K = 512, 4096, 14336
    a, b, s = _int8_problem(M, N, K, 3)
    ta, tb, ts = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), to_torch(s, "bf16", DEV)
    assert quanto_hip.cdll.quanto_hip_qbytes_mm_workspace_size(M, N, K, 3, 3, 2, 0) > 0
    y = quanto_hip.lib.qbytes_mm(ta, tb, ts)
    assert quanto_hip.lib.last_kernel() == "mfma_native8"
    np.testing.assert_array_equal(to_numpy(y), O.qbytes_int_mm_ref(a, b, s, "bf16"))
    a2, b2, s2 = _int8_problem(384, 1024, 4096, 4)
    for _ in range(2):
        y2 = quanto_hip.lib.qbytes_mm(torch.from_numpy(a2).to(DEV), torch.from_numpy(b2).to(DEV), to_torch(s2, "bf16", DEV))
        np.testing.assert_array_equal(to_numpy(y2), O.qbytes_int_mm_ref(a2, b2, s2, "bf16"))
    assert torch.equal(y, quanto_hip.lib.qbytes_mm(ta, tb, ts))


def test_native8_split_k_auto_plan_fp8():
    """(256,8192,8192) with fp8 activations and weights - AUTO splits two ways -, whole output against float64 math, run-to-run identical bits."""
    M, N, K = 256, 8192, 8192
    rng = np.random.default_rng(21)
    a = O.fp8_encode(rng.standard_normal((M, K)).astype(np.float32), "e4m3fn")
    b = O.fp8_encode(rng.standard_normal((N, K)).astype(np.float32), "e4m3fn")
    s = O.round_to(((rng.random((N, 1)) + 0.5) / 1e2).astype(np.float32), "bf16")
    ta, tb, ts = fp8_tensor(a, "e4m3fn", DEV), fp8_tensor(b, "e4m3fn", DEV), to_torch(s, "bf16", DEV)
    y = quanto_hip.lib.qbytes_mm(ta, tb, ts)
    want = np.matmul(O.fp8_decode(a, "e4m3fn").astype(np.float64), O.fp8_decode(b, "e4m3fn").astype(np.float64).T) * s.astype(np.float64).reshape(1, -1)
    assert_close_to_exact(to_numpy(y), want, "bf16", "(256,8192,8192) fp8, AUTO split")
    assert torch.equal(y, quanto_hip.lib.qbytes_mm(ta, tb, ts))
