"""Seeded shape fuzz: whatever kernel AUTO selects must agree with the naive one-thread-per-output kernel of the same library
on ragged / odd shapes (dispatch boundaries, clamped loads, masked stores, split-K and pass logic).  The naive kernels are
themselves pinned against the oracle in test_hip_parity.py; this test only looks for shape-dependent bugs, so it stays on the
device and runs a few hundred shapes in seconds."""
import numpy as np
import pytest
import torch

from optimum_quanto_amd.library.hip import quanto_hip

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _close(y, ref, dt):
    y, ref = y.float(), ref.float()
    scale = ref.abs().max().clamp_min(1e-6)
    tol = 2.5e-2 if dt == torch.bfloat16 else 4e-3  # two independently rounded 16-bit results of the same exact sums
    err = ((y - ref).abs().max() / scale).item()
    assert err < tol, f"max rel err {err:.3e}"
    assert torch.isfinite(y).all()


def _shapes(n, seed, kmult, kmax):
    rng = np.random.default_rng(seed)
    ms = [1, 2, 3, 4, 5, 8, 9, 16, 17, 33, 64, 65, 100, 128, 129, 255, 256, 257, 300, 513]
    out = []
    for _ in range(n):
        M = int(rng.choice(ms))
        N = int(rng.choice([1, 2, 15, 16, 17, 48, 63, 64, 65, 100, 128, 130, 255, 256, 384, 500, 512, 1000, 1024]))
        K = int(rng.integers(1, kmax // kmult + 1)) * kmult
        out.append((M, N, K))
    return out


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("wkind", ["i8", "e4m3"])
def test_qbytes_auto_matches_naive_on_random_shapes(dt, wkind):
    g = torch.Generator().manual_seed(11)
    lib = quanto_hip.lib
    seen = set()
    for (M, N, K) in _shapes(70, 5, 16, 2304):
        x = torch.randn((M, K), generator=g).to(dt).to(DEV)
        if wkind == "i8":
            w = torch.randint(-128, 128, (N, K), dtype=torch.int8, generator=g).to(DEV)
        else:
            w = (torch.randn((N, K), generator=g) * 50).clamp(-448, 448).to(torch.float8_e4m3fn).to(DEV)
        s = (torch.rand((N, 1), generator=g) * 1e-2 + 1e-3).to(dt).to(DEV)
        bias = torch.randn((N,), generator=g).to(dt).to(DEV) if (M + N) % 3 == 0 else None
        ref = lib.qbytes_mm(x, w, s, bias, kernel="naive")
        y = lib.qbytes_mm(x, w, s, bias)
        seen.add(lib.last_kernel())
        try:
            _close(y, ref, dt)
        except AssertionError as e:
            raise AssertionError(f"qbytes {wkind} (M,N,K)=({M},{N},{K}) kernel={lib.last_kernel()}: {e}")
    assert {"gemv", "skinny"} <= seen, seen


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("zeropoint", [False, True], ids=["float-shift", "zero-point"])
def test_qbits_auto_matches_naive_on_random_shapes(dt, zeropoint):
    g = torch.Generator().manual_seed(12)
    lib = quanto_hip.lib
    seen = set()
    for (M, N, K) in _shapes(70, 6, 128, 2304):
        N = N + (N & 1)  # the fused kernels need an even N (two nibble planes); odd N is covered by the naive tests
        x = torch.randn((M, K), generator=g).to(dt).to(DEV)
        packed = torch.randint(0, 256, (N * K // 256, 128), dtype=torch.uint8, generator=g).to(DEV)
        scale = (torch.rand((N * K // 128, 1), generator=g) * 0.01 + 0.005).to(dt).to(DEV)
        if zeropoint:
            shift = torch.randint(0, 16, (N * K // 128, 1), dtype=torch.uint8, generator=g).to(DEV)
        else:
            shift = (torch.rand((N * K // 128, 1), generator=g) * 0.05 + 0.05).to(dt).to(DEV)
        bias = torch.randn((N,), generator=g).to(dt).to(DEV) if (M + N) % 3 == 0 else None
        ref = lib.qbits_mm(x, packed, scale, shift, bias, 4, 128, N, K, kernel="naive")
        y = lib.qbits_mm(x, packed, scale, shift, bias, 4, 128, N, K)
        seen.add(lib.last_kernel())
        try:
            _close(y, ref, dt)
        except AssertionError as e:
            raise AssertionError(f"qbits (M,N,K)=({M},{N},{K}) kernel={lib.last_kernel()}: {e}")
    assert {"gemv", "skinny", "dequant_mfma"} <= seen, seen


def test_quantized_activation_auto_matches_naive_on_random_shapes():
    g = torch.Generator().manual_seed(13)
    lib = quanto_hip.lib
    for (M, N, K) in _shapes(50, 7, 64, 2304):
        a = torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g).to(DEV)
        b = torch.randint(-128, 128, (N, K), dtype=torch.int8, generator=g).to(DEV)
        s = (torch.rand((N, 1), generator=g) * 1e-4 + 1e-5).to(torch.bfloat16).to(DEV)
        ref = lib.qbytes_mm(a, b, s, kernel="naive")
        y = lib.qbytes_mm(a, b, s)
        assert lib.last_kernel() == "mfma_native8"
        assert torch.equal(y, ref), f"int8 x int8 (M,N,K)=({M},{N},{K}) not bit-identical"
