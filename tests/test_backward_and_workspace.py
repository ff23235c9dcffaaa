"""Gradients through quantized modules (the reference's straight-through behaviour, tensor/function.py:49-63 and the
qfallback convolution) and the split-K workspace contract (include/quanto_hip.h: one fixed counter region shared by every
split-K kernel, partial sums always behind it).

The reference has backward tests for QLinear / QConv2d in tests/nn/test_qlinear.py:228-262 and tests/nn/test_qconv2d.py:
an unfrozen (or frozen, input requiring grad) module must give the same input / bias gradients as the float module run on
the dequantized weight."""
import numpy as np
import pytest
import torch

import optimum_quanto_amd as Q
from oracle import quanto_oracle as O

from helpers import assert_close_to_exact, make_qbits_problem, make_qbytes_problem, to_numpy, to_torch

QTYPES = ["qint8", "qint4", "qfloat8_e4m3fn"]


def _grad_check(qmod, float_factory, x, device):
    """Gradients of sum(q(x) * g) w.r.t. x and bias equal those of the float module carrying the dequantized weight."""
    qmod = qmod.to(device)
    x = x.to(device)
    xq = x.clone().requires_grad_(True)
    yq = qmod(xq)
    g = torch.randn(yq.shape, generator=torch.Generator().manual_seed(7)).to(device=device, dtype=yq.dtype)
    (yq * g).sum().backward()
    ref = float_factory().to(device)
    with torch.no_grad():
        ref.weight.copy_(qmod.qweight.dequantize() if hasattr(qmod.qweight, "dequantize") else qmod.qweight)
        ref.bias.copy_(qmod.bias)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    (yr * g).sum().backward()
    assert xq.grad is not None and qmod.bias.grad is not None, "no gradient reached the input / bias"
    torch.testing.assert_close(yq, yr, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(xq.grad, xr.grad, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(qmod.bias.grad, ref.bias.grad, rtol=2e-2, atol=2e-2)
    return qmod


def _linear_case(qt, frozen, device):
    torch.manual_seed(11)
    lin = torch.nn.Linear(256, 64)
    q = Q.QLinear.from_module(lin, weights=getattr(Q, qt))
    if frozen:
        Q.freeze(q)
    x = torch.randn(5, 256)
    q = _grad_check(q, lambda: torch.nn.Linear(256, 64), x, device)
    if not frozen:
        assert q.weight.grad is not None and q.weight.grad.shape == q.weight.shape, "unfrozen weight must receive a gradient"


def _conv_case(qt, frozen, device):
    torch.manual_seed(12)
    conv = torch.nn.Conv2d(32, 16, 3, padding=1)
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, qt))
    if frozen:
        Q.freeze(q)
    x = torch.randn(2, 32, 9, 7)
    q = _grad_check(q, lambda: torch.nn.Conv2d(32, 16, 3, padding=1), x, device)
    if not frozen:
        assert q.weight.grad is not None and q.weight.grad.shape == q.weight.shape, "unfrozen weight must receive a gradient"


@pytest.mark.parametrize("frozen", [False, True])
@pytest.mark.parametrize("qt", QTYPES)
def test_qlinear_backward_cpu(qt, frozen):
    _linear_case(qt, frozen, "cpu")


@pytest.mark.parametrize("frozen", [False, True])
@pytest.mark.parametrize("qt", QTYPES)
def test_qconv2d_backward_cpu(qt, frozen):
    _conv_case(qt, frozen, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("frozen", [False, True])
@pytest.mark.parametrize("qt", QTYPES)
def test_qlinear_backward_gpu(qt, frozen):
    _linear_case(qt, frozen, "cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("frozen", [False, True])
@pytest.mark.parametrize("qt", QTYPES)
def test_qconv2d_backward_gpu(qt, frozen):
    """On the device a frozen conv in no_grad mode takes the fused GEMM; with gradients wanted it must fall back to the
    differentiable path (the fused ops have no autograd formula) - a missing gradient here is the regression."""
    _conv_case(qt, frozen, "cuda")


# ---------------------------------------------------------------------------------------------------------------------
# split-K workspace: small-N split (many partial sums, few counters) followed by large-N split on the SAME stream
# ---------------------------------------------------------------------------------------------------------------------
def _run_qbytes(lib, M, N, K, seed, dev):
    p = make_qbytes_problem(M, N, K, "bf16", None, seed=seed)
    y = lib.qbytes_mm(to_torch(p["x"], "bf16", dev), torch.from_numpy(p["data"]).to(dev), to_torch(p["scale"], "bf16", dev))
    return p, y


def _run_qbits(lib, M, N, K, seed, dev):
    p = make_qbits_problem(M, N, K, "bf16", seed=seed)
    y = lib.qbits_mm(to_torch(p["x"], "bf16", dev), torch.from_numpy(p["packed"]).to(dev), to_torch(p["scale"], "bf16", dev),
                     to_torch(p["shift"], "bf16", dev), None, 4, 128, N, K)
    return p, y


@pytest.mark.gpu
def test_split_k_workspace_survives_a_change_of_problem_size():
    """N=1024,K=8192 splits 8 ways and leaves non-zero partial sums right behind a 256-byte counter region;
    N=8192,K=8192 then needs 512 bytes of counters.  With per-problem counter sizes the second call read stale partials as
    counters and never wrote some feature blocks.  Same stream, int8 and int4, then a large-tile split, then back."""
    from optimum_quanto_amd.library.hip import quanto_hip

    lib, dev = quanto_hip.lib, "cuda"
    M = 16
    seq = []
    for rep in range(2):
        for kind, N, K in (("i8", 1024, 8192), ("i8", 8192, 8192), ("i4", 1024, 8192), ("i4", 8192, 8192), ("i8", 512, 14336),
                           ("i4", 4096, 4096), ("i8", 4096, 4096)):
            run = _run_qbytes if kind == "i8" else _run_qbits
            p, y = run(lib, M if N != 512 else 512, N, K, seed=100 + len(seq), dev=dev)
            seq.append((kind, N, K, p, y, lib.last_kernel()))
    torch.cuda.synchronize()
    for kind, N, K, p, y, kernel in seq:
        want = (O.qbytes_mm_exact(p["x"], p["data"], p["scale"]) if kind == "i8"
                else O.qbits_mm_exact(p["x"], p["packed"], 4, p["scale"], p["shift"], 128, N, K))
        got = to_numpy(y)
        assert np.isfinite(got).all(), f"{kind} N={N} K={K} ({kernel}): non-finite output (unwritten feature block?)"
        assert_close_to_exact(got, want, "bf16", f"{kind} N={N} K={K} kernel={kernel}")


@pytest.mark.gpu
def test_split_k_inside_a_graph_capture_then_eager():
    """First split-K call of a stream happening during hipGraph capture: the zero-fill must belong to that capture only, and
    eager calls afterwards (same stream) must not see the graph-pool buffer."""
    from optimum_quanto_amd.library.hip import quanto_hip

    lib, dev = quanto_hip.lib, "cuda"
    p = make_qbits_problem(16, 2048, 4096, "bf16", seed=5)
    x, packed = to_torch(p["x"], "bf16", dev), torch.from_numpy(p["packed"]).to(dev)
    scale, shift = to_torch(p["scale"], "bf16", dev), to_torch(p["shift"], "bf16", dev)
    want = O.qbits_mm_exact(p["x"], p["packed"], 4, p["scale"], p["shift"], 128, 2048, 4096)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            ys = [lib.qbits_mm(x, packed, scale, shift, None, 4, 128, 2048, 4096) for _ in range(3)]
        # eager on the very same stream, before any replay
        ye = lib.qbits_mm(x, packed, scale, shift, None, 4, 128, 2048, 4096)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert_close_to_exact(to_numpy(ye), want, "bf16", "eager after capture")
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    for y in ys:
        assert_close_to_exact(to_numpy(y), want, "bf16", "graph replay")


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 40])
def test_auto_dispatch_accepts_misaligned_views(M):
    """the fast kernels want 16-byte aligned x / weights; a view into a larger buffer may start anywhere.  AUTO must
    then run the kernel without an alignment requirement instead of surfacing QUANTO_HIP_EALIGN."""
    from optimum_quanto_amd.library.hip import quanto_hip

    lib, dev = quanto_hip.lib, "cuda"
    p = make_qbits_problem(M, 256, 512, "bf16", seed=M)
    xbuf = torch.zeros(M * 512 + 8, dtype=torch.bfloat16, device=dev)
    x = xbuf[1:1 + M * 512].view(M, 512)  # 2 bytes past a 16-byte boundary
    x.copy_(to_torch(p["x"], "bf16", dev))
    assert x.data_ptr() % 16 != 0
    y = lib.qbits_mm(x, torch.from_numpy(p["packed"]).to(dev), to_torch(p["scale"], "bf16", dev), to_torch(p["shift"], "bf16", dev), None,
                     4, 128, 256, 512)
    assert lib.last_kernel() == "naive"
    assert_close_to_exact(to_numpy(y), O.qbits_mm_exact(p["x"], p["packed"], 4, p["scale"], p["shift"], 128, 256, 512), "bf16", "misaligned qbits")
    q = make_qbytes_problem(M, 256, 512, "bf16", None, seed=M)
    x.copy_(to_torch(q["x"], "bf16", dev))
    y = lib.qbytes_mm(x, torch.from_numpy(q["data"]).to(dev), to_torch(q["scale"], "bf16", dev))
    assert lib.last_kernel() == "naive"
    assert_close_to_exact(to_numpy(y), O.qbytes_mm_exact(q["x"], q["data"], q["scale"]), "bf16", "misaligned qbytes")


@pytest.mark.gpu
def test_c_api_multi_fallback_restores_the_counter_words_behind_scratch_members():
    """quanto_hip_qbits_mm_multi_ws falls back to separate calls that share ONE workspace.  Member 0 (N = 4096, K = 4096,
    M = 1800) resolves to DEQUANT_MFMA and writes the dequantized weight from offset 0 - over the arrival counters; member 1 (N = 256)
    resolves to the split-K form of the fused int4 GEMM and needs those counters to be zero.  Straight through the C ABI with one
    caller-owned buffer (the Python binding keeps zeroed and scratch buffers apart, so only a C caller can hit this)."""
    import ctypes

    from optimum_quanto_amd.library.hip import quanto_hip

    lib, dev = quanto_hip.lib, "cuda"
    c = lib._c
    M, K, Ns = 1800, 4096, [4096, 256]
    BF16, DEQUANT, FUSED4 = 2, 7, 8
    c.quanto_hip_qbits_mm_pick.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int] * 3
    assert [c.quanto_hip_qbits_mm_pick(M, n, K, 4, 128, BF16) for n in Ns] == [DEQUANT, FUSED4]
    ps = [make_qbits_problem(M, n, K, "bf16", seed=7 + i) for i, n in enumerate(Ns)]
    x = to_torch(ps[0]["x"], "bf16", dev)
    packed = [torch.from_numpy(p["packed"]).to(dev) for p in ps]
    scale = [to_torch(p["scale"], "bf16", dev) for p in ps]
    shift = [to_torch(p["shift"], "bf16", dev) for p in ps]
    ys = [torch.full((M, n), float("nan"), dtype=torch.bfloat16, device=dev) for n in Ns]
    nfs = (ctypes.c_int64 * 2)(*Ns)
    c.quanto_hip_qbits_mm_multi_workspace_size.restype = ctypes.c_int64
    c.quanto_hip_qbits_mm_multi_workspace_size.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    need = c.quanto_hip_qbits_mm_multi_workspace_size(2, nfs, M, K, 4, 128, BF16)
    assert need >= 4096 * K * 2
    ws = torch.zeros((need,), dtype=torch.uint8, device=dev)  # the documented contract: counters zero on entry
    arr = lambda ts: (ctypes.c_void_p * 2)(*[t.data_ptr() for t in ts])  # noqa: E731
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for rep in range(2):  # the second round starts from whatever the first one left in the buffer
        st = c.quanto_hip_qbits_mm_multi_ws(ctypes.c_void_p(x.data_ptr()), 2, arr(packed), arr(scale), arr(shift), None, arr(ys), nfs,
                                            ctypes.c_int64(M), ctypes.c_int64(K), 4, 128, BF16, BF16, ctypes.c_void_p(ws.data_ptr()),
                                            ctypes.c_size_t(need), stream)
        assert st == 0, st
        torch.cuda.synchronize()
        assert int(ws[:4096].max()) == 0, "counter words left dirty"
        got = to_numpy(ys[1])
        assert np.isfinite(got).all(), "member 1: unwritten tiles (arrival election failed on dirty counters)"
        assert_close_to_exact(got, O.qbits_mm_exact(ps[0]["x"], ps[1]["packed"], 4, ps[1]["scale"], ps[1]["shift"], 128, 256, K), "bf16", f"member 1, round {rep}")
    rows = slice(0, 8)
    w0 = O.dequantize_qbits_ref(ps[0]["packed"], 4, ps[0]["scale"], ps[0]["shift"], 0, 128, (4096, K), "bf16")  # the reference's rounded weight
    want0 = np.matmul(ps[0]["x"][rows].astype(np.float64), w0.astype(np.float64).T)
    assert_close_to_exact(to_numpy(ys[0])[rows], want0, "bf16", "member 0 (dequantize + dense), first rows")
