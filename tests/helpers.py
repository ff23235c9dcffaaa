"""Shared helpers for the parity tests: oracle-built inputs and numpy <-> torch conversions."""
import numpy as np
import torch

from oracle import quanto_oracle as O

TORCH_DT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}
FP8_TORCH = {"e4m3fn": torch.float8_e4m3fn, "e4m3fnuz": torch.float8_e4m3fnuz, "e5m2": torch.float8_e5m2}


def to_torch(a: np.ndarray, dt: str, device="cpu") -> torch.Tensor:
    """float32 array holding dt-representable values -> torch tensor of that dtype (exact)."""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(TORCH_DT[dt]).to(device)


def to_numpy(t: torch.Tensor) -> np.ndarray:
    if t.dtype in FP8_TORCH.values():
        return t.view(torch.uint8).cpu().numpy()
    if t.dtype.is_floating_point:
        return t.detach().to(torch.float32).cpu().numpy()
    return t.detach().cpu().numpy()


def fp8_tensor(codes: np.ndarray, kind: str, device="cpu") -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(codes, dtype=np.uint8)).view(FP8_TORCH[kind]).to(device)


_WEIGHT_CACHE = {}  # weight_seed given: the quantized weight of a (shape, format, seed) is built once per session and shared by every M


def _activations(M, K, dt, rng):
    return O.round_to(rng.standard_normal((M, K)).astype(np.float32), dt)


def make_qbits_problem(M, N, K, dt, bits=4, group_size=128, zeropoint=False, seed=0, wscale=0.02, weight_seed=None):
    """Seeded activations + a weight quantized by the (reference-pinned) oracle, generic PackedTensor layout.
    ``weight_seed``: draw the weight from its own stream and cache the quantized result (suites that sweep M over one weight)."""

    def weight(rng):
        w = O.round_to((rng.standard_normal((N, K)) * wscale).astype(np.float32), dt)
        scale, shift = O.max_scale_shift(w, bits, 0, group_size, dt)
        if zeropoint:
            shift = np.clip(np.rint(O.round_to(shift / scale, dt)), 0, 2**bits - 1).astype(np.uint8)
        q = O.quantize_affine(w, bits, 0, group_size, scale, shift, dt)
        return dict(packed=O.pack_weights(q, bits), scale=scale, shift=shift)

    rng = np.random.default_rng(seed)
    if weight_seed is None:
        wq = weight(rng)  # historical stream order: weight first, then activations
    else:
        key = ("qbits", N, K, dt, bits, group_size, zeropoint, weight_seed, wscale)
        if key not in _WEIGHT_CACHE:
            _WEIGHT_CACHE[key] = weight(np.random.default_rng(weight_seed))
        wq = _WEIGHT_CACHE[key]
    x = _activations(M, K, dt, rng)
    return dict(x=x, packed=wq["packed"], scale=wq["scale"], shift=wq["shift"], bits=bits, group_size=group_size, N=N, K=K, dt=dt,
                wkey=None if weight_seed is None else key)


_EXACT_W = {}  # float64 dequantized weights of cached problems, the last six: an M sweep alternates dtype x zero-point (x members of a multi launch)


def qbits_exact(p, x=None, bias=None):
    """O.qbits_mm_exact on a problem of make_qbits_problem (``x``: another activation, e.g. the shared input of a multi launch); the float64
    weight of a cached problem (``weight_seed``) is kept between the Ms of a sweep instead of being dequantized again for every M."""
    x = p["x"] if x is None else x
    key = p.get("wkey")
    if key is None:
        return O.qbits_mm_exact(x, p["packed"], p["bits"], p["scale"], p["shift"], p["group_size"], p["N"], p["K"], bias)
    if key not in _EXACT_W:
        while len(_EXACT_W) >= 6:
            _EXACT_W.pop(next(iter(_EXACT_W)))
        _EXACT_W[key] = O.dequantize_qbits_exact(p["packed"], p["bits"], p["scale"], p["shift"], 0, p["group_size"], (p["N"], p["K"]))
    y = np.matmul(np.asarray(x, np.float64), _EXACT_W[key].T)
    return y if bias is None else y + np.asarray(bias, np.float64)


def make_qbytes_problem(M, N, K, dt, kind=None, seed=0, wscale=0.02, weight_seed=None):
    """int8 (kind None) or fp8 weight with per-row absmax scale, built by the oracle (``weight_seed``: as make_qbits_problem)."""

    def weight(rng):
        w = O.round_to((rng.standard_normal((N, K)) * wscale).astype(np.float32), dt)
        if kind is None:
            scale = O.absmax_scale(w, 127.0, 0, dt)
            data = O.quantize_symmetric_int8(w, scale, dt)
        else:
            scale = O.absmax_scale(w, O.FP8_MAX[kind], 0, dt)
            data = O.quantize_symmetric_fp8(w, scale, kind, dt)
        return dict(data=data, scale=scale)

    rng = np.random.default_rng(seed)
    if weight_seed is None:
        wq = weight(rng)
    else:
        key = ("qbytes", N, K, dt, kind, weight_seed, wscale)
        if key not in _WEIGHT_CACHE:
            _WEIGHT_CACHE[key] = weight(np.random.default_rng(weight_seed))
        wq = _WEIGHT_CACHE[key]
    x = _activations(M, K, dt, rng)
    return dict(x=x, data=wq["data"], scale=wq["scale"], kind=kind, N=N, K=K, dt=dt, wkey=None if weight_seed is None else key)


def qbytes_exact(p, x=None):
    """O.qbytes_mm_exact on a problem of make_qbytes_problem; the float64 image of a cached weight is kept between the Ms of a sweep."""
    x = p["x"] if x is None else x
    key = p.get("wkey")
    if key is None:
        return O.qbytes_mm_exact(x, p["data"], p["scale"], p["kind"])
    if key not in _EXACT_W:
        while len(_EXACT_W) >= 6:
            _EXACT_W.pop(next(iter(_EXACT_W)))
        w = O.fp8_decode(p["data"], p["kind"]).astype(np.float64) if p["kind"] else np.asarray(p["data"]).astype(np.float64)
        _EXACT_W[key] = np.ascontiguousarray(w.T)
    return np.matmul(np.asarray(x, np.float64), _EXACT_W[key]) * np.asarray(p["scale"], np.float64).reshape(1, -1)


def assert_close_to_exact(y: np.ndarray, y_exact: np.ndarray, dt: str, what=""):
    """The parity gate (DESIGN.md "Parity"):

    * fp32 / fp16 outputs: relative Frobenius AND relative max error vs exact math <= 1e-3 (north-star tolerance);
      expected ~1e-6 (fp32) and ~3e-4 (fp16, pure output rounding).
    * bf16 outputs: one bf16 ulp is 3.9e-3, so the gate is "within 1 ulp of the correctly rounded exact result on
      >= 99.5 % of the elements, never more than 2 ulp, and <= 1e-3 Frobenius against that rounded result".
    """
    y = np.asarray(y, np.float64)
    if dt in ("fp32", "fp16"):
        fro, mx = O.rel_fro(y, y_exact), O.rel_max(y, y_exact)
        assert fro <= 1e-3 and mx <= 1e-3, f"{what}: rel_fro={fro:.3e} rel_max={mx:.3e}"
    else:
        target = O.round_to(np.asarray(y_exact, np.float32), dt)
        ulps = O.ulp_distance(y, target, dt)
        frac = float((ulps <= 1).mean())
        fro = O.rel_fro(y, target)
        # tiny outputs (cancellation) can sit many bf16 ulps away while being accurate in absolute terms
        scale_abs = np.abs(y_exact).max()
        big = np.abs(y_exact) > 1e-2 * scale_abs
        assert frac >= 0.995 and ulps[big].max(initial=0) <= 2 and fro <= 1e-3, \
            f"{what}: frac<=1ulp={frac:.4f} max_ulp={ulps[big].max(initial=0)} rel_fro={fro:.3e}"


def assert_close_with_bias(y: np.ndarray, prod_exact: np.ndarray, bias: np.ndarray, dt: str, what=""):
    """Reference order of operations: round the product to ``dt``, add the bias, round again
    (tensor/function.py:45-46, tensor/weights/qbytes.py:79-81).  A 1-ulp difference of the rounded product is
    legitimate (accumulation order), so the bound is one ulp of the product plus one ulp of the result."""
    y = np.asarray(y, np.float64)
    prod = O.round_to(np.asarray(prod_exact, np.float32), dt).astype(np.float64)
    want = O.round_to((prod + bias).astype(np.float32), dt).astype(np.float64)
    eps = {"fp32": 2.0**-23, "fp16": 2.0**-10, "bf16": 2.0**-7}[dt]
    bound = eps * (np.abs(prod) + np.abs(want)) * 1.01 + 1e-30
    bad = np.abs(y - want) > bound
    assert not bad.any(), f"{what}: {int(bad.sum())} elements beyond 1 ulp(product)+1 ulp(result); max err {np.abs(y - want).max():.3e}"
    assert (y == want).mean() > 0.97, f"{what}: only {(y == want).mean():.4f} identical to the reference sequence"


def assert_similar(a: torch.Tensor, b: torch.Tensor, atol=None, rtol=None):
    """The reference's own similarity check (tests/helpers.py:85-99): cosine similarity ~ 1."""
    assert a.dtype == b.dtype and a.shape == b.shape
    if atol is None:
        atol = torch.finfo(a.dtype).resolution
    if rtol is None:
        rtol = {torch.float32: 1e-5, torch.float16: 1e-3, torch.bfloat16: 1e-1}[a.dtype]
    sim = torch.nn.functional.cosine_similarity(a.flatten().float(), b.flatten().float(), dim=0)
    assert torch.allclose(sim, torch.tensor(1.0, dtype=sim.dtype, device=sim.device), atol=atol, rtol=rtol), \
        f"alignment {float(sim):.8f} deviates from 1"


class observed_activation_scales:
    """Test stand-in for a calibration pass (the reference's ``Calibration`` is host code outside this backend's scope; plug-in mode
    uses the reference's own).  While active, every quantized module with quantized activations takes ``input_scale`` /
    ``output_scale`` = the absmax scale of the tensor it just saw (last batch wins, no running average)."""

    def __enter__(self):
        from torch.nn.modules.module import register_module_forward_hook, register_module_forward_pre_hook

        import optimum_quanto_amd as Q

        def wanted(m):
            return isinstance(m, Q.QModuleMixin) and m.activation_qtype is not None

        def before(m, args):  # global hooks run before the module's own quantize_input / quantize_output hooks
            if wanted(m):
                x = args[0]
                m.input_scale = (torch.max(x._scale) if isinstance(x, Q.ActivationQBytesTensor) else Q.absmax_scale(x, m.activation_qtype)).to(m.input_scale.dtype)

        def after(m, args, out):
            if wanted(m):
                m.output_scale = Q.absmax_scale(out, m.activation_qtype).to(m.output_scale.dtype)

        self.handles = [register_module_forward_pre_hook(before), register_module_forward_hook(after)]
        return self

    def __exit__(self, *exc):
        for h in self.handles:
            h.remove()
