"""CPU oracle for the optimum-quanto QLinear hot path (TEST INFRASTRUCTURE ONLY).

This module is a from-scratch numpy restatement of the algorithms behind
``optimum.quanto.nn.QLinear.forward`` for frozen weights: int4/int2 packing,
group-wise affine dequantization, per-channel int8/fp8 dequantization and the
``qbytes_mm`` / (implicit) ``qbits_mm`` products.  Every function cites the
reference file:line (paths relative to ``/root/reference/optimum/quanto``) it
follows.

Rules (see DESIGN.md "Oracle"):

* Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
  ``bench.py`` may import this file.  The product package never does, and it has
  no CPU fallback for CUDA tensors: it raises when the HIP library is missing.
* Parity is PINNED: ``tests/test_oracle_golden.py`` checks every function here
  against vectors produced by the real reference (``tests/golden/make_golden.py``
  imports ``/root/reference`` in this container and writes ``tests/golden/*.npz``).

Low precision float dtypes are emulated on float32 arrays whose values are
exactly representable in the emulated dtype ("bf16-valued float32").  The
reference runs torch CPU kernels that compute each elementwise op in float32
and round once to the tensor dtype; the ``*_ref`` functions below reproduce that
rounding sequence, the ``*_exact`` functions evaluate the same integers and
scale values in float64 with no intermediate rounding.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np

# --------------------------------------------------------------------------
# float formats
# --------------------------------------------------------------------------


def round_bf16(x: np.ndarray) -> np.ndarray:
    """Round float32 values to the nearest bfloat16 (ties to even); returns float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    bias = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    r = ((u + bias) & np.uint32(0xFFFF0000)).astype(np.uint32)
    out = r.view(np.float32).copy()
    nan = np.isnan(x)
    if nan.any():
        out[nan] = np.nan
    return out


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """uint16 bit patterns of bf16-valued float32 numbers."""
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> np.uint32(16)).astype(np.uint16)


def bf16_from_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def round_to(x: np.ndarray, dtype: str) -> np.ndarray:
    """Round a float array to ``dtype`` in {"fp32","fp16","bf16"}; result is float32/float16-valued float32."""
    if dtype == "fp32":
        return np.asarray(x, dtype=np.float32)
    if dtype == "fp16":
        return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)
    if dtype == "bf16":
        return round_bf16(np.asarray(x, dtype=np.float32))
    raise ValueError(dtype)


def _fp8_table(kind: str) -> np.ndarray:
    t = np.zeros(256, dtype=np.float64)
    for b in range(256):
        s = -1.0 if b & 0x80 else 1.0
        if kind == "e4m3fn":  # OCP E4M3: bias 7, no inf, S.1111.111 = NaN
            e, m = (b >> 3) & 0xF, b & 7
            if e == 15 and m == 7:
                v = math.nan
            elif e == 0:
                v = s * (m / 8.0) * 2.0**-6
            else:
                v = s * (1 + m / 8.0) * 2.0 ** (e - 7)
        elif kind == "e4m3fnuz":  # bias 8, no inf, no -0, 0x80 = NaN
            e, m = (b >> 3) & 0xF, b & 7
            if b == 0x80:
                v = math.nan
            elif e == 0:
                v = s * (m / 8.0) * 2.0**-7
            else:
                v = s * (1 + m / 8.0) * 2.0 ** (e - 8)
        elif kind == "e5m2":  # IEEE-like, bias 15
            e, m = (b >> 2) & 0x1F, b & 3
            if e == 31:
                v = s * math.inf if m == 0 else math.nan
            elif e == 0:
                v = s * (m / 4.0) * 2.0**-14
            else:
                v = s * (1 + m / 4.0) * 2.0 ** (e - 15)
        else:
            raise ValueError(kind)
        t[b] = v
    return t


FP8_TABLES = {k: _fp8_table(k) for k in ("e4m3fn", "e4m3fnuz", "e5m2")}
FP8_MAX = {"e4m3fn": 448.0, "e4m3fnuz": 240.0, "e5m2": 57344.0}


def fp8_decode(codes: np.ndarray, kind: str = "e4m3fn") -> np.ndarray:
    """uint8 codes -> float32 values (exact)."""
    return FP8_TABLES[kind][np.asarray(codes, dtype=np.uint8)].astype(np.float32)


def fp8_encode(x: np.ndarray, kind: str = "e4m3fn") -> np.ndarray:
    """Finite, in-range float values -> uint8 codes, round-to-nearest-even.

    Matches ``tensor.to(torch.float8_*)`` for |x| <= FP8_MAX (the reference clamps
    before converting: library/quantize.py:51-55).
    """
    x = np.asarray(x, dtype=np.float64)
    tab = FP8_TABLES[kind]
    pos_codes = np.array([c for c in range(128) if np.isfinite(tab[c])], dtype=np.uint8)
    pos_vals = tab[pos_codes]
    order = np.argsort(pos_vals, kind="stable")
    pos_codes, pos_vals = pos_codes[order], pos_vals[order]
    a = np.abs(x)
    hi = np.clip(np.searchsorted(pos_vals, a, side="left"), 0, len(pos_vals) - 1)
    lo = np.clip(hi - 1, 0, len(pos_vals) - 1)
    dlo, dhi = np.abs(a - pos_vals[lo]), np.abs(pos_vals[hi] - a)
    pick_hi = dhi < dlo
    tie = dhi == dlo
    # ties: even mantissa code (codes are monotone in value, so parity of code == parity of mantissa lsb)
    pick_hi = np.where(tie, (pos_codes[hi] & 1) == 0, pick_hi)
    code = np.where(pick_hi, pos_codes[hi], pos_codes[lo]).astype(np.uint8)
    neg = np.signbit(x)
    if kind == "e4m3fnuz":
        neg = neg & (code != 0)  # no negative zero
    return (code | (neg.astype(np.uint8) << 7)).astype(np.uint8)


# --------------------------------------------------------------------------
# packing (bit-exact integer work)
# --------------------------------------------------------------------------


def pack_weights(unpacked: np.ndarray, bits: int) -> np.ndarray:
    """int4/int2 values (uint8, one per byte) -> packed uint8.

    Follows tensor/packed.py:24-69: the first dimension is split in 8/bits planes
    of ``row_dim = ceil(rows / (8/bits))`` rows; plane ``i`` is stored in bits
    ``[bits*i, bits*(i+1))`` of the packed byte at the same (row, col).
    """
    assert bits in (2, 4)
    u = np.asarray(unpacked).astype(np.uint8)
    vpi = 8 // bits
    rows = u.shape[0]
    row_dim = (rows + vpi - 1) // vpi
    packed = np.zeros((row_dim,) + u.shape[1:], dtype=np.uint8)
    for i in range(vpi):
        start = i * row_dim
        end = min(start + row_dim, rows)
        if end > start:
            packed[: end - start] |= (u[start:end] << np.uint8(bits * i)).astype(np.uint8)
    return packed


def unpack(packed: np.ndarray, bits: int) -> np.ndarray:
    """Inverse of :func:`pack_weights` without the final row trim.

    Follows library/unpack.py:21-54 (quanto::unpack): the output first dimension
    is ``packed.shape[0] * 8 / bits``; callers slice ``[:rows]``
    (tensor/packed.py:101-104).
    """
    assert bits in (2, 4)
    p = np.asarray(packed, dtype=np.uint8)
    vpi = 8 // bits
    mask = np.uint8((1 << bits) - 1)
    planes = [((p >> np.uint8(bits * i)) & mask) for i in range(vpi)]
    return np.concatenate(planes, axis=0).astype(np.uint8)


def group(base: np.ndarray, axis: int, group_size: int) -> np.ndarray:
    """tensor/grouped.py:17-39."""
    if axis not in (0, -1):
        raise ValueError("Axis must be 0 or -1 for group-wise quantization")
    axis_dim = base.shape[axis]
    axis_numel = base.size // axis_dim
    if group_size > axis_numel or axis_numel % group_size != 0:
        raise ValueError(f"Group size ({group_size}) must be a divisor of ({axis_numel})")
    axis_groups = axis_numel // group_size
    if axis == 0:
        return base.reshape(-1, group_size)
    g = base.reshape(axis_groups, group_size, axis_dim).transpose(1, 2, 0)
    return g.reshape(group_size, axis_dim * axis_groups)


def ungroup(grouped: np.ndarray, axis: int, orig_shape: Tuple[int, ...]) -> np.ndarray:
    """tensor/grouped.py:39-51."""
    if tuple(grouped.shape) == tuple(orig_shape):
        return grouped
    if axis == 0:
        return grouped.reshape(orig_shape)
    group_size = grouped.shape[0]
    axis_dim = orig_shape[axis]
    axis_groups = grouped.size // axis_dim // group_size
    u = grouped.reshape(group_size, axis_dim, axis_groups).transpose(2, 0, 1)
    return u.reshape(orig_shape)


# --------------------------------------------------------------------------
# quantize-time helpers (needed to build inputs the same way the reference does)
# --------------------------------------------------------------------------


def _reduce_dims(ndim: int, axis: int):
    dims = tuple(range(1, ndim)) if axis == 0 else tuple(range(0, ndim - 1))
    # torch.amin/amax with an empty dim list reduce over every dimension (1-D inputs)
    return dims if dims else None


def absmax_scale(base: np.ndarray, qmax: float, axis: Optional[int], dtype: str = "fp32") -> np.ndarray:
    """tensor/optimizers/absmax_optimizer.py:26-36 (+ symmetric_optimizer.py axis collapse)."""
    b = np.abs(np.asarray(base, dtype=np.float32))
    if axis is not None and base.shape[axis] == 1:
        axis = None
    rmax = b.max() if axis is None else b.max(axis=_reduce_dims(b.ndim, axis), keepdims=True)
    return round_to(np.float32(rmax) / np.float32(qmax), dtype)


def max_scale_shift(base: np.ndarray, bits: int, axis: int, group_size: Optional[int], dtype: str = "fp32"):
    """tensor/optimizers/max_optimizer.py:26-37 with the grouping of affine_optimizer.py:50-51."""
    b = np.asarray(base, dtype=np.float32)
    if group_size is not None:
        b = group(b, axis, group_size)
    dims = _reduce_dims(b.ndim, axis)
    rmin = b.min(axis=dims, keepdims=True)
    rmax = b.max(axis=dims, keepdims=True)
    qmin, qmax = -(2 ** (bits - 1)), 2 ** (bits - 1) - 1
    scale = round_to(round_to(rmax - rmin, dtype) / np.float32(qmax - qmin), dtype)
    shift = round_to(-rmin, dtype)
    return scale, shift


def quantize_symmetric_int8(base: np.ndarray, scale: np.ndarray, dtype: str = "fp32") -> np.ndarray:
    """library/quantize.py:26-55 for an int8 target: clamp(round(base/scale), -128, 127)."""
    d = round_to(np.asarray(base, np.float32) / np.asarray(scale, np.float32), dtype)
    return np.clip(np.rint(d), -128, 127).astype(np.int8)


def quantize_symmetric_fp8(base: np.ndarray, scale: np.ndarray, kind: str = "e4m3fn", dtype: str = "fp32") -> np.ndarray:
    """library/quantize.py:26-55 for a float8 target: clamp(base/scale, -max, max).to(fp8). Returns uint8 codes."""
    d = round_to(np.asarray(base, np.float32) / np.asarray(scale, np.float32), dtype)
    m = FP8_MAX[kind]
    return fp8_encode(np.clip(d, -m, m), kind)


def quantize_affine(base, bits, axis, group_size, scale, shift, dtype: str = "fp32") -> np.ndarray:
    """library/quantize.py:66-78.  ``shift`` float -> round((base+shift)/scale); integer -> round(base/scale)+shift."""
    b = np.asarray(base, np.float32)
    if group_size is not None:
        b = group(b, axis, group_size)
    scale = np.asarray(scale, np.float32)
    if np.issubdtype(np.asarray(shift).dtype, np.floating):
        data = np.rint(round_to(round_to(b + np.asarray(shift, np.float32), dtype) / scale, dtype))
    else:
        data = np.rint(round_to(b / scale, dtype)) + np.asarray(shift).astype(np.float32)
    return np.clip(data, 0, 2**bits - 1).astype(np.uint8)


# --------------------------------------------------------------------------
# dequantization
# --------------------------------------------------------------------------


def dequantize_qbytes_ref(data, scale, dtype: str, fp8_kind: Optional[str] = None) -> np.ndarray:
    """tensor/qbytes.py:23-36: ``scale * data`` in the scale dtype (one rounding)."""
    w = fp8_decode(data, fp8_kind) if fp8_kind else np.asarray(data).astype(np.float32)
    return round_to(np.asarray(scale, np.float32) * w, dtype)


def unpacked_rows(packed: np.ndarray, bits: int, rows: int) -> np.ndarray:
    """tensor/packed.py:101-104: quanto::unpack then trim to the original first dim."""
    return unpack(packed, bits)[:rows]


def dequantize_qbits_ref(packed, bits, scale, shift, axis, group_size, shape, dtype: str) -> np.ndarray:
    """tensor/qbits.py:27-49 with the reference's rounding sequence.

    integer zero-point:  W = round(scale * (q - zp))
    float shift:         W = round(round(scale * q) - shift)      (two roundings)
    then ungroup (tensor/grouped.py:39-51).
    """
    n = int(np.prod(shape))
    rows = n // group_size if group_size is not None else shape[0]
    q = unpacked_rows(packed, bits, rows).astype(np.float32)
    scale = np.asarray(scale, np.float32)
    sh = np.asarray(shift)
    if np.issubdtype(sh.dtype, np.floating):
        dq = round_to(scale * q, dtype)
        dq = round_to(dq - sh.astype(np.float32), dtype)
    else:
        dq = round_to(scale * (q - sh.astype(np.float32)), dtype)
    if group_size is None:
        return dq.reshape(shape)
    return ungroup(dq, axis, tuple(shape))


def dequantize_qbits_exact(packed, bits, scale, shift, axis, group_size, shape) -> np.ndarray:
    """Same integers / scale / shift values, float64, no intermediate rounding."""
    n = int(np.prod(shape))
    rows = n // group_size if group_size is not None else shape[0]
    q = unpacked_rows(packed, bits, rows).astype(np.float64)
    scale = np.asarray(scale, np.float64)
    sh = np.asarray(shift)
    if np.issubdtype(sh.dtype, np.floating):
        dq = scale * q - sh.astype(np.float64)
    else:
        dq = scale * (q - sh.astype(np.float64))
    if group_size is None:
        return dq.reshape(shape)
    return ungroup(dq, axis, tuple(shape))


# --------------------------------------------------------------------------
# the products behind QLinear.forward
# --------------------------------------------------------------------------


def qbytes_mm_ref(a, b, scales, dtype: str, fp8_kind: Optional[str] = None) -> np.ndarray:
    """library/qbytes_mm.py:25-33: A.to(sdt) @ round(scales * B.to(sdt)).T, fp32 accumulate, output rounded."""
    w = dequantize_qbytes_ref(b, scales, dtype, fp8_kind)
    a = round_to(np.asarray(a, np.float32), dtype)
    return round_to(np.matmul(a, w.T), dtype)


def qbytes_int_mm_ref(a_i8, b_i8, scales, dtype: str) -> np.ndarray:
    """library/qbytes_mm.py:36-50: int32 = A_i8 @ B_i8.T; (fp32(int32) * scales.T).to(dtype)."""
    # float64 BLAS matmul of small integers is exact (every partial sum is an integer below 2^53) and, unlike numpy's integer
    # matmul, multi-threaded: full BASELINE-size checks stay within seconds
    acc = np.matmul(a_i8.astype(np.float64), b_i8.astype(np.float64).T)
    assert np.abs(acc).max() < 2**31
    acc = acc.astype(np.int64)
    out = acc.astype(np.float32) * np.asarray(scales, np.float32).reshape(1, -1)
    return round_to(out, dtype)


def qbytes_mm_exact(a, b, scales, fp8_kind: Optional[str] = None) -> np.ndarray:
    w = fp8_decode(b, fp8_kind).astype(np.float64) if fp8_kind else np.asarray(b).astype(np.float64)
    acc = np.matmul(np.asarray(a, np.float64), w.T)
    return acc * np.asarray(scales, np.float64).reshape(1, -1)


def qbits_mm_ref(x, packed, bits, scale, shift, group_size, out_features, in_features, dtype: str, bias=None):
    """tensor/function.py:41-47 on the dequantized weight (tensor/qbits.py:27-49): x @ W.T (+ bias)."""
    w = dequantize_qbits_ref(packed, bits, scale, shift, 0, group_size, (out_features, in_features), dtype)
    y = round_to(np.matmul(round_to(np.asarray(x, np.float32), dtype), w.T), dtype)
    if bias is not None:
        y = round_to(y + np.asarray(bias, np.float32), dtype)
    return y


def qbits_mm_exact(x, packed, bits, scale, shift, group_size, out_features, in_features, bias=None):
    w = dequantize_qbits_exact(packed, bits, scale, shift, 0, group_size, (out_features, in_features))
    y = np.matmul(np.asarray(x, np.float64), w.T)
    if bias is not None:
        y = y + np.asarray(bias, np.float64)
    return y


def qbits_mm_a8_exact(a_values, a_scale, packed, bits, scale, shift, group_size, out_features, in_features, bias=None):
    """F.linear(quantized activation, int4 weight) in exact (float64) math on the stored values: ``a_values`` are the activation's stored
    integers / decoded fp8 values, ``a_scale`` its per-tensor scale - the product the reference approximates after dequantizing both
    operands (tensor/weights/qbits.py:262-287 -> tensor/function.py:41-47; tests/tensor/ops/test_linear_dispatch.py:22-42)."""
    w = dequantize_qbits_exact(packed, bits, scale, shift, 0, group_size, (out_features, in_features))
    y = np.matmul(np.asarray(a_values, np.float64), w.T) * float(np.asarray(a_scale, np.float64).reshape(-1)[0])
    if bias is not None:
        y = y + np.asarray(bias, np.float64)
    return y


def _fma32(a, b, c):
    """fp32 fused multiply-add on arrays of fp32 values: the product of an 8-bit-mantissa scale and an integer below 2^24 is exact in float64,
    and so is its sum with an fp32 addend of comparable magnitude (the spans met here stay far below 53 bits): one rounding, to fp32."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def qbits_mm_a8_chain(a_i8, a_scale, packed, bits, scale, shift, group_size, out_features, in_features, dtype: str, bias=None):
    """The W4A8 kernel's arithmetic for int8 activations, restated operation by operation (csrc/qbits_a8_fused.hip, unsplit form):
    per group g in order, acc = fma(s[n,g], fp32(P_g), acc); acc = fma(-z[n,g], fp32(A_g), acc) with the exact integers
    P_g = sum_{k in g} a q and A_g = sum_{k in g} a, z = shift (float shift) or fp32(s * zero_point); then fp32(acc * a_scale) rounded to
    ``dtype`` (+ bias: rounded product plus bias, rounded again - tensor/function.py:45-46).  Bit-exact gate of the int8 path."""
    N, K = out_features, in_features
    G = K // group_size
    qT = np.ascontiguousarray(unpacked_rows(packed, bits, N * G).reshape(N, G, group_size).astype(np.float64).transpose(1, 2, 0))  # [g][k][n]
    a = np.asarray(a_i8).astype(np.float64).reshape(-1, G, group_size)
    s = np.asarray(scale, np.float32).reshape(N, G)
    sh = np.asarray(shift).reshape(N, G)
    z = sh.astype(np.float32) if np.issubdtype(sh.dtype, np.floating) else (s * sh.astype(np.float32)).astype(np.float32)
    s64, nz64 = s.astype(np.float64), -z.astype(np.float64)
    acc = np.zeros((a.shape[0], N), np.float32)

    def rows(r0, r1):
        # the same two fused operations per group as _fma32 states them (product and sum exact in float64, ONE rounding to fp32 each), on a
        # block of rows with the fp32-valued accumulator carried in float64 between roundings: no broadcast temporaries, cache-sized blocks
        acc64 = np.zeros((r1 - r0, N), np.float64)
        t = np.empty_like(acc64)
        for g in range(G):
            ag = a[r0:r1, g, :]
            np.matmul(ag, qT[g], out=t)                 # P_g: exact integers below 2^24
            t *= s64[:, g][None, :]
            t += acc64
            acc64[...] = t.astype(np.float32)           # acc = fma(s, P, acc)
            A = ag.sum(axis=1)                          # exact integer
            np.multiply(A[:, None], nz64[:, g][None, :], out=t)
            t += acc64
            acc64[...] = t.astype(np.float32)           # acc = fma(-z, A, acc)
        acc[r0:r1] = acc64

    step = 256
    blocks = [(r0, min(r0 + step, a.shape[0])) for r0 in range(0, a.shape[0], step)]
    if len(blocks) > 1:
        import os
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=min(len(blocks), max(1, (os.cpu_count() or 1)))) as pool:
            list(pool.map(lambda b: rows(*b), blocks))
    else:
        rows(*blocks[0])
    sx = np.float32(np.asarray(a_scale, np.float32).reshape(-1)[0])
    y = round_to((acc * sx).astype(np.float32), dtype)
    if bias is not None:
        y = round_to(y + np.asarray(bias, np.float32), dtype)
    return y


# --------------------------------------------------------------------------
# error metrics shared by the tests
# --------------------------------------------------------------------------


def rel_fro(y, y_ref) -> float:
    y, y_ref = np.asarray(y, np.float64), np.asarray(y_ref, np.float64)
    d = np.linalg.norm(y_ref)
    return float(np.linalg.norm(y - y_ref) / (d if d > 0 else 1.0))


def rel_max(y, y_ref) -> float:
    y, y_ref = np.asarray(y, np.float64), np.asarray(y_ref, np.float64)
    d = np.abs(y_ref).max()
    return float(np.abs(y - y_ref).max() / (d if d > 0 else 1.0))


def ulp_distance(y, y_ref, dtype: str) -> np.ndarray:
    """Distance in units of the last place of ``dtype`` between two dtype-valued arrays."""
    if dtype == "bf16":
        a = bf16_bits(y).astype(np.int32)
        b = bf16_bits(y_ref).astype(np.int32)
    elif dtype == "fp16":
        a = np.asarray(y, np.float32).astype(np.float16).view(np.uint16).astype(np.int32)
        b = np.asarray(y_ref, np.float32).astype(np.float16).view(np.uint16).astype(np.int32)
    else:
        a = np.asarray(y, np.float32).view(np.uint32).astype(np.int64)
        b = np.asarray(y_ref, np.float32).view(np.uint32).astype(np.int64)
        sign = np.int64(1) << 31
        a = np.where(a & sign, sign - a, a)
        b = np.where(b & sign, sign - b, b)
        return np.abs(a - b)
    sign = 1 << 15
    a = np.where(a & sign, sign - a, a)
    b = np.where(b & sign, sign - b, b)
    return np.abs(a - b)
