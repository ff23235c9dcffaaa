"""Pin the numpy oracle (oracle/quanto_oracle.py) against vectors from the REAL reference.

Bit-exact for every integer/byte result and for the dtype-faithful dequantizers;
tolerance (stated per test) for matmul outputs, whose summation order inside
ATen is not part of the reference's contract.
"""
import numpy as np
import pytest

from oracle import quanto_oracle as O


def _keys(golden, prefix):
    return sorted({k[: k.rfind("/")] for k in golden if k.startswith(prefix)})


def test_pack_unpack_bit_exact(golden):
    cases = _keys(golden, "pack/")
    assert len(cases) == 16
    for c in cases:
        bits = int(c.split("/")[1][1:])
        a, packed, unpacked = golden[c + "/a"], golden[c + "/packed"], golden[c + "/unpacked"]
        np.testing.assert_array_equal(O.pack_weights(a, bits), packed)
        np.testing.assert_array_equal(O.unpack(packed, bits), unpacked)
        np.testing.assert_array_equal(O.unpack(packed, bits)[: a.shape[0]], a)


@pytest.mark.parametrize("dt", ["fp32", "fp16"])
@pytest.mark.parametrize("bits", [2, 4])
def test_integer_ramp_known_answer(golden, dt, bits):
    k = f"ramp/{dt}/b{bits}"
    a = golden[k + "/a"]
    scale, shift = O.max_scale_shift(a, bits, 0, None, dt)
    np.testing.assert_array_equal(scale, golden[k + "/scale"])
    np.testing.assert_array_equal(shift, golden[k + "/shift"])
    zp = np.rint(O.round_to(shift / scale, dt))
    np.testing.assert_array_equal(zp, golden[k + "/zeropoint"])
    data = O.quantize_affine(a, bits, 0, None, scale, zp.astype(np.int32), dt)
    np.testing.assert_array_equal(data, golden[k + "/data"])
    # the reference's own assertion: the integer ramp is preserved
    np.testing.assert_array_equal(data.astype(np.float32) - zp, a)


def _dt_of(tag):
    for dt in ("fp32", "fp16", "bf16"):
        if dt in tag:
            return dt
    raise AssertionError(tag)


QBITS_CASES = [
    "int4_g128_fp32", "int4_g128_fp16", "int4_g128_bf16", "int4_g128_bf16_bias", "int4_g128_fp16_zp",
    "int4_g64_fp32", "int4_perchannel_fp32", "int4_oddrows_fp32", "int2_g128_fp32", "int2_g128_bf16",
    "int4_g128_bf16_small_w",
]


@pytest.mark.parametrize("tag", QBITS_CASES)
def test_qbits_quantize_pack_dequantize_bit_exact(golden, tag):
    k = f"qbits/{tag}"
    dt = _dt_of(tag)
    N, K, bits, gs, zp = [int(v) for v in golden[k + "/meta"]]
    gs = gs or None
    w = golden[k + "/w"]
    shift_g = golden[k + "/shift"]
    if not zp:
        scale, shift = O.max_scale_shift(w, bits, 0, gs, dt)
        np.testing.assert_array_equal(scale, golden[k + "/scale"])
        np.testing.assert_array_equal(shift, shift_g)
    scale = golden[k + "/scale"]
    q = O.quantize_affine(w, bits, 0, gs, scale, shift_g, dt)
    np.testing.assert_array_equal(q, golden[k + "/unpacked"])
    np.testing.assert_array_equal(O.pack_weights(q, bits), golden[k + "/packed"])
    dq = O.dequantize_qbits_ref(golden[k + "/packed"], bits, scale, shift_g, 0, gs, (N, K), dt)
    np.testing.assert_array_equal(dq, golden[k + "/dequantized"])
    # exact dequantization stays within the two roundings of the reference
    dqe = O.dequantize_qbits_exact(golden[k + "/packed"], bits, scale, shift_g, 0, gs, (N, K))
    tol = {"fp32": 1e-6, "fp16": 2e-3, "bf16": 1.6e-2}[dt]
    assert np.abs(dqe - dq).max() <= tol * max(1.0, np.abs(dq).max())


@pytest.mark.parametrize("tag", QBITS_CASES)
def test_qbits_linear_matches_reference(golden, tag):
    k = f"qbits/{tag}"
    dt = _dt_of(tag)
    N, K, bits, gs, _ = [int(v) for v in golden[k + "/meta"]]
    gs = gs or None
    bias = golden.get(k + "/bias")
    Ms = sorted(int(key.rsplit("/x", 1)[1]) for key in golden if key.startswith(k + "/x"))
    assert Ms
    for M in Ms:
        x, y = golden[k + f"/x{M}"], golden[k + f"/y{M}"]
        y_ref = O.qbits_mm_ref(x, golden[k + "/packed"], bits, golden[k + "/scale"], golden[k + "/shift"],
                               gs, N, K, dt, bias)
        # same rounded weights, fp32 accumulate: only the summation order differs ->
        # fp32: 1e-5 relative; fp16/bf16: at most 1 ulp of the output dtype on a few elements
        if dt == "fp32":
            assert O.rel_max(y_ref, y) < 1e-5
        else:
            ulps = O.ulp_distance(y_ref, y, dt)
            assert ulps.max() <= 1 or O.rel_max(y_ref, y) < {"fp16": 1e-3, "bf16": 8e-3}[dt]
            assert (ulps <= 1).mean() > 0.99
        # the exact evaluation is within the reference's own noise floor
        y_ex = O.qbits_mm_exact(x, golden[k + "/packed"], bits, golden[k + "/scale"], golden[k + "/shift"],
                                gs, N, K, bias)
        assert O.rel_fro(y_ex, y) < {"fp32": 1e-5, "fp16": 2e-3, "bf16": 1.5e-2}[dt]


QBYTES_CASES = ["int8_fp32", "int8_fp16", "int8_bf16", "e4m3fn_fp32", "e4m3fn_fp16", "e4m3fn_bf16",
                "int8_bf16_bias", "e4m3fnuz_fp16", "e5m2_fp16", "cfg1_int8_fp32_1x1024x1024"]


def _fp8_kind(tag):
    for kind in ("e4m3fnuz", "e4m3fn", "e5m2"):
        if tag.startswith(kind):
            return kind
    return None


@pytest.mark.parametrize("tag", QBYTES_CASES)
def test_qbytes_quantize_dequantize_bit_exact(golden, tag):
    k = f"qbytes/{tag}"
    dt = _dt_of(tag)
    kind = _fp8_kind(tag)
    scale, data = golden[k + "/scale"], golden[k + "/data"]
    if k + "/w" in golden:
        w = golden[k + "/w"]
        qmax = O.FP8_MAX[kind] if kind else 127.0
        np.testing.assert_array_equal(O.absmax_scale(w, qmax, 0, dt), scale)
        if kind:
            np.testing.assert_array_equal(O.quantize_symmetric_fp8(w, scale, kind, dt), data)
        else:
            np.testing.assert_array_equal(O.quantize_symmetric_int8(w, scale, dt), data)
    if k + "/dequantized" in golden:
        np.testing.assert_array_equal(O.dequantize_qbytes_ref(data, scale, dt, kind), golden[k + "/dequantized"])


@pytest.mark.parametrize("tag", QBYTES_CASES)
def test_qbytes_mm_matches_reference(golden, tag):
    k = f"qbytes/{tag}"
    dt = _dt_of(tag)
    kind = _fp8_kind(tag)
    scale, data = golden[k + "/scale"], golden[k + "/data"]
    bias = golden.get(k + "/bias")
    Ms = sorted(int(key.rsplit("/x", 1)[1]) for key in golden if key.startswith(k + "/x"))
    for M in Ms:
        x, y, yop = golden[k + f"/x{M}"], golden[k + f"/y{M}"], golden[k + f"/yop{M}"]
        y_ref = O.qbytes_mm_ref(x, data, scale, dt, kind)
        y_ex = O.qbytes_mm_exact(x, data, scale, kind)
        if dt == "fp32":
            assert O.rel_max(y_ref, yop) < 1e-5
            assert O.rel_max(y_ex, yop) < 1e-5
        else:
            # the CPU bf16 x int8 branch (library/qbytes_mm.py:101-104) scales AFTER the product,
            # so it is closer to y_ex than to y_ref; both within the stated tolerance
            tol = {"fp16": 2e-3, "bf16": 1.2e-2}[dt]
            assert min(O.rel_max(y_ref, yop), O.rel_max(O.round_to(y_ex, dt), yop)) < tol
        if bias is not None:
            y_b = O.round_to(O.round_to(y_ex, dt) + bias, dt)
            assert O.rel_max(y_b, y) < {"fp32": 1e-5, "fp16": 2e-3, "bf16": 1.6e-2}[dt]
        else:
            np.testing.assert_array_equal(y, yop)


@pytest.mark.parametrize("dt", ["fp32", "fp16", "bf16"])
def test_qbytes_int8_int8(golden, dt):
    k = f"qbytes_i8i8/{dt}"
    y = O.qbytes_int_mm_ref(golden[k + "/a"], golden[k + "/b"], golden[k + "/scales"], dt)
    np.testing.assert_array_equal(y, golden[k + "/y"])


def test_fp8_codec_roundtrip():
    for kind in ("e4m3fn", "e4m3fnuz", "e5m2"):
        codes = np.arange(256, dtype=np.uint8)
        vals = O.fp8_decode(codes, kind)
        ok = np.isfinite(vals)
        enc = O.fp8_encode(vals[ok], kind)
        # -0.0 and +0.0 both decode to 0: compare values, not codes
        np.testing.assert_array_equal(O.fp8_decode(enc, kind), vals[ok])
    assert O.fp8_decode(np.array([0x7E], np.uint8))[0] == 448.0
    # ties to even: 17 is halfway between 16 (0x58) and 18 (0x59) in e4m3fn
    assert O.fp8_encode(np.array([17.0]))[0] == 0x58
    assert O.fp8_encode(np.array([19.0]))[0] == 0x5A


def test_group_ungroup_roundtrip():
    rng = np.random.default_rng(0)
    a = rng.standard_normal((6, 8, 4)).astype(np.float32)
    for axis in (0, -1):
        g = O.group(a, axis, 8)
        assert g.shape == ((24, 8) if axis == 0 else (8, 24))
        np.testing.assert_array_equal(O.ungroup(g, axis, a.shape), a)
    with pytest.raises(ValueError):
        O.group(a, 1, 8)
    with pytest.raises(ValueError):
        O.group(a, 0, 5)
