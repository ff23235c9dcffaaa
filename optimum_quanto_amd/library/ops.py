"""The ``quanto::`` operator library for this backend.

Schemas kept verbatim from the reference so existing call sites work unchanged:

* ``quanto::unpack(Tensor self, int bits) -> Tensor``                          library/unpack.py:18
* ``quanto::qbytes_mm(Tensor A, Tensor B, Tensor scales) -> Tensor``            library/qbytes_mm.py:22
* ``quanto::quantize_symmetric(Tensor base, ScalarType dtype, int? axis, Tensor scale) -> Tensor``   library/quantize.py:22-24
* ``quanto::quantize_affine(Tensor base, int bits, int axis, int? group_size, Tensor scale, Tensor shift) -> Tensor``  :58-61

New ops (the reference has no fused int4 product outside its CUDA-only AWQ/Marlin ops; the schema follows its own
template, library/extensions/README.md:24-48 and cuda/__init__.py:82-94):

* ``quanto::qbits_mm(Tensor input, Tensor packed, Tensor scale, Tensor shift, Tensor? bias, int bits, int? group_size,
  int out_features, int in_features) -> Tensor``
* ``quanto::dequantize_qbits(Tensor packed, Tensor scale, Tensor shift, int bits, int? group_size, int out_features,
  int in_features) -> Tensor``

Dispatch: the ``default`` implementations are plain torch and serve CPU tensors (quantize-time work and the CPU
plumbing config).  The ``CUDA`` key - which is what a ROCm device uses - always goes to ``libquanto_hip.so``;
if the library cannot be loaded the call raises: there is no silent fallback for device tensors.

When ``optimum.quanto`` itself is already imported in the process the ops exist; we then only (re)register the
``CUDA`` implementations, which is how this backend plugs into an unmodified reference install.
"""
from typing import Optional, Union

import os

import torch

from ..tensor.dtypes import dtype_info
from ..tensor.grouping import group, ungroup
from .hip import quanto_hip

__all__ = []

_lib_def = torch.library.Library("quanto", "FRAGMENT")
_lib_impl = torch.library.Library("quanto", "IMPL")


def _op_exists(name: str) -> bool:
    try:
        return hasattr(torch.ops.quanto, name) and getattr(torch.ops.quanto, name) is not None
    except (AttributeError, RuntimeError):
        return False


def _define(name: str, schema: str) -> bool:
    """Define ``quanto::name`` unless another definer (the reference package) already did. Returns True if we own it."""
    if _op_exists(name):
        return False
    _lib_def.define(name + schema)
    return True


def _impl(name: str, key: str, fn, owned: bool):
    if owned:
        _lib_impl.impl(name, fn, key)
    else:  # plug-in mode: replace the reference's registration for this key (intended: silence torch's override notice)
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                _lib_impl.impl(name, fn, key, allow_override=True)
            except TypeError:  # older torch without allow_override
                _lib_impl.impl(name, fn, key)


# ------------------------------------------------------------------------------------------------
# quanto::unpack
# ------------------------------------------------------------------------------------------------
def unpack_default(packed: torch.Tensor, bits: int) -> torch.Tensor:
    """Planes of ``bits`` bits, concatenated along dim 0 (library/unpack.py:21-54)."""
    planes = [(packed >> (bits * i)) & ((1 << bits) - 1) for i in range(8 // bits)]
    return torch.cat(planes).to(torch.uint8)


def unpack_hip(packed: torch.Tensor, bits: int) -> torch.Tensor:
    return quanto_hip.lib.unpack(packed, bits)


_owned = _define("unpack", "(Tensor self, int bits) -> Tensor")
if _owned:
    _impl("unpack", "CompositeExplicitAutograd", unpack_default, True)
_impl("unpack", "CUDA", unpack_hip, _owned)


# ------------------------------------------------------------------------------------------------
# quanto::qbytes_mm
# ------------------------------------------------------------------------------------------------
def _qbytes_mm_dense(activations, weights, output_scales):
    """Generic path (library/qbytes_mm.py:25-33): scale the weights, then a dense matmul."""
    activations = activations.to(output_scales.dtype)
    if weights.dtype.is_floating_point:
        weights = weights.to(output_scales.dtype)
    return torch.matmul(activations, (output_scales * weights).t())


def _qbytes_mm_int(activations, weights, output_scales):
    """int8 x int8 (library/qbytes_mm.py:36-50): exact int32 product, fp32 rescale."""
    k, n = activations.shape[-1], weights.shape[0]
    acc = torch._int_mm(activations.reshape(-1, k), weights.t()).reshape(activations.shape[:-1] + (n,))
    return (acc.to(torch.float32) * output_scales.t()).to(output_scales.dtype)


def qbytes_mm_default(activations, weights, output_scales):
    return _qbytes_mm_dense(activations, weights, output_scales)


def qbytes_mm_cpu(activations, weights, output_scales):
    """CPU selection logic of library/qbytes_mm.py:91-105."""
    if activations.dtype == torch.int8 and weights.dtype == torch.int8:
        return _qbytes_mm_int(activations, weights, output_scales)
    k = activations.shape[-1]
    if activations.dtype == torch.bfloat16 and weights.dtype == torch.int8 and k % 4 == 0:
        n = weights.shape[0]
        out = torch._weight_int8pack_mm(activations.reshape(-1, k), weights, output_scales.flatten())
        return out.reshape(activations.shape[:-1] + (n,))
    return _qbytes_mm_dense(activations, weights, output_scales)


def qbytes_mm_hip(activations, weights, output_scales, bias=None):
    """ROCm: one fused kernel, scales applied to the fp32 accumulator (csrc/qbytes_gemv.hip, csrc/qmm_mfma.hip)."""
    assert activations.ndim >= 1 and weights.ndim == 2
    n = weights.shape[0]
    if output_scales.numel() != n:
        # per-tensor weight scale or an exotic broadcast: expand to one scale per output feature
        output_scales = (output_scales * torch.ones((1, n), dtype=output_scales.dtype, device=output_scales.device))
        if output_scales.numel() != n:
            raise ValueError(f"qbytes_mm: cannot broadcast scales of shape {tuple(output_scales.shape)} to {n} features")
    return quanto_hip.lib.qbytes_mm(activations, weights, output_scales, bias)


def qbytes_mm_bias_default(activations, weights, output_scales, bias):
    """What tensor/weights/qbytes.py:73-81 computes: the product in the output dtype, then the bias added."""
    out = torch.ops.quanto.qbytes_mm(activations, weights, output_scales)
    return out if bias is None else out + bias


def qbytes_mm_bias_hip(activations, weights, output_scales, bias):
    return qbytes_mm_hip(activations, weights, output_scales, bias)


_owned = _define("qbytes_mm", "(Tensor A, Tensor B, Tensor scales) -> Tensor")
if _owned:
    _impl("qbytes_mm", "CompositeExplicitAutograd", qbytes_mm_default, True)
    _impl("qbytes_mm", "CPU", qbytes_mm_cpu, True)
_impl("qbytes_mm", "CUDA", qbytes_mm_hip, _owned)
# new op: the same product with the bias of the Linear fused into the kernel epilogue (rounded product + bias, rounded again:
# bit-identical to the two-op sequence) - saves one elementwise kernel per biased Linear
if _define("qbytes_mm_bias", "(Tensor A, Tensor B, Tensor scales, Tensor? bias) -> Tensor"):
    _impl("qbytes_mm_bias", "CompositeExplicitAutograd", qbytes_mm_bias_default, True)
    _impl("qbytes_mm_bias", "CUDA", qbytes_mm_bias_hip, True)


def qbytes_conv2d_default(input, weight, scales, bias, stride, padding, dilation):
    """What the reference computes for F.conv2d on a WeightQBytesTensor (nn/qconv2d.py:54-55 -> qfallback): dequantize, float convolution."""
    w = scales.reshape(-1, 1, 1, 1).to(input.dtype) * weight.to(input.dtype)
    groups = input.shape[1] // weight.shape[1]  # 1, or the channel count of a depthwise layer (weight [OC, 1, KH, KW])
    return torch.nn.functional.conv2d(input, w, bias, tuple(stride), tuple(padding), tuple(dilation), groups)


def qbytes_conv2d_hip(input, weight, scales, bias, stride, padding, dilation):
    return quanto_hip.lib.qbytes_conv2d(input, weight, scales, bias, tuple(stride), tuple(padding), tuple(dilation))


# new op: dense convolution with an int8 / fp8 weight as an implicit GEMM on the device (csrc/qconv_mfma.hip): no im2col tensor
if _define("qbytes_conv2d", "(Tensor input, Tensor weight, Tensor scales, Tensor? bias, int[] stride, int[] padding, int[] dilation) -> Tensor"):
    _impl("qbytes_conv2d", "CompositeExplicitAutograd", qbytes_conv2d_default, True)
    _impl("qbytes_conv2d", "CUDA", qbytes_conv2d_hip, True)


# ------------------------------------------------------------------------------------------------
# quanto::quantize_symmetric / quantize_affine (quantize-time, plain torch on every device)
# ------------------------------------------------------------------------------------------------
def _check_symmetric_args(base: torch.Tensor, axis: Union[int, None], scale: torch.Tensor) -> Union[int, None]:
    """Argument contract of library/quantize.py:26-49; returns the normalised axis (None, 0 or -1)."""
    if axis is None:
        if scale.ndim > 0:
            raise ValueError("Scale must be a scalar when quantizing per-tensor")
        return None
    if base.ndim == 1:
        raise ValueError("1D Tensors cannot be quantized per-axis")
    if axis == base.ndim - 1:
        axis = -1
    if axis not in (0, -1):
        raise ValueError("Quantization is only supported along the first or last axis.")
    if base.shape[axis] == 1:
        raise ValueError(f"Cannot quantize Tensor of shape {base.shape} along axis {axis} of size 1")
    if torch.squeeze(scale).ndim > 1:
        raise ValueError("Quantizing along multiple axis is not supported")
    if scale.ndim != base.ndim:
        raise ValueError(
            "When quantizing per-axis, the scale must be broadcastable to the base (Tip: try to add missing dims of length zero).")
    return axis


def quantize_symmetric(base: torch.Tensor, dtype: torch.dtype, axis: Union[int, None], scale: torch.Tensor) -> torch.Tensor:
    """clamp(round(base / scale)) to ``dtype`` (library/quantize.py:26-55; float8 targets are not rounded first)."""
    _check_symmetric_args(base, axis, scale)
    data = base / scale
    if not dtype.is_floating_point:
        data = torch.round(data)
    info = dtype_info(dtype)
    return torch.clamp(data, min=info.min, max=info.max).to(dtype)


def quantize_symmetric_hip(base: torch.Tensor, dtype: torch.dtype, axis: Union[int, None], scale: torch.Tensor) -> torch.Tensor:
    """Device tensors: the one-pass kernel (csrc/quantize.hip) for int8 / OCP float8 targets; the formats the hardware
    converters do not produce (e4m3fnuz) and non-float bases keep the elementwise torch sequence - still on the device."""
    axis = _check_symmetric_args(base, axis, scale)
    lib = quanto_hip.lib
    if dtype in lib.QUANTIZE_TARGETS and base.dtype in (torch.float32, torch.float16, torch.bfloat16) and (
            axis is None or scale.numel() == base.shape[axis]):
        return lib.quantize_symmetric(base, dtype, axis, scale)
    return quantize_symmetric(base, dtype, axis, scale)


def quantize_affine(base: torch.Tensor, bits: int, axis: int, group_size: Union[int, None], scale: torch.Tensor,
                    shift: torch.Tensor) -> torch.Tensor:
    """uint8 in [0, 2^bits): round((base + shift) / scale), or round(base / scale) + zero-point (library/quantize.py:66-78)."""
    if axis not in (0, -1):
        raise ValueError("axis parameter must be 0 (first axis) or -1 (last axis)")
    if group_size is not None:
        base = group(base, axis=axis, group_size=group_size)
    if shift.dtype.is_floating_point:
        data = torch.round((base + shift) / scale)
    else:
        data = torch.round(base / scale) + shift
    return torch.clamp(data, min=0, max=2**bits - 1).to(torch.uint8)


if _define("quantize_symmetric", "(Tensor base, ScalarType dtype, int? axis, Tensor scale) -> Tensor"):
    _impl("quantize_symmetric", "CompositeExplicitAutograd", quantize_symmetric, True)
    _impl("quantize_symmetric", "CUDA", quantize_symmetric_hip, True)
else:
    _impl("quantize_symmetric", "CUDA", quantize_symmetric_hip, False)
def quantize_affine_hip(base: torch.Tensor, bits: int, axis: int, group_size: Union[int, None], scale: torch.Tensor,
                        shift: torch.Tensor) -> torch.Tensor:
    """Device tensors: the one-pass kernel (csrc/quantize.hip) for the layout of the hot path - axis-0 2-D weights with
    one scale/shift per group; every other case keeps the torch sequence on the device."""
    if axis not in (0, -1):
        raise ValueError("axis parameter must be 0 (first axis) or -1 (last axis)")
    if (axis == 0 and base.ndim == 2 and bits in (2, 4) and base.dtype in (torch.float32, torch.float16, torch.bfloat16)
            and (group_size is None or base.shape[1] % group_size == 0)):
        rows = base.numel() // (group_size or base.shape[1])
        if scale.numel() == rows and shift.numel() == rows and shift.dtype in (base.dtype, torch.uint8, torch.int8):
            return quanto_hip.lib.quantize_affine(base, bits, group_size, scale, shift)
    return quantize_affine(base, bits, axis, group_size, scale, shift)


if _define("quantize_affine", "(Tensor base, int bits, int axis, int? group_size, Tensor scale, Tensor shift) -> Tensor"):
    _impl("quantize_affine", "CompositeExplicitAutograd", quantize_affine, True)
    _impl("quantize_affine", "CUDA", quantize_affine_hip, True)
else:
    _impl("quantize_affine", "CUDA", quantize_affine_hip, False)


# ------------------------------------------------------------------------------------------------
# quanto::dequantize_qbits and quanto::qbits_mm (new ops)
# ------------------------------------------------------------------------------------------------
def dequantize_qbits_default(packed, scale, shift, bits: int, group_size: Optional[int], out_features: int, in_features: int):
    """Reference sequence: unpack, trim, remove shift, scale, ungroup (tensor/packed.py:101-104, tensor/qbits.py:27-49)."""
    rows = (out_features * in_features) // group_size if group_size is not None else out_features
    data = torch.ops.quanto.unpack(packed, bits)[:rows]
    if not shift.dtype.is_floating_point:
        data = data.to(torch.int8) - shift.to(torch.int8)
    out = scale * data
    if shift.dtype.is_floating_point:
        out -= shift
    return ungroup(out, axis=0, orig_shape=torch.Size([out_features, in_features]))


def dequantize_qbits_hip(packed, scale, shift, bits: int, group_size: Optional[int], out_features: int, in_features: int):
    return quanto_hip.lib.dequantize_qbits(packed, scale, shift, bits, group_size, out_features, in_features)


def qbits_mm_default(input, packed, scale, shift, bias, bits: int, group_size: Optional[int], out_features: int, in_features: int):
    """x @ dequantize(W).T (+ bias): what tensor/function.py:41-47 computes through qfallback."""
    w = torch.ops.quanto.dequantize_qbits(packed, scale, shift, bits, group_size, out_features, in_features)
    out = torch.matmul(input, w.t())
    return out if bias is None else out + bias


def qbits_mm_hip(input, packed, scale, shift, bias, bits: int, group_size: Optional[int], out_features: int, in_features: int):
    return quanto_hip.lib.qbits_mm(input, packed, scale, shift, bias, bits, group_size, out_features, in_features)


if _define("dequantize_qbits",
           "(Tensor packed, Tensor scale, Tensor shift, int bits, int? group_size, int out_features, int in_features) -> Tensor"):
    _impl("dequantize_qbits", "CompositeExplicitAutograd", dequantize_qbits_default, True)
    _impl("dequantize_qbits", "CUDA", dequantize_qbits_hip, True)
if _define("qbits_mm",
           "(Tensor input, Tensor packed, Tensor scale, Tensor shift, Tensor? bias, int bits, int? group_size, "
           "int out_features, int in_features) -> Tensor"):
    _impl("qbits_mm", "CompositeExplicitAutograd", qbits_mm_default, True)
    _impl("qbits_mm", "CUDA", qbits_mm_hip, True)


# ------------------------------------------------------------------------------------------------
# quanto::qbits_mm_a8 (r6): F.linear(quantized activation, int4 weight) without dequantizing the activation
# ------------------------------------------------------------------------------------------------
def qbits_mm_a8_default(input, input_scale, packed, scale, shift, bias, bits: int, group_size: Optional[int], out_features: int, in_features: int):
    """What the reference computes (tensor/weights/awq/qbits.py:57-58, tensor/function.py:41-47): dequantize the activation, then the float product."""
    x = None
    if input.is_cuda:  # r6: cast + multiply in one pass (bit-identical: csrc/quantize.hip dequantize_symmetric)
        x = quanto_hip.lib.dequantize_symmetric(input, input_scale.to(scale.dtype))
    if x is None:
        x = input.to(scale.dtype) * input_scale.to(scale.dtype)
    return torch.ops.quanto.qbits_mm(x, packed, scale, shift, bias, bits, group_size, out_features, in_features)


_A8_MAX_TILES = int(os.environ.get("QUANTO_HIP_A8_MAX_TILES", "512")) if os.environ.get("QUANTO_HIP_EXPERIMENT", "0") not in ("", "0") else 512


def qbits_mm_a8_hip(input, input_scale, packed, scale, shift, bias, bits: int, group_size: Optional[int], out_features: int, in_features: int):
    lib = quanto_hip.lib
    m = input.numel() // in_features if in_features else 0
    # batched-decode sizes keep the weight-streaming kernels (the activation is dequantized: M x K elements, nothing next to the weight stream);
    # from 64 rows on the stored integers / fp8 values go to the 8-bit matrix instructions
    # ... while the output's 128 x 128 tiles are all resident at once (two workgroups per CU): the kernel moves 24 KiB through a CU's vector L1 per tile
    # and group and is bound by that, not by the matrix pipe; beyond one residency round the dequantize-first sequence on the dense bf16 GEMM is faster
    # (r6 sweep, profiles/r06_w4a8_crossover.jsonl: (2048,4096,4096) 93 vs 109 us, (4096,4096,4096) 184 vs 146, (768,14336,4096) 137 vs 114)
    tiles = -(-m // 128) * -(-out_features // 128)
    if (64 < m and tiles <= _A8_MAX_TILES and input.dtype in lib.A8_DTYPES and input_scale.numel() == 1
            and lib.qbits_mm_a8_workspace(m, out_features, in_features, bits, group_size, input.dtype, scale.dtype) >= 0):
        return lib.qbits_mm_a8(input, input_scale, packed, scale, shift, bias, bits, group_size, out_features, in_features)
    return qbits_mm_a8_default(input, input_scale, packed, scale, shift, bias, bits, group_size, out_features, in_features)


if _define("qbits_mm_a8",
           "(Tensor input, Tensor input_scale, Tensor packed, Tensor scale, Tensor shift, Tensor? bias, int bits, int? group_size, "
           "int out_features, int in_features) -> Tensor"):
    _impl("qbits_mm_a8", "CompositeExplicitAutograd", qbits_mm_a8_default, True)
    _impl("qbits_mm_a8", "CUDA", qbits_mm_a8_hip, True)


def qbits_conv2d_default(input, packed, scale, shift, bias, bits: int, group_size: Optional[int], weight_size, stride, padding, dilation):
    """What the reference computes for F.conv2d on a WeightQBitsTensor (nn/qconv2d.py:54-55 -> qfallback): dequantize, float convolution."""
    oc, c, kh, kw = weight_size
    w = torch.ops.quanto.dequantize_qbits(packed, scale, shift, bits, group_size, oc, c * kh * kw).reshape(oc, c, kh, kw)
    return torch.nn.functional.conv2d(input, w.to(input.dtype), bias, tuple(stride), tuple(padding), tuple(dilation), 1)


def qbits_conv2d_hip(input, packed, scale, shift, bias, bits: int, group_size: Optional[int], weight_size, stride, padding, dilation):
    return quanto_hip.lib.qbits_conv2d(input, packed, scale, shift, bias, bits, group_size, tuple(weight_size), tuple(stride), tuple(padding),
                                       tuple(dilation))


# new op: dense convolution with a packed int4 weight as an implicit GEMM on the device (csrc/qconv_mfma.hip, W_I4R staging): no im2col tensor,
# no dequantized weight in memory
if _define("qbits_conv2d",
           "(Tensor input, Tensor packed, Tensor scale, Tensor shift, Tensor? bias, int bits, int? group_size, int[] weight_size, "
           "int[] stride, int[] padding, int[] dilation) -> Tensor"):
    _impl("qbits_conv2d", "CompositeExplicitAutograd", qbits_conv2d_default, True)
    _impl("qbits_conv2d", "CUDA", qbits_conv2d_hip, True)


# several Linears applied to the same input in one launch (q/k/v, gate/up of a decoder layer at decode time)
def qbits_mm_multi_default(input, packed, scale, shift, bias, bits: int, group_size: Optional[int], out_features, in_features: int):
    return [torch.ops.quanto.qbits_mm(input, packed[i], scale[i], shift[i], bias[i], bits, group_size, out_features[i], in_features)
            for i in range(len(packed))]


def qbits_mm_multi_hip(input, packed, scale, shift, bias, bits: int, group_size: Optional[int], out_features, in_features: int):
    return quanto_hip.lib.qbits_mm_multi(input, packed, scale, shift, bias, bits, group_size, list(out_features), in_features)


if _define("qbits_mm_multi",
           "(Tensor input, Tensor[] packed, Tensor[] scale, Tensor[] shift, Tensor?[] bias, int bits, int? group_size, "
           "int[] out_features, int in_features) -> Tensor[]"):
    _impl("qbits_mm_multi", "CompositeExplicitAutograd", qbits_mm_multi_default, True)
    _impl("qbits_mm_multi", "CUDA", qbits_mm_multi_hip, True)


def qbytes_mm_multi_default(activations, weights, output_scales, bias):
    return [torch.ops.quanto.qbytes_mm_bias(activations, weights[i], output_scales[i], bias[i]) for i in range(len(weights))]


def qbytes_mm_multi_hip(activations, weights, output_scales, bias):
    return quanto_hip.lib.qbytes_mm_multi(activations, list(weights), list(output_scales), list(bias))


# the 8-bit counterpart: several WeightQBytes Linears applied to the same (float) input in one launch
if _define("qbytes_mm_multi", "(Tensor A, Tensor[] B, Tensor[] scales, Tensor?[] bias) -> Tensor[]"):
    _impl("qbytes_mm_multi", "CompositeExplicitAutograd", qbytes_mm_multi_default, True)
    _impl("qbytes_mm_multi", "CUDA", qbytes_mm_multi_hip, True)
