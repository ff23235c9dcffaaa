"""Quantized weight tensors and ``quantize_weight``.

API mirror of optimum/quanto/tensor/weights/{qbytes,qbits,quantization}.py.  What differs is what happens on
``F.linear``:

* ``WeightQBytesTensor``: same as the reference - ``quanto::qbytes_mm(input, data, scale)``
  (weights/qbytes.py:68-82) - but on a ROCm device the op is one fused HIP kernel (csrc/qbytes_gemv.hip,
  csrc/qmm_mfma.hip) instead of "materialise scale*W, then matmul" (library/qbytes_mm.py:25-33).
* ``WeightQBitsTensor``: the reference dequantizes the whole weight on every call (tensor/function.py:41-47 via
  qfallback) unless a CUDA-only AWQ/TinyGemm subclass was selected by ``create()`` (weights/qbits.py:97-136).
  Here every axis-0 2-D int4/int2 weight calls ``quanto::qbits_mm`` which consumes the *generic* PackedTensor
  layout directly, so no device-specific re-packing subclass is needed: ``create()`` keeps the serialisable
  layout, ``optimize()`` and ``weight_qbits_tensor()`` are identities, and checkpoints stay in the
  kernel-agnostic format (weights/qbits.py:223-235).
"""
import ast
from typing import Optional

import torch
from torch.autograd import Function

from .base import QBitsTensor, QBytesTensor, qfallback
from .dtypes import qint2, qint4, qtype, qtypes
from .grouping import grouped_shape
from .linear_function import QuantizedLinearFunction
from .packing import PackedTensor

__all__ = ["WeightQBytesTensor", "WeightQBitsTensor", "quantize_weight"]


_OPS = {}


def _op(name: str):
    """``torch.ops.quanto.<name>.default``, resolved once."""
    op = _OPS.get(name)
    if op is None:
        op = _OPS[name] = getattr(torch.ops.quanto, name).default
    return op


class _NoCtx:
    """Stands in for the autograd context when a linear function's ``forward`` is called directly (nothing is saved: no backward will run)."""

    @staticmethod
    def save_for_backward(*tensors):
        pass


_NO_CTX = _NoCtx()


def _wants_grad(input, weight, bias) -> bool:
    return torch.is_grad_enabled() and (input.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad))


def _pair(v):
    """``int`` or 1- / 2-element sequence -> [h, w] (torch accepts ``stride=(2,)`` for a 2-D convolution)."""
    if isinstance(v, int):
        return [v, v]
    v = [int(e) for e in v]
    return v * 2 if len(v) == 1 else v


def conv2d_patches(input, kernel_size, stride, padding, dilation):
    """im2col: NCHW ``input`` -> ([B*L, C*kh*kw] rows in the weight's (c, i, j) order, output height, output width)."""
    pair = lambda v: tuple(_pair(v))  # noqa: E731
    (kh, kw), stride, padding, dilation = pair(kernel_size), pair(stride), pair(padding), pair(dilation)
    b, c, h, w = input.shape
    oh = (h + 2 * padding[0] - dilation[0] * (kh - 1) - 1) // stride[0] + 1
    ow = (w + 2 * padding[1] - dilation[1] * (kw - 1) - 1) // stride[1] + 1
    if kh == 1 and kw == 1 and stride == (1, 1) and padding == (0, 0):
        a = input.permute(0, 2, 3, 1).reshape(b * h * w, c)  # pointwise convolution: no patches to gather
    else:
        cols = torch.nn.functional.unfold(input, (kh, kw), dilation=dilation, padding=padding, stride=stride)  # [B, K, L]
        a = cols.transpose(1, 2).reshape(b * oh * ow, c * kh * kw)
    return a.contiguous(), oh, ow


def _implicit_conv2d(input, weight, scale, bias, stride, padding, dilation, groups):
    """``quanto::qbytes_conv2d`` - the convolution as an implicit GEMM, im2col inside the kernel's staging loads (r4) - when the call is
    eligible: dense (groups = 1), batched NCHW 16-bit input on a ROCm device, int8 / OCP fp8 weight, windows of up to 127 taps (r5: any
    C*kh*kw), no gradient wanted.  None otherwise: the caller then lowers to a materialised im2col + GEMM (conv2d_as_gemm) or keeps the reference behaviour."""
    from ..library.hip import quanto_hip

    if isinstance(padding, str) or type(input) is not torch.Tensor or input.dim() != 4 or input.device.type != "cuda":
        return None
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (input, weight, bias)):
        return None
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    pair = _pair
    if groups != 1:
        # r6: depthwise layers (groups = in_channels, weight [OC, 1, KH, KW]) have their own stencil kernel; other groupings keep the reference behaviour
        if (weight.dim() == 4 and groups == input.shape[1] and weight.shape[1] == 1
                and quanto_hip.lib.qbytes_conv2d_depthwise_supported(input, weight._data, stride, padding, dilation)):
            return torch.ops.quanto.qbytes_conv2d(input, weight._data, scale, bias, pair(stride), pair(padding), pair(dilation))
        return None
    if weight.dim() != 4 or input.shape[1] != weight.shape[1] or not quanto_hip.lib.qbytes_conv2d_supported(input, weight._data, stride, padding, dilation):
        return None
    # (until r5 pointwise convolutions with ONE K-tile - fewer than 128 input channels - went to a permuted view + the tuned GEMM kernels: 21.5 vs 25.0 us at
    # (8,64,56,56) -> 256; with the r5 gather and epilogue the convolution kernel takes that shape in 12.0 us: profiles/r05_qconv2d_paths_grid.jsonl)
    return torch.ops.quanto.qbytes_conv2d(input, weight._data, scale, bias, pair(stride), pair(padding), pair(dilation))


def _implicit_conv2d_qbits(input, weight, bias, stride, padding, dilation, groups):
    """``quanto::qbits_conv2d`` - the same implicit GEMM for a packed int4 weight, dequantized with the reference's roundings while it is staged
    (r4) - under the conditions of _implicit_conv2d; None otherwise (im2col + qbits_mm, or the reference behaviour)."""
    from ..library.hip import quanto_hip

    if groups != 1 or isinstance(padding, str) or type(input) is not torch.Tensor or input.dim() != 4 or input.device.type != "cuda":
        return None
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (input, weight, bias)):
        return None
    packed = weight._data
    if weight.dim() != 4 or input.shape[1] != weight.shape[1] or input.dtype != weight._scale.dtype:
        return None
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    if not quanto_hip.lib.qbits_conv2d_supported(input, tuple(weight.shape), packed.bits, weight._group_size, stride, padding, dilation):
        return None
    pair = _pair
    return torch.ops.quanto.qbits_conv2d(input, packed._data, weight._scale, weight._shift, bias, packed.bits, weight._group_size,
                                         list(weight.shape), pair(stride), pair(padding), pair(dilation))


def conv2d_as_gemm(input, weight, bias, stride, padding, dilation, groups, gemm):
    """``F.conv2d`` with a quantized [N, C, kh, kw] weight as im2col + one fused GEMM on the device.

    The reference has no kernel here: ``QConv2d.forward`` (nn/qconv2d.py:54-55) reaches aten::convolution with a
    QTensor, which dequantizes the whole weight on every call (qfallback).  An axis-0 quantized convolution weight is,
    byte for byte, the [N, K = C*kh*kw] operand of ``quanto::qbytes_mm`` / ``quanto::qbits_mm`` (per-channel scales;
    sub-byte groups run along the flattened K), so a dense convolution is ``conv2d_patches(x)`` [B*L, K] times that
    operand.  Returns None when the call is not eligible (grouped convolution, string padding, not a ROCm device, not
    batched NCHW input); the caller then keeps the reference behaviour.  ``gemm(a)`` maps [B*L, K] -> [B*L, N] (+ bias).
    """
    if groups != 1 or isinstance(padding, str) or input.dim() != 4 or input.device.type != "cuda" or weight.dim() != 4:
        return None
    # The fused GEMM ops are opaque to autograd (ctypes kernels, no backward registered): whenever a gradient is wanted -
    # QAT, calibration with grad enabled, an unfrozen module - keep the reference's differentiable dequantize + convolution.
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (input, weight, bias)):
        return None
    if type(input) is not torch.Tensor:
        input = input.dequantize()
    n, c, kh, kw = weight.shape
    if input.shape[1] != c:
        return None
    a, oh, ow = conv2d_patches(input, (kh, kw), stride, padding, dilation)
    y = gemm(a)  # [B*L, N]
    return y.view(input.shape[0], oh * ow, n).permute(0, 2, 1).reshape(input.shape[0], n, oh, ow)


# ================================================================================================
# 8-bit weights
# ================================================================================================
class _QuantizeBytesWeight(Function):
    @staticmethod
    def forward(ctx, base, qtype, axis, scale, activation_qtype, optimized):
        if qtype.bits != 8:
            raise ValueError("QBytesTensor can only be of 8-bit qtype")
        data = torch.ops.quanto.quantize_symmetric(base, dtype=qtype.dtype, axis=axis, scale=scale)
        make = WeightQBytesTensor.create if optimized else WeightQBytesTensor
        return make(qtype, axis, size=base.size(), stride=base.stride(), data=data, scale=scale,
                    activation_qtype=activation_qtype)

    @staticmethod
    def backward(ctx, grad):
        return grad, None, None, None, None, None


class WeightQBytesLinearFunction(QuantizedLinearFunction):
    """F.linear on an int8 / float8 weight: one ``quanto::qbytes_mm`` call (weights/qbytes.py:68-82)."""

    @staticmethod
    def forward(ctx, input, other, bias=None):
        ctx.save_for_backward(input, other)
        if isinstance(input, QBytesTensor):
            # quantized activations: integer/fp8 product, rescaled by the product of both scales
            return torch.ops.quanto.qbytes_mm_bias(input._data, other._data, input._scale * other._scale, bias)
        k, n = input.shape[-1], other.shape[0]
        output = _op("qbytes_mm_bias")(input.reshape(-1, k), other._data, other._scale, bias)
        return output.reshape(input.shape[:-1] + (n,))


class WeightQBytesTensor(QBytesTensor):
    @staticmethod
    def create(qtype, axis, size, stride, data, scale, activation_qtype: Optional[qtype] = None, requires_grad=False):
        """Factory used by quantize / ``.to(device)`` / reload (selection point of weights/qbytes.py:86-143).

        The MI355X kernels read the plain [N, K] byte layout, so the generic class is already the optimized one.
        """
        return WeightQBytesTensor(qtype, axis, size, stride, data, scale, activation_qtype, requires_grad)

    @staticmethod
    def __new__(cls, qtype, axis, size, stride, data, scale, activation_qtype, requires_grad=False):
        assert data.device == scale.device
        return torch.Tensor._make_wrapper_subclass(
            cls, size, strides=stride, dtype=scale.dtype, device=data.device, requires_grad=requires_grad)

    def __init__(self, qtype, axis, size, stride, data, scale, activation_qtype, requires_grad=False):
        super().__init__(qtype, axis, size, stride, data, scale, requires_grad=requires_grad)
        self.activation_qtype = activation_qtype

    @classmethod
    def quantize(cls, base, qtype, axis, scale, activation_qtype: Optional[qtype] = None, optimized: Optional[bool] = True):
        return _QuantizeBytesWeight.apply(base, qtype, axis, scale, activation_qtype, optimized)

    # -- (de)serialization --------------------------------------------------------------------------
    @staticmethod
    def load_from_state_dict(state_dict, prefix, qtype, axis, size, stride, activation_qtype, missing_keys):
        inner, missing = {}, False
        for name in ("_data", "_scale"):
            if prefix + name in state_dict:
                inner[name] = state_dict.pop(prefix + name)
            else:
                missing_keys.append(prefix + name)
                missing = True
        if missing:
            return None
        meta = {"qtype": qtype.name, "axis": str(axis), "size": str(list(size)), "stride": str(list(stride)),
                "activation_qtype": "none" if activation_qtype is None else activation_qtype.name}
        return WeightQBytesTensor.__tensor_unflatten__(inner, meta, None, None)

    def optimize(self):
        return self

    def weight_qbytes_tensor(self):
        return self

    def save_to_state_dict(self, destination, prefix, keep_vars):
        super().save_to_state_dict(destination, prefix, keep_vars)

    def __tensor_flatten__(self):
        meta = {"qtype": self._qtype.name, "axis": str(self._axis), "size": str(list(self.size())),
                "stride": str(list(self.stride())),
                "activation_qtype": "none" if self.activation_qtype is None else self.activation_qtype.name}
        return ["_data", "_scale"], meta

    @staticmethod
    def __tensor_unflatten__(inner_tensors, meta, outer_size, outer_stride):
        assert len(inner_tensors) == 2 and len(meta) == 5
        act = None if meta["activation_qtype"] == "none" else qtypes[meta["activation_qtype"]]
        return WeightQBytesTensor(qtypes[meta["qtype"]], ast.literal_eval(meta["axis"]), ast.literal_eval(meta["size"]),
                                  ast.literal_eval(meta["stride"]), inner_tensors["_data"], inner_tensors["_scale"], act)

    # -- dispatch -----------------------------------------------------------------------------------
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.nn.functional.linear:
            def qlinear(input, other, bias=None):
                if not _wants_grad(input, other, bias):  # inference: the op itself, no autograd.Function node around it (~3 us per call)
                    return WeightQBytesLinearFunction.forward(_NO_CTX, input, other, bias)
                return WeightQBytesLinearFunction.apply(input, other, bias)

            return qlinear(*args, **kwargs)
        if func is torch.nn.functional.conv2d:
            def qconv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
                if not isinstance(weight, WeightQBytesTensor) or weight.axis not in (0, None):
                    return None
                n = weight.shape[0]
                scale = weight._scale.reshape(-1, 1).expand(n, 1).contiguous()  # per-channel [N,1,1,1] or per-tensor
                implicit = _implicit_conv2d(input, weight, scale, bias, stride, padding, dilation, groups)
                if implicit is not None:
                    return implicit
                data = weight._data.reshape(n, -1)
                return conv2d_as_gemm(input, weight, bias, stride, padding, dilation, groups,
                                      lambda a: torch.ops.quanto.qbytes_mm_bias(a, data, scale, bias))

            out = qconv2d(*args, **kwargs)
            if out is not None:
                return out
        if func is torch.equal:
            a, b = args
            return a.equal(b)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)

    @classmethod
    def __torch_dispatch__(cls, op, types, args, kwargs=None):
        kwargs = dict(kwargs or {})
        op = op.overloadpacket
        if op is torch.ops.aten.detach:
            t = args[0]
            names, meta = t.__tensor_flatten__()
            return cls.__tensor_unflatten__({n: op(getattr(t, n)) for n in names}, meta, t.size(), t.stride())
        if op in (torch.ops.aten._to_copy, torch.ops.aten.to):
            t = args[0]
            dtype = kwargs.pop("dtype", t.dtype)
            device = kwargs.pop("device", t.device)
            if dtype is not None and dtype != t.dtype:
                raise ValueError("The dtype of a weights Tensor cannot be changed")
            data = op(t._data, device=device, **kwargs)
            scale = op(t._scale, device=device, **kwargs)
            return WeightQBytesTensor.create(t.qtype, t.axis, t.size(), t.stride(), data, scale,
                                             activation_qtype=t.activation_qtype, requires_grad=t.requires_grad)
        if op is torch.ops.aten.t and cls is WeightQBytesTensor:
            t = args[0]
            rows, cols = t.size()
            scale, axis = t._scale, t.axis
            if axis is not None:
                scale = op(scale)
                axis = 0 if axis == -1 else -1
            return WeightQBytesTensor(t.qtype, axis, torch.Size([cols, rows]), t.stride()[::-1], op(t._data), scale,
                                      t.activation_qtype)
        return qfallback(op, *args, **kwargs)


# ================================================================================================
# sub-byte weights
# ================================================================================================
def _quantize_affine_packed(base, bits, axis, group_size, scale, shift):
    """Device weights in the hot-path layout (axis 0, one scale / shift per group): quantize and pack in ONE kernel
    (csrc/quantize.hip::quantize_affine_pack_kernel) and wrap the bytes as the PackedTensor the two-step path would have
    produced (library/quantize.py:66-78 + tensor/packed.py:24-69).  None when not eligible."""
    if not (base.is_cuda and axis == 0 and base.ndim >= 2 and base.dtype in (torch.float32, torch.float16, torch.bfloat16)):
        return None
    n, k = base.shape[0], base.numel() // base.shape[0]
    if group_size is not None and k % group_size != 0:
        return None
    rows = n * k // (group_size or k)
    if scale.numel() != rows or shift.numel() != rows or shift.dtype not in (base.dtype, torch.uint8, torch.int8):
        return None
    from ..library.hip import quanto_hip  # raises if the HIP library is missing: no silent fallback for device tensors

    packed = quanto_hip.lib.quantize_affine_packed(base, bits, group_size, scale, shift)
    size = torch.Size(grouped_shape(base.shape, axis, group_size)) if group_size is not None else base.size()
    if group_size is None:
        packed = packed.reshape((packed.shape[0],) + tuple(base.shape[1:]))
    stride = torch.empty(size, device="meta").stride()
    return PackedTensor(packed, bits, size, stride)


class _QuantizeBitsWeight(Function):
    @staticmethod
    def forward(ctx, base, qtype, axis, group_size, scale, shift, optimized):
        if qtype not in (qint2, qint4):
            raise ValueError("WeightQBitsTensor can only be of qint2 or qint4 qtype")
        if axis not in (0, -1):
            raise ValueError("WeightQBitsTensor axis parameter must be 0 (first axis) or -1 (last axis)")
        data = _quantize_affine_packed(base, qtype.bits, axis, group_size, scale, shift)
        if data is None:
            data = torch.ops.quanto.quantize_affine(base, bits=qtype.bits, axis=axis, group_size=group_size, scale=scale,
                                                    shift=shift)
        make = WeightQBitsTensor.create if optimized else WeightQBitsTensor
        return make(qtype, axis, group_size, base.size(), base.stride(), data, scale, shift)

    @staticmethod
    def backward(ctx, grad):
        return grad, None, None, None, None, None, None


def _fusable(w) -> bool:
    """True when ``quanto::qbits_mm`` can consume this weight (axis-0, 2-D, integer qtype, packed data)."""
    return w.axis == 0 and w.ndim == 2 and isinstance(w._data, PackedTensor) and not w.qtype.is_floating_point


class WeightQBitsLinearFunction(QuantizedLinearFunction):
    """F.linear on an int4 / int2 weight through the fused ``quanto::qbits_mm`` op.

    Same structure as the reference's optimized-subclass functions (weights/awq/qbits.py:53-74): dequantize a
    quantized activation first, hand the packed data + scale + shift to the kernel, add the bias.
    """

    @staticmethod
    def forward(ctx, input, other, bias=None):
        ctx.save_for_backward(input, other)
        n, k = other.shape
        if type(input) is not torch.Tensor:
            # r6: a per-tensor quantized activation meets the packed weight as stored (quanto::qbits_mm_a8: the 8-bit matrix instructions from 64
            # rows on, the reference's dequantize-first sequence below that and for the formats the kernel does not take)
            if isinstance(input, QBytesTensor) and input.axis is None and input._data.is_cuda and input._scale.numel() == 1:
                return _op("qbits_mm_a8")(input._data, input._scale, other._data._data, other._scale, other._shift, bias, other._data.bits,
                                          other._group_size, n, k)
            input = input.dequantize()
        # the resolved overload: torch.ops.quanto.qbits_mm(...) looks the overload up on every call (~1 us of a 15 us decode call)
        output = _op("qbits_mm")(input, other._data._data, other._scale, other._shift, bias, other._data.bits, other._group_size, n, k)
        return output


class WeightQBitsTensor(QBitsTensor):
    @staticmethod
    def create(qtype, axis, group_size, size, stride, data, scale, shift, requires_grad=False):
        """Factory / selection point (weights/qbits.py:66-138): the generic layout *is* the MI355X layout."""
        return WeightQBitsTensor(qtype, axis, group_size, size, stride, data, scale, shift, requires_grad)

    @staticmethod
    def __new__(cls, qtype, axis, group_size, size, stride, data, scale, shift, requires_grad=False):
        assert data.device == scale.device
        assert data.device == shift.device
        return torch.Tensor._make_wrapper_subclass(
            cls, size, strides=stride, dtype=scale.dtype, device=data.device, requires_grad=requires_grad)

    def __init__(self, qtype, axis, group_size, size, stride, data, scale, shift, requires_grad=False):
        if type(data) is torch.Tensor:
            data = PackedTensor.pack(data, qtype.bits)
        super().__init__(qtype, axis, group_size, size, stride, data, scale, shift)

    @classmethod
    def quantize(cls, base, qtype, axis, group_size, scale, shift, optimized: Optional[bool] = True):
        return _QuantizeBitsWeight.apply(base, qtype, axis, group_size, scale, shift, optimized)

    # -- (de)serialization --------------------------------------------------------------------------
    @staticmethod
    def load_from_state_dict(state_dict, prefix, qtype, axis, group_size, size, stride, missing_keys):
        if group_size is None:
            data_size, data_stride = size, stride
        else:
            data_size = grouped_shape(size, axis, group_size)
            data_stride = (data_size[1], 1)
        inner = {"_data": PackedTensor.load_from_state_dict(state_dict, prefix + "_data.", qtype.bits, data_size,
                                                            data_stride, missing_keys=missing_keys)}
        missing = inner["_data"] is None
        for name in ("_scale", "_shift"):
            if prefix + name in state_dict:
                inner[name] = state_dict.pop(prefix + name)
            else:
                missing_keys.append(prefix + name)
                missing = True
        if missing:
            return None
        meta = {"qtype": qtype.name, "axis": str(axis), "group_size": str(group_size), "size": str(list(size)),
                "stride": str(list(stride))}
        return WeightQBitsTensor.__tensor_unflatten__(inner, meta, None, None)

    def optimize(self):
        return self

    def weight_qbits_tensor(self):
        return self

    def __tensor_flatten__(self):
        meta = {"qtype": self._qtype.name, "axis": str(self._axis), "group_size": str(self._group_size),
                "size": str(list(self.size())), "stride": str(list(self.stride()))}
        return ["_data", "_scale", "_shift"], meta

    @staticmethod
    def __tensor_unflatten__(inner_tensors, meta, outer_size, outer_stride):
        assert len(inner_tensors) == 3 and len(meta) == 5
        return WeightQBitsTensor(qtypes[meta["qtype"]], ast.literal_eval(meta["axis"]), ast.literal_eval(meta["group_size"]),
                                 ast.literal_eval(meta["size"]), ast.literal_eval(meta["stride"]), inner_tensors["_data"],
                                 inner_tensors["_scale"], inner_tensors["_shift"])

    # -- dispatch -----------------------------------------------------------------------------------
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.nn.functional.linear:
            def qlinear(input, other, bias=None):
                if _fusable(other):
                    if not _wants_grad(input, other, bias):  # inference: see WeightQBytesTensor
                        return WeightQBitsLinearFunction.forward(_NO_CTX, input, other, bias)
                    return WeightQBitsLinearFunction.apply(input, other, bias)
                return QuantizedLinearFunction.apply(input, other, bias)

            return qlinear(*args, **kwargs)
        if func is torch.nn.functional.conv2d:
            def qconv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
                if not (isinstance(weight, WeightQBitsTensor) and weight.axis == 0 and isinstance(weight._data, PackedTensor)
                        and not weight.qtype.is_floating_point):
                    return None
                n, k = weight.shape[0], weight.numel() // weight.shape[0]
                implicit = _implicit_conv2d_qbits(input, weight, bias, stride, padding, dilation, groups)
                if implicit is not None:
                    return implicit
                return conv2d_as_gemm(input, weight, bias, stride, padding, dilation, groups,
                                      lambda a: torch.ops.quanto.qbits_mm(a, weight._data._data, weight._scale, weight._shift, bias,
                                                                          weight._data.bits, weight._group_size, n, k))

            out = qconv2d(*args, **kwargs)
            if out is not None:
                return out
        if func is torch.equal:
            a, b = args
            return a.equal(b)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)

    @classmethod
    def __torch_dispatch__(cls, op, types, args, kwargs=None):
        kwargs = dict(kwargs or {})
        op = op.overloadpacket
        if op is torch.ops.aten.detach:
            t = args[0]
            names, meta = t.__tensor_flatten__()
            return cls.__tensor_unflatten__({n: op(getattr(t, n)) for n in names}, meta, t.size(), t.stride())
        if op in (torch.ops.aten._to_copy, torch.ops.aten.to):
            t = args[0]
            dtype = kwargs.pop("dtype", t.dtype)
            device = kwargs.pop("device", t.device)
            if dtype is not None and dtype != t.dtype:
                raise ValueError("The dtype of a WeightQBitsTensor cannot be changed")
            scale = op(t._scale, dtype=dtype, device=device, **kwargs)
            data = op(t._data, device=device, **kwargs)
            shift = op(t._shift, device=device, **kwargs)
            return WeightQBitsTensor.create(t._qtype, t._axis, t._group_size, t.size(), t.stride(), data, scale, shift)
        return qfallback(op, *args, **kwargs)


# ================================================================================================
def quantize_weight(t: torch.Tensor, qtype: qtype, axis: int, scale: torch.Tensor, shift: Optional[torch.Tensor] = None,
                    group_size: Optional[int] = None, activation_qtype: Optional[qtype] = None,
                    optimized: Optional[bool] = True):
    """Quantize a weight per-axis (weights/quantization.py:27-73): 8-bit -> WeightQBytesTensor, else WeightQBitsTensor."""
    if axis not in (0, -1):
        raise ValueError("axis parameter must be 0 (first axis) or -1 (last axis)")
    if qtype.bits == 8:
        if shift is not None:
            raise ValueError("shift cannot be specified for 8-bit qtypes")
        if group_size is not None:
            raise ValueError("group_size cannot be specified for 8-bit qtypes.")
        if axis is not None and t.shape[axis] == 1:
            axis = None  # per-axis over a dimension of size one is per-tensor
        return WeightQBytesTensor.quantize(t, qtype, axis, scale, activation_qtype, optimized)
    if shift is None:
        raise ValueError("shift must be specified for qtypes lower than 8-bit")
    return WeightQBitsTensor.quantize(t, qtype, axis, group_size, scale, shift, optimized)
