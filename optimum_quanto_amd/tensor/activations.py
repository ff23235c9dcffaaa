"""Quantized activations: ``ActivationQBytesTensor`` and ``quantize_activation``.

API mirror of optimum/quanto/tensor/activations/{qbytes.py:28-92, quantization.py:24-39, qbytes_ops.py:31-284}.
Activations are quantized per-tensor (scalar scale) to int8 or float8.  They matter to the hot path through one
call only: ``F.linear(qinput, qweight)`` -> ``quanto::qbytes_mm(input._data, weight._data, input._scale * weight._scale)``
(tensor/weights/qbytes.py:72-73), which on an MI355X is the int8 x int8 / fp8 x fp8 MFMA kernel (csrc/qmm_native8.hip);
the quantization itself is the one-pass ``quanto::quantize_symmetric`` kernel (csrc/quantize.hip).

The aten-level behaviour follows the reference's table (shape ops stay quantized and share the scale, scalar mul/div
fold into the scale, softmax re-quantizes with the known 1/max scale, everything else dequantizes first).
"""
import ast
import numbers

import torch
from torch.autograd import Function

from .base import QBytesTensor, QTensor, qfallback
from .dtypes import axis_to_dim, dtype_info, qint8, qtype, qtypes

__all__ = ["ActivationQBytesTensor", "quantize_activation", "absmax_scale"]

aten = torch.ops.aten


class _QuantizeActivation(Function):
    """Straight-through quantizer: forward builds the tensor subclass, backward passes the gradient unchanged."""

    @staticmethod
    def forward(ctx, base, qtype, scale):
        if qtype.bits != 8:
            raise ValueError("QBytesTensor can only be of 8-bit qtype")
        data = torch.ops.quanto.quantize_symmetric(base, dtype=qtype.dtype, axis=None, scale=scale)
        return ActivationQBytesTensor(qtype, base.size(), base.stride(), data, scale)

    @staticmethod
    def backward(ctx, grad):
        return grad, None, None


def _is_scalar(v) -> bool:
    return isinstance(v, numbers.Number) or (type(v) is torch.Tensor and v.ndim == 0)


def _like(t, data, scale=None, size=None, stride=None):
    """A new quantized activation sharing ``t``'s qtype; shape/strides default to those of ``data``."""
    return ActivationQBytesTensor(t.qtype, data.size() if size is None else size, data.stride() if stride is None else stride,
                                  data, t._scale if scale is None else scale)


def _same_quantization(a, b) -> bool:
    return (isinstance(a, ActivationQBytesTensor) and isinstance(b, ActivationQBytesTensor) and a.qtype == b.qtype
            and torch.equal(a._scale, b._scale))


# ---- aten handlers: each receives the overload packet and the original arguments -----------------------------------
def _h_to(op, t, dtype=None, **kw):
    # the 8-bit payload keeps its dtype; only the scale follows the requested float dtype
    return _like(t, op(t._data, dtype=t._data.dtype, **kw), op(t._scale, dtype=dtype, **kw), t.size(), t.stride())


def _h_detach(op, t):
    return _like(t, op(t._data), op(t._scale), t.size(), t.stride())


def _h_clone(op, t, memory_format=torch.preserve_format):
    data = op(t._data, memory_format=memory_format)
    return _like(t, data, op(t._scale, memory_format=memory_format), t.size(), data.stride())


def _h_copy_(op, dst, src):
    assert dst.qtype == src.qtype
    dst._data = op(dst._data, src._data)
    dst._scale = op(dst._scale, src._scale)
    return dst


def _h_shape_op(op, t, *args, **kw):
    # expand / permute / select / slice / unsqueeze / transpose / view: transparent for a scalar scale
    if t.axis is not None:
        return op(t.dequantize(), *args, **kw)
    return _like(t, op(t._data, *args, **kw))


def _h_t(op, t):
    rows, cols = t.size()
    return _like(t, op(t._data), None, torch.Size([cols, rows]), t.stride()[::-1])


def _h_cat_or_stack(op, tensors, dim=0):
    if len(tensors) == 2 and _same_quantization(*tensors) and all(t.axis is None for t in tensors):
        a, b = tensors
        if not (op is aten.cat and a.qtype.is_floating_point):  # cat has no float8 kernel
            return _like(a, op([a._data, b._data], dim))
    return qfallback(op, tensors, dim)


def _h_split(op, t, *args, **kw):
    if t.axis is not None:
        return qfallback(op, t, *args, **kw)
    return [_like(t, piece) for piece in op(t._data, *args, **kw)]


def _h_lt(op, a, b):
    if _same_quantization(a, b):
        return op(a._data, b._data)
    return qfallback(op, a, b)


def _h_is_same_size(op, a, b):
    return op(a._data if isinstance(a, ActivationQBytesTensor) else a, b._data if isinstance(b, ActivationQBytesTensor) else b)


def _h_mul(op, a, b):
    if _is_scalar(a):
        return _like(b, b._data, a * b._scale, b.size(), b.stride())
    if _is_scalar(b):
        return _like(a, a._data, b * a._scale, a.size(), a.stride())
    return qfallback(op, a, b)


def _h_div(op, a, b):
    if not _is_scalar(b):
        return op(a.dequantize(), b)
    return _like(a, a._data, op(a._scale, b), a.size(), a.stride())


def _h_int_only(op, t, *args, **kw):
    # neg / relu act on the integer payload; float8 has no such kernels
    if t.qtype.is_floating_point:
        return qfallback(op, t, *args, **kw)
    return _like(t, op(t._data, *args, **kw), None, t.size(), t.stride())


def _h_softmax(op, t, dim, half_to_float):
    out = op(t.dequantize(), dim, half_to_float)
    # a softmax output lies in [0, 1]: the optimal per-tensor scale is known
    scale = torch.tensor(1 / dtype_info(t.qtype.dtype).max, dtype=t._scale.dtype).to(t.device)
    return quantize_activation(out, qtype=t.qtype, scale=scale)


def _h_where(op, cond, t, other):
    if isinstance(cond, QTensor) or isinstance(other, QTensor):
        raise NotImplementedError
    out = op(cond, t.dequantize(), other)
    return quantize_activation(out, qtype=t.qtype, scale=t._scale) if t.axis is None else out


def _h_bmm(op, a, b):
    if not isinstance(a, ActivationQBytesTensor):
        return op(a, b.dequantize())
    if not isinstance(b, QTensor) or a.axis is not None:
        return op(a.dequantize(), b)
    if a.qtype != qint8 or b.qtype != qint8 or (b.axis is not None and b.size() != b._data.size()):
        return qfallback(op, a, b)
    out = op(a._data.to(torch.float32), b._data.to(torch.float32))
    return (out * (a._scale * b._scale).to(torch.float32)).to(a._scale.dtype)


_HANDLERS = {
    aten._to_copy: _h_to, aten.to: _h_to, aten.detach: _h_detach, aten.clone: _h_clone, aten.copy_: _h_copy_,
    aten.expand: _h_shape_op, aten.permute: _h_shape_op, aten.select: _h_shape_op, aten.slice: _h_shape_op,
    aten.unsqueeze: _h_shape_op, aten.transpose: _h_shape_op, aten.view: _h_shape_op, aten._unsafe_view: _h_shape_op,
    aten.t: _h_t, aten.cat: _h_cat_or_stack, aten.stack: _h_cat_or_stack, aten.split: _h_split, aten.lt: _h_lt,
    aten.is_same_size: _h_is_same_size, aten.mul: _h_mul, aten.div: _h_div, aten.neg: _h_int_only, aten.relu: _h_int_only,
    aten._softmax: _h_softmax, aten.where: _h_where, aten.bmm: _h_bmm,
}


class ActivationQBytesTensor(QBytesTensor):
    @staticmethod
    def __new__(cls, qtype, size, stride, data, scale, requires_grad=False):
        assert data.device == scale.device
        return torch.Tensor._make_wrapper_subclass(cls, size, strides=stride, dtype=scale.dtype, device=data.device,
                                                   requires_grad=requires_grad)

    def __init__(self, qtype, size, stride, data, scale, requires_grad=False):
        super().__init__(qtype, None, size, stride, data, scale, requires_grad)

    @classmethod
    def quantize(cls, base: torch.Tensor, qtype: qtype, scale: torch.Tensor) -> torch.Tensor:
        return _QuantizeActivation.apply(base, qtype, scale)

    def __tensor_flatten__(self):
        meta = {"qtype": self._qtype.name, "size": str(list(self.size())), "stride": str(list(self.stride()))}
        return ["_data", "_scale"], meta

    @staticmethod
    def __tensor_unflatten__(inner_tensors, meta, outer_size, outer_stride):
        assert len(inner_tensors) == 2 and len(meta) == 3
        return ActivationQBytesTensor(qtypes[meta["qtype"]], ast.literal_eval(meta["size"]), ast.literal_eval(meta["stride"]),
                                      inner_tensors["_data"], inner_tensors["_scale"])

    @classmethod
    def __torch_dispatch__(cls, op, types, args, kwargs=None):
        packet = op.overloadpacket
        handler = _HANDLERS.get(packet)
        if handler is not None:
            return handler(packet, *args, **(kwargs or {}))
        return qfallback(packet, *args, **(kwargs or {}))


def quantize_activation(t: torch.Tensor, qtype: qtype, scale: torch.Tensor):
    """Quantize an activation per-tensor with the scalar ``scale`` (tensor/activations/quantization.py:24-39)."""
    if scale.numel() != 1:
        raise ValueError("Parameter scale must be a scalar because activations can only be quantized per-tensor")
    return ActivationQBytesTensor.quantize(t, qtype, scale)


def absmax_scale(base: torch.Tensor, qtype: qtype = qint8, axis=None) -> torch.Tensor:
    """The scale ``quantize_activation`` is called with: max(|base|) / qmax, per tensor (axis=None) or per slice along ``axis``
    (optimum/quanto/calibrate.py:38-64).  The calibration pass that maintains these scales over sample batches
    (``Calibration``) is host code outside this backend's scope: in plug-in mode the unmodified reference
    supplies it; a stand-alone user sets ``module.input_scale`` / ``module.output_scale`` directly."""
    mag = torch.abs(base)
    peak = torch.max(mag) if axis is None else torch.amax(mag, dim=axis_to_dim(base, axis), keepdim=True)
    return peak / dtype_info(qtype.dtype).max
