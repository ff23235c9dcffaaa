"""Quantized tensor containers: ``QTensor`` and its two storage families.

API mirror of optimum/quanto/tensor/qtensor.py:21-96, qbytes.py:23-54 and qbits.py:27-74.  The containers
are wrapper subclasses that own the integer data and its scale (and shift); every torch op that has no
quantized implementation falls back to ``dequantize()`` (``qfallback``).

MI355X specifics: for axis-0 2-D ``QBitsTensor`` weights ``dequantize()`` is a single fused kernel
(``quanto::dequantize_qbits`` -> csrc/unpack.hip) that reproduces the reference's rounding sequence bit for
bit, instead of unpack + mul + sub + reshape passes over a 2x intermediate.
"""
import torch
from torch.autograd import Function
from torch.utils import _pytree as pytree

from .grouping import ungroup
from .packing import PackedTensor

__all__ = ["QTensor", "QBytesTensor", "QBitsTensor", "qfallback"]


def qfallback(callable, *args, **kwargs):
    """Call ``callable`` after dequantizing every QTensor argument."""
    args, kwargs = pytree.tree_map_only(QTensor, lambda q: q.dequantize(), (args, kwargs or {}))
    return callable(*args, **kwargs)


def _deepcopy_flattened(t, memo):
    import copy

    if id(t) in memo:
        return memo[id(t)]
    names, meta = t.__tensor_flatten__()
    inner = {n: copy.deepcopy(getattr(t, n), memo) for n in names}
    out = type(t).__tensor_unflatten__(inner, meta, None, None)
    out.requires_grad_(t.requires_grad)
    memo[id(t)] = out
    return out


class QTensor(torch.Tensor):
    def __init__(self, qtype, axis):
        self._qtype = qtype
        self._axis = axis

    @property
    def qtype(self):
        return self._qtype

    @property
    def axis(self):
        return self._axis

    def dequantize(self):
        raise NotImplementedError

    def numpy(self):
        return self.dequantize().cpu().numpy()

    def save_to_state_dict(self, destination, prefix, keep_vars):
        """Flatten recursively into plain tensors: ``<prefix>_data``, ``<prefix>_scale``, ... (qtensor.py:40-53)."""

        def walk(t, pfx):
            names, _ = t.__tensor_flatten__()
            for name in names:
                inner = getattr(t, name)
                if type(inner) is torch.Tensor:
                    destination[pfx + name] = inner if keep_vars else inner.detach()
                else:
                    walk(inner, pfx + name + ".")

        walk(self, prefix)

    def __deepcopy__(self, memo):
        """copy.deepcopy(model) of a frozen model: rebuild the same class around deep copies of the inner tensors (torch's default
        for wrapper subclasses wants an aten.clone that returns the subclass, which weights do not have - nor do the reference's)."""
        return _deepcopy_flattened(self, memo)

    def equal(self, other) -> bool:
        if type(self) is not type(other):
            return False
        names, meta = self.__tensor_flatten__()
        _, other_meta = other.__tensor_flatten__()
        if any(other_meta[k] != v for k, v in meta.items()):
            return False
        for name in names:
            a, b = getattr(self, name), getattr(other, name)
            if a.device.type == "cpu" and a.dtype in (torch.float8_e4m3fn, torch.float8_e5m2, torch.float8_e4m3fnuz):
                # torch.equal has no CPU kernel for float8
                if a.dtype != b.dtype or not torch.equal(a.to(torch.float32), b.to(torch.float32)):
                    return False
            elif not torch.equal(a, b):
                return False
        return True


class _DequantizeBytes(Function):
    """scale * data, in the scale dtype (qbytes.py:23-36).  Straight-through gradient."""

    @staticmethod
    def forward(ctx, t):
        if t._data.is_cuda and t._scale.numel() == 1 and t._data.dim() >= max(t._scale.dim(), 1):
            # r6: a per-tensor scale on the device (a quantized activation): cast + multiply in one pass, bit-identical (csrc/quantize.hip)
            from ..library.hip import quanto_hip

            out = quanto_hip.lib.dequantize_symmetric(t._data, t._scale)
            if out is not None:
                return out
        data = t._data.to(t._scale.dtype) if t.qtype.is_floating_point else t._data
        return t._scale * data

    @staticmethod
    def backward(ctx, grad):
        return grad


class QBytesTensor(QTensor):
    """8-bit data (int8 or float8) with a scale."""

    def __init__(self, qtype, axis, size, stride, data, scale, requires_grad=False):
        super().__init__(qtype, axis)
        self._data = data
        self._scale = scale

    def __repr__(self):
        return f"{type(self).__name__}({self._data}, scale={self._scale}, dtype={self.dtype})"

    def dequantize(self):
        return _DequantizeBytes.apply(self)


class _DequantizeBits(Function):
    """Unpack, remove the zero-point / shift, scale, restore the shape (qbits.py:27-49)."""

    @staticmethod
    def forward(ctx, t):
        data, scale, shift = t._data, t._scale, t._shift
        if isinstance(data, PackedTensor) and t.axis == 0 and t.ndim == 2 and not t.qtype.is_floating_point:
            return torch.ops.quanto.dequantize_qbits(
                data._data, scale, shift, data.bits, t._group_size, t.shape[0], t.shape[1])
        values = data.unpack() if isinstance(data, PackedTensor) else data
        if not shift.dtype.is_floating_point:
            values = values.to(torch.int8) - shift.to(torch.int8)
        if t.qtype.is_floating_point:
            values = values.to(scale.dtype)
        out = scale * values
        if shift.dtype.is_floating_point:
            out -= shift
        if t.axis is None:
            return out
        return ungroup(out, axis=t.axis, orig_shape=t.shape)

    @staticmethod
    def backward(ctx, grad):
        return grad


class QBitsTensor(QTensor):
    """Sub-byte data (packed) with a per-group scale and shift."""

    def __init__(self, qtype, axis, group_size, size, stride, data, scale, shift, requires_grad=False):
        super().__init__(qtype, axis)
        self._data = data
        self._scale = scale
        self._shift = shift
        self._group_size = group_size

    def __repr__(self):
        return f"{type(self).__name__}({self._data}, scale={self._scale}, shift={self._shift}, dtype={self.dtype})"

    def dequantize(self):
        return _DequantizeBits.apply(self)
