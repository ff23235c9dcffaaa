"""Sub-byte packing: ``pack_weights`` and the ``PackedTensor`` wrapper subclass.

Bit layout (defined by optimum/quanto/tensor/packed.py:24-69 and consumed unchanged by the HIP kernels):
with ``vpi = 8 // bits`` values per byte and ``row_dim = ceil(rows / vpi)``, plane ``i`` (rows
``[i*row_dim, (i+1)*row_dim)`` of the unpacked tensor) is stored in bits ``[bits*i, bits*(i+1))`` of the packed
byte at the same (row - i*row_dim, col).  ``unpack`` is ``torch.ops.quanto.unpack`` + a trim of the padding rows
(packed.py:101-104), which on ROCm tensors runs csrc/unpack.hip.
"""
import ast

import torch
from torch.utils import _pytree as pytree

__all__ = ["PackedTensor", "pack_weights"]


def pack_weights(intweights: torch.Tensor, bits: int) -> torch.Tensor:
    """Pack 2-bit or 4-bit values stored one per byte into a ``torch.uint8`` tensor."""
    if bits not in (2, 4):
        raise ValueError("bits must be 2 or 4")
    if intweights.is_cuda and intweights.ndim >= 1 and intweights.numel() > 0:
        from ..library.hip import quanto_hip  # one pass on the device (csrc/quantize.hip::pack_kernel); raises if the library is missing

        return quanto_hip.lib.pack(intweights, bits)
    vpi = 8 // bits
    rows = intweights.shape[0]
    row_dim = -(-rows // vpi)
    values = intweights.to(torch.uint8)
    packed = torch.zeros((row_dim,) + tuple(intweights.shape[1:]), dtype=torch.uint8, device=intweights.device)
    for plane in range(vpi):
        lo = plane * row_dim
        hi = min(lo + row_dim, rows)
        if hi <= lo:
            break
        packed[: hi - lo] |= values[lo:hi] << (bits * plane)
    return packed


class PackedTensor(torch.Tensor):
    """A ``torch.uint8`` tensor of ``size`` whose storage holds 2 or 4 values per byte."""

    @staticmethod
    def __new__(cls, data, bits, size, stride, requires_grad=False):
        assert data.dtype == torch.uint8
        assert requires_grad is False  # integer data never carries a gradient
        return torch.Tensor._make_wrapper_subclass(
            cls, size, strides=stride, dtype=torch.uint8, device=data.device, requires_grad=False)

    def __init__(self, data, bits, size, stride, requires_grad=False):
        self._bits = bits
        self._data = data

    def __repr__(self):
        return f"PackedTensor({self._data}, bits={self._bits}, public_dtype={self.dtype})"

    @classmethod
    def pack(cls, t: torch.Tensor, bits: int = 4):
        assert bits in (2, 4)
        assert t.dtype in (torch.uint8, torch.int8)
        return cls(pack_weights(t, bits), bits, t.size(), t.stride())

    def unpack(self) -> torch.Tensor:
        full = torch.ops.quanto.unpack(self._data, self._bits)
        return full[: self.shape[0]]  # drop the rows added when the first dim is not a multiple of 8 // bits

    @property
    def bits(self):
        return self._bits

    @property
    def dtype(self):
        return torch.uint8

    def __deepcopy__(self, memo):
        import copy

        return PackedTensor(copy.deepcopy(self._data, memo), self._bits, self.size(), self.stride())

    # -- serialization (flatten protocol; meta values are AST-evaluable strings) --------------------
    def __tensor_flatten__(self):
        meta = {"bits": str(self._bits), "size": str(list(self.size())), "stride": str(self.stride())}
        return ["_data"], meta

    @staticmethod
    def __tensor_unflatten__(inner_tensors, meta, outer_size, outer_stride):
        assert len(inner_tensors) == 1 and len(meta) == 3
        return PackedTensor(inner_tensors["_data"], ast.literal_eval(meta["bits"]), ast.literal_eval(meta["size"]),
                            ast.literal_eval(meta["stride"]))

    @staticmethod
    def load_from_state_dict(state_dict, prefix, bits, size, stride, missing_keys):
        key = prefix + "_data"
        if key not in state_dict:
            missing_keys.append(key)
            return None
        meta = {"bits": str(bits), "size": str(list(size)), "stride": str(stride)}
        return PackedTensor.__tensor_unflatten__({"_data": state_dict.pop(key)}, meta, None, None)

    __torch_function__ = torch._C._disabled_torch_function_impl

    @classmethod
    def __torch_dispatch__(cls, op, types, args, kwargs=None):
        kwargs = kwargs or {}
        packet = op.overloadpacket
        if packet is torch.ops.aten.detach:
            t = args[0]
            return PackedTensor(op(t._data), t._bits, t.size(), t.stride())
        if packet in (torch.ops.aten._to_copy, torch.ops.aten.to):
            t = args[0]
            if kwargs.get("dtype", torch.uint8) != torch.uint8:
                raise ValueError(f"PackedTensor are torch.uint8 only and cannot be moved to {kwargs['dtype']}.")
            return PackedTensor(op(t._data, **kwargs), t._bits, t.size(), t.stride())
        # anything else operates on the unpacked values
        args, kwargs = pytree.tree_map_only(PackedTensor, lambda p: p.unpack(), (args, kwargs))
        return op(*args, **kwargs)

    def numpy(self):
        return self.unpack().cpu().numpy()
