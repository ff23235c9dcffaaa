"""Quantized type descriptors (API of optimum/quanto/tensor/qtype.py:22-72)."""
from dataclasses import dataclass

import torch

__all__ = ["qtype", "qtypes", "qint2", "qint4", "qint8", "qfloat8", "qfloat8_e4m3fn", "qfloat8_e4m3fnuz", "qfloat8_e5m2",
           "dtype_info", "axis_to_dim"]


@dataclass(eq=True, frozen=True)
class qtype:
    """Mimics a torch dtype for a quantized element type.

    ``dtype`` is the torch storage type, ``bits`` the number of significant bits, ``qmin``/``qmax`` the
    representable range used by the scale optimizers.
    """

    name: str
    is_floating_point: bool
    bits: int
    dtype: torch.dtype
    qmin: float
    qmax: float

    def __str__(self):
        return f"quanto.{self.name}"


def _integer_qtype(bits: int) -> qtype:
    half = 1 << (bits - 1)
    return qtype(f"qint{bits}", False, bits, torch.int8, -half, half - 1)


def _float_qtype(dtype: torch.dtype) -> qtype:
    fi = torch.finfo(dtype)
    return qtype(f"q{fi.dtype}", True, 8, dtype, fi.min, fi.max)


qint2 = _integer_qtype(2)
qint4 = _integer_qtype(4)
qint8 = _integer_qtype(8)
qfloat8_e4m3fn = _float_qtype(torch.float8_e4m3fn)
qfloat8_e4m3fnuz = _float_qtype(torch.float8_e4m3fnuz)
qfloat8_e5m2 = _float_qtype(torch.float8_e5m2)
qfloat8 = qfloat8_e4m3fn  # default float8 flavour

qtypes = {
    "qint2": qint2,
    "qint4": qint4,
    "qint8": qint8,
    "qfloat8_e4m3fn": qfloat8_e4m3fn,
    "qfloat8_e4m3fnuz": qfloat8_e4m3fnuz,
    "qfloat8_e5m2": qfloat8_e5m2,
    "qfloat8": qfloat8,
}


def dtype_info(dtype: torch.dtype):
    """torch.finfo / torch.iinfo depending on the dtype (tensor/core.py:21-23)."""
    return torch.finfo(dtype) if dtype.is_floating_point else torch.iinfo(dtype)


def axis_to_dim(t: torch.Tensor, axis: int):
    """Dimensions to reduce when keeping ``axis`` (tensor/core.py:26-32)."""
    dims = list(range(t.ndim))
    if axis == -1:
        return dims[:-1]
    dims.remove(axis)
    return dims
