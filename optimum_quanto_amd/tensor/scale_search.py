"""Scale (and shift) selection for weight quantization.

Mirrors optimum/quanto/tensor/optimizers/: ``Optimizer`` -> ``SymmetricOptimizer`` -> ``AbsmaxOptimizer`` for
8-bit types, ``Optimizer`` -> ``AffineOptimizer`` -> ``MaxOptimizer`` for sub-byte types.  They run once at
``freeze()`` time in plain torch (on whichever device holds the float weights) and define the integers the
hot path consumes, so their arithmetic follows the reference exactly (absmax_optimizer.py:26-36,
max_optimizer.py:26-37, affine_optimizer.py:27-64).
"""
from typing import Optional, Tuple

import torch

from .dtypes import qtype
from .grouping import group

__all__ = ["Optimizer", "SymmetricOptimizer", "AbsmaxOptimizer", "AffineOptimizer", "MaxOptimizer"]


class Optimizer:
    def __call__(self, base: torch.Tensor, *args, **kwargs):
        raise NotImplementedError


def _other_dims(t: torch.Tensor, axis: int):
    return list(range(1, t.ndim)) if axis == 0 else list(range(0, t.ndim - 1))


class SymmetricOptimizer(Optimizer):
    def __call__(self, base: torch.Tensor, qtype: qtype, axis: Optional[int] = None) -> torch.Tensor:
        if axis not in (None, 0, -1):
            raise ValueError("axis parameter must be None, 0 (first axis) or -1 (last axis)")
        if axis is not None and base.shape[axis] == 1:
            axis = None
        scale = self.optimize(base, qtype, axis)
        assert scale.dtype == base.dtype
        return scale

    def optimize(self, base, qtype, axis):
        raise NotImplementedError


class AbsmaxOptimizer(SymmetricOptimizer):
    """scale = max|w| / qmax, per tensor or per index of ``axis``."""

    def optimize(self, base, qtype, axis=None):
        mag = torch.abs(base)
        peak = torch.max(mag) if axis is None else torch.amax(mag, dim=_other_dims(base, axis), keepdim=True)
        return peak / qtype.qmax


class AffineOptimizer(Optimizer):
    def __call__(self, base: torch.Tensor, qtype: qtype, axis: int, group_size: Optional[int] = None,
                 zeropoint: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """Return (scale, shift).  With ``zeropoint`` the shift is rounded to an integer in [0, 2^bits)."""
        if axis not in (0, -1):
            raise ValueError("axis parameter must be 0 (first axis) or -1 (last axis)")
        if group_size is not None:
            base = group(base, axis, group_size)
        if axis is not None and base.shape[axis] == 1:
            axis = None
        scale, shift = self.optimize(base, qtype, axis)
        assert scale.dtype == base.dtype and shift.dtype == base.dtype
        if zeropoint:
            shift = torch.clamp(torch.round(shift / scale), 0, 2**qtype.bits - 1).to(torch.uint8)
        return scale, shift

    def optimize(self, base, qtype, axis):
        raise NotImplementedError


class MaxOptimizer(AffineOptimizer):
    """scale = (max - min) / (2^bits - 1), shift = -min, per (grouped) row."""

    def optimize(self, base, qtype, axis):
        dims = _other_dims(base, axis)
        lo = torch.amin(base, dim=dims, keepdim=True)
        hi = torch.amax(base, dim=dims, keepdim=True)
        levels = 2**qtype.bits - 1
        return (hi - lo) / levels, -lo
