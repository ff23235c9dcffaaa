"""Group-wise views of a weight (semantics of optimum/quanto/tensor/grouped.py:10-51).

For ``axis == 0`` - the only case on the QLinear hot path - grouping is a pure reshape
``(N, K) -> (N*K/g, g)``: grouped row ``n*(K/g) + kg`` holds ``W[n, kg*g : (kg+1)*g]``.  The HIP kernels
index the packed tensor with exactly this formula (csrc/qh_common.h::PackedGeom).
"""
import math
from typing import Sequence

import torch

__all__ = ["group", "ungroup", "grouped_shape"]


def _check_axis(axis):
    if axis not in (0, -1):
        raise ValueError("Axis must be 0 or -1 for group-wise quantization")


def grouped_shape(shape: Sequence[int], axis: int, group_size: int):
    _check_axis(axis)
    n_groups = math.prod(shape) // group_size
    return (n_groups, group_size) if axis == 0 else (group_size, n_groups)


def group(base: torch.Tensor, axis: int, group_size: int) -> torch.Tensor:
    _check_axis(axis)
    axis_dim = base.shape[axis]
    per_feature = base.numel() // axis_dim
    if group_size > per_feature or per_feature % group_size != 0:
        raise ValueError(f"Group size ({group_size}) must be a divisor of ({per_feature})")
    if axis == 0:
        return base.reshape(-1, group_size)
    n_groups = per_feature // group_size
    # (groups, group_size, features) -> (group_size, features, groups) -> (group_size, features*groups)
    return base.reshape(n_groups, group_size, axis_dim).permute(1, 2, 0).reshape(group_size, axis_dim * n_groups)


def ungroup(grouped: torch.Tensor, axis: int, orig_shape) -> torch.Tensor:
    if grouped.shape == orig_shape:
        return grouped
    if axis == 0:
        return grouped.reshape(orig_shape)
    group_size = grouped.shape[0]
    axis_dim = orig_shape[axis]
    n_groups = grouped.numel() // axis_dim // group_size
    return grouped.reshape(group_size, axis_dim, n_groups).permute(2, 0, 1).reshape(orig_shape)
