"""Column (output-feature) sharding of a frozen QLinear across GPUs - the only multi-GPU piece the path needs.

Output features of a quantized Linear are independent: per-row scales, and the int4 K-groups never cross rows
(SURVEY.md section 8e).  Rank ``r`` of ``G`` keeps ``N/G`` output features and the activations are replicated; one
``all_gather`` of the ``[M, N/G]`` partial outputs (RCCL over xGMI with the ``nccl`` backend, gloo in the CPU tests)
rebuilds ``y``.  The shard is taken on the **packed** tensor, never by re-quantizing or re-packing:

* int8 / fp8 (``WeightQBytesTensor``): rows ``[r*N/G, (r+1)*N/G)`` of ``_data`` and ``_scale``;
* int4 / int2 (``WeightQBitsTensor``, axis 0): with ``vpi = 8 // bits`` planes per byte, packed n-row ``p`` holds
  output features ``p + i*N/vpi``.  Rank ``r`` takes packed n-rows ``[r*P/G, (r+1)*P/G)`` (``P = N/vpi``), i.e. ``vpi``
  disjoint feature ranges; the slice is itself a valid generic-layout tensor of ``N/G`` features, so the same kernels run
  on it, and ``gather_columns`` puts the ranges back in order.

Throughput runs that process independent Linears/batches need no communication at all (bench.py --gpus N).
"""
from typing import Optional

import torch
import torch.distributed as dist

from .tensor import PackedTensor, WeightQBitsTensor, WeightQBytesTensor

__all__ = ["shard_qweight", "shard_bias", "gather_columns", "ColumnParallelQLinear"]


def _planes(qweight) -> int:
    return 8 // qweight.qtype.bits if isinstance(qweight, WeightQBitsTensor) else 1


def shard_qweight(qweight, rank: int, world_size: int):
    """Local shard (same tensor class, generic layout) of a frozen axis-0 weight."""
    n, k = qweight.shape
    if isinstance(qweight, WeightQBytesTensor):
        if qweight.axis != 0 or n % world_size:
            raise ValueError("column sharding needs an axis-0 weight with out_features divisible by the world size")
        rows = slice(rank * n // world_size, (rank + 1) * n // world_size)
        data, scale = qweight._data[rows].contiguous(), qweight._scale[rows].contiguous()
        return WeightQBytesTensor(qweight.qtype, 0, torch.Size([n // world_size, k]), (k, 1), data, scale, qweight.activation_qtype)
    if not isinstance(qweight, WeightQBitsTensor):
        raise TypeError(f"cannot shard {type(qweight).__name__}")
    vpi = _planes(qweight)
    if qweight.axis != 0 or n % (vpi * world_size):
        raise ValueError(f"column sharding of a {qweight.qtype.name} weight needs out_features divisible by {vpi * world_size}")
    gs = qweight._group_size
    groups = k // gs if gs is not None else 1           # grouped rows per output feature
    p_local = n // vpi // world_size                     # packed n-rows owned by this rank
    lo, hi = rank * p_local * groups, (rank + 1) * p_local * groups
    packed = qweight._data._data[lo:hi].contiguous()     # packed matrix rows are grouped rows of plane 0
    plane_rows = (n // vpi) * groups                     # grouped rows per plane
    pick = torch.cat([torch.arange(lo, hi) + i * plane_rows for i in range(vpi)]).to(qweight._scale.device)
    scale, shift = qweight._scale[pick].contiguous(), qweight._shift[pick].contiguous()
    n_local = n // world_size
    inner_size = torch.Size([n_local * groups, gs]) if gs is not None else torch.Size([n_local, k])
    data = PackedTensor(packed, qweight.qtype.bits, inner_size, (inner_size[1], 1))
    return WeightQBitsTensor(qweight.qtype, 0, gs, torch.Size([n_local, k]), (k, 1), data, scale, shift)


def _feature_index(n: int, planes: int, rank: int, world_size: int) -> torch.Tensor:
    """Global output features owned by ``rank``, in the order of its local columns."""
    p_local = n // planes // world_size
    return torch.cat([torch.arange(rank * p_local, (rank + 1) * p_local) + i * (n // planes) for i in range(planes)])


def shard_bias(bias: Optional[torch.Tensor], qweight, rank: int, world_size: int):
    if bias is None:
        return None
    return bias[_feature_index(qweight.shape[0], _planes(qweight), rank, world_size).to(bias.device)].contiguous()


def gather_columns(y_local: torch.Tensor, planes: int, group=None) -> torch.Tensor:
    """all_gather the ``[..., N/G]`` partial outputs and restore the global feature order: ONE collective
    (``all_gather_into_tensor`` into a ``[G * rows, N/G]`` buffer) and ONE copy (the permuted view of that buffer)."""
    world_size = dist.get_world_size(group) if dist.is_initialized() else 1  # a single process needs no process group
    if world_size == 1:
        return y_local
    y_local = y_local.contiguous()
    lead, n_local = y_local.shape[:-1], y_local.shape[-1]
    rows = 1
    for d in lead:
        rows *= d
    if rows == 0 or n_local == 0:  # nothing to exchange (every rank sees the same leading shape): an empty result of the full width
        return y_local.new_empty((*lead, world_size * n_local))
    buf = torch.empty((world_size * rows, n_local), dtype=y_local.dtype, device=y_local.device)  # rank-major concatenation
    dist.all_gather_into_tensor(buf, y_local.reshape(rows, n_local), group=group)
    # rank g's columns are `planes` runs of h features: global feature (i, g, c) <- buf[g, ..., i*h + c]
    h = n_local // planes
    src = buf.reshape(world_size, rows, planes, h).permute(1, 2, 0, 3)
    return src.reshape(*lead, world_size * n_local)


class ColumnParallelQLinear(torch.nn.Module):
    """A frozen ``QLinear`` whose output features are split over the ranks of ``group``; ``forward`` returns the full output."""

    def __init__(self, qweight_local, bias_local, planes: int, group=None):
        super().__init__()
        self.weight = torch.nn.Parameter(qweight_local, requires_grad=False)
        self.bias = None if bias_local is None else torch.nn.Parameter(bias_local, requires_grad=False)
        self.planes = planes
        self.group = group

    @classmethod
    def from_qlinear(cls, qlinear, group=None):
        if not qlinear.frozen:
            raise ValueError("freeze() the module before sharding it")
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        qw = qlinear.weight
        return cls(shard_qweight(qw, rank, world), shard_bias(qlinear.bias, qw, rank, world), _planes(qw), group)

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        y_local = torch.nn.functional.linear(input, self.weight, self.bias)
        return gather_columns(y_local, self.planes, self.group)
