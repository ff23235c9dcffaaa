"""gloo tests (CPU, world_size 2, 4 and 8) of the column shard: the N > 1 path is correct by construction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, weights, dtype, out):
    import optimum_quanto_amd as Q
    from optimum_quanto_amd.parallel import ColumnParallelQLinear

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)  # every rank builds the same model
        model = torch.nn.Sequential(torch.nn.Linear(256, 128, bias=True).to(dtype))
        Q.quantize(model, weights=weights)
        Q.freeze(model)
        x = torch.randn(5, 256).to(dtype)
        with torch.no_grad():
            ref = model(x)
            sharded = ColumnParallelQLinear.from_qlinear(model[0])
            assert sharded.weight.shape == (128 // world, 256)
            y = sharded(x)
            empty = sharded(x[:0])  # an empty activation keeps the full output width (no collective is needed for it)
        assert y.shape == ref.shape and empty.shape == (0, ref.shape[1])
        # the shard is a pure slice of the same integers and scales (test_shard_layout_is_a_pure_slice): the only freedom left is
        # the CPU GEMM's summation order for a different N.  Distance in units of the output dtype's last place at each element
        yf, rf = y.double(), ref.double()
        ulp = torch.finfo(dtype).eps * torch.maximum(rf.abs(), torch.full_like(rf, torch.finfo(dtype).tiny)) 
        out[rank] = ((yf - rf).abs() / ulp).max().item() if not torch.equal(y, ref) else 0.0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("weights", ["qint4", "qint2", "qint8", "qfloat8"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_column_shard_matches_unsharded(weights, dtype):
    world = 2
    port = _free_port()
    out = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, port, weights, dtype, out), nprocs=world, join=True)
    # fp32: bit-identical or within a few last places of the fp32 sum (a different blocking of the same dot products);
    # bf16: at most one last place of the rounded output
    tol = 4.0 if dtype == torch.float32 else 1.0
    assert len(out) == world and all(e <= tol for e in out.values()), dict(out)


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("weights", ["qint4", "qint2"])
def test_column_shard_at_4_and_8_ranks(weights, world):
    """The sub-byte shard cuts the PACKED rows (vpi output features per byte): N must be a multiple of vpi * G and every rank's
    slice must stay plane-aligned - exercised at the group sizes the driver's 8-GPU node will run."""
    port = _free_port()
    out = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, port, weights, torch.float32, out), nprocs=world, join=True)
    assert len(out) == world and all(e <= 4.0 for e in out.values()), dict(out)


def test_shard_layout_is_a_pure_slice():
    import optimum_quanto_amd as Q
    from optimum_quanto_amd.parallel import shard_qweight, _feature_index

    torch.manual_seed(1)
    w = torch.randn(64, 256)
    scale, shift = Q.MaxOptimizer()(w, qtype=Q.qint4, axis=0, group_size=128)
    qw = Q.quantize_weight(w, Q.qint4, 0, scale, shift, group_size=128)
    full = qw.dequantize()
    for world in (2, 4):
        for rank in range(world):
            local = shard_qweight(qw, rank, world)
            idx = _feature_index(64, 2, rank, world)
            assert torch.equal(local.dequantize(), full[idx])  # same integers, same scales: bit-identical rows
    with pytest.raises(ValueError):
        shard_qweight(qw, 0, 3)
    # 8 ranks, int2 (four features per packed byte): 64 features = 2 per rank and plane, bit-identical rows again ...
    scale2, shift2 = Q.MaxOptimizer()(w, qtype=Q.qint2, axis=0, group_size=128)
    q2 = Q.quantize_weight(w, Q.qint2, 0, scale2, shift2, group_size=128)
    full2 = q2.dequantize()
    for rank in range(8):
        assert torch.equal(shard_qweight(q2, rank, 8).dequantize(), full2[_feature_index(64, 4, rank, 8)])
    # ... and 48 features cannot be cut into 8 plane-aligned slices of an int2 weight
    w48 = torch.randn(48, 256)
    s48, z48 = Q.MaxOptimizer()(w48, qtype=Q.qint2, axis=0, group_size=128)
    with pytest.raises(ValueError):
        shard_qweight(Q.quantize_weight(w48, Q.qint2, 0, s48, z48, group_size=128), 0, 8)
