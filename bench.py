#!/usr/bin/env python3
"""Headline benchmark: QLinear GEMM throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]
                    [--workload cfg2|cfg3|northstar|cfg4|w8a8|fp8a8|int4_prefill|int8_decode|int4_decode32]

A *step* is one pass of the hot path (one ``quanto::qbytes_mm`` / ``quanto::qbits_mm`` call through the C ABI) over
one batch of synthetic input already resident in HBM.  The default workload is BASELINE.json ``configs[1]``:
bf16 x int8, per-channel scale, (M,K,N) = (4096,4096,4096).  Decode workloads (cfg3 / northstar) rotate over
> 512 MB of distinct weight buffers so every launch streams its weights from HBM, not from the 256 MB Infinity Cache.

With N > 1 (launched by ``python -m torch.distributed.run``) every rank runs the same workload on its own GPU: the
path is embarrassingly parallel per Linear, no data-path collective, weak scaling; ``value`` is the whole-job
aggregate (sum over ranks of work / max-over-ranks time).

Before the timed K steps the same steps are replayed, untimed, for ``--ramp-ms`` (default 300 ms): an idle MI355X takes
tens of milliseconds of sustained load to reach its steady clock / power state (the number is in ``config.clock_ramp_ms``;
``--ramp-ms 0`` gives the cold-start figure).

One JSON line is printed by rank 0, carrying ``roofline`` (dominant kernel, measured with device events around the
timed region) and ``cpu_baseline`` (the numpy oracle timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec)
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16/fp16 MFMA peak
MFMA_PEAK_8BIT_TOPS = 5000.0  # dense int8 / fp8 MFMA peak (the dtype's peak: fp8 x fp8 is priced against it although the
                              # non-scaled 16x16x32 fp8 MFMA it uses today runs at the bf16 rate)

WORKLOADS = {
    # name: (kind, M, K, N, description)
    "cfg2": ("qbytes_i8", 4096, 4096, 4096, "bf16 x int8 qbytes_mm, per-channel scale, (M,K,N)=(4096,4096,4096)"),
    "cfg3": ("qbits_i4", 1, 4096, 11008, "bf16 x int4 qbits_mm, group_size=128 scale+shift, (M,K,N)=(1,4096,11008)"),
    "northstar": ("qbits_i4", 1, 4096, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, (M,K,N)=(1,4096,4096)"),
    "cfg4": ("qbytes_f8", 512, 8192, 8192, "bf16 x fp8-e4m3fn qbytes_mm, per-channel scale, (M,K,N)=(512,8192,8192)"),
    # SURVEY.md 8f rank 1 (quantized activations) and the int4 prefill shape of the same layer size as cfg2
    "w8a8": ("qbytes_i8i8", 4096, 4096, 4096, "int8 x int8 qbytes_mm (quantized activations), int32 accumulate, (M,K,N)=(4096,4096,4096)"),
    "fp8a8": ("qbytes_f8f8", 4096, 4096, 4096, "fp8-e4m3fn x fp8-e4m3fn qbytes_mm (quantized activations), (M,K,N)=(4096,4096,4096)"),
    "int4_prefill": ("qbits_i4", 4096, 4096, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, prefill (M,K,N)=(4096,4096,4096)"),
    "int8_decode": ("qbytes_i8", 1, 4096, 4096, "bf16 x int8 qbytes_mm, per-channel scale, decode (M,K,N)=(1,4096,4096)"),
    "int4_decode32": ("qbits_i4", 32, 4096, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(32,4096,4096)"),
}
ARITH_DTYPE = {"qbytes_i8": "bf16", "qbytes_f8": "bf16", "qbits_i4": "bf16", "qbytes_i8i8": "int8", "qbytes_f8f8": "fp8"}


def algorithmic_work(kind, M, K, N):
    """FLOPs and bytes per call, SURVEY.md section 8d (same formulas as the reference's bench/kernels/benchmark_w4a16.py:44-71)."""
    flops = 2.0 * M * N * K
    if kind == "qbits_i4":
        G = K // 128
        nbytes = N * K // 2 + 2 * (N * G * 2) + M * K * 2 + M * N * 2
    elif kind in ("qbytes_i8i8", "qbytes_f8f8"):
        nbytes = N * K + N * 2 + M * K + M * N * 2
    else:
        nbytes = N * K + N * 2 + M * K * 2 + M * N * 2
    return flops, float(nbytes)


def build_inputs(kind, M, K, N, device, n_weights, seed):
    """Synthetic inputs of the workload's shape, generated as SURVEY.md 8(d) prescribes: activations ``randn``, weights
    ``randn * 0.02`` pushed through the reference quantizer's arithmetic (per-row absmax int8 / fp8, ``library/quantize.py:26-56``
    with ``absmax_optimizer.py:26-36``; per-group max-min int4 with float shift, ``library/quantize.py:64-78`` with
    ``max_optimizer.py:26-37``, packed as ``tensor/packed.py:24-69``).  Values do matter to a compute-bound GEMM on this
    part: the matrix pipes draw data-dependent power (4096^3 bf16 x int8: 101 us per launch on random operands, 79 us on
    constant ones; the vendor dense GEMM 90 vs 67 us), so the bench never uses constant or zero operands."""
    g = torch.Generator(device=device).manual_seed(seed)

    def randn(*shape):
        return torch.randn(shape, generator=g, device=device, dtype=torch.float32)

    def absmax_quantize(w, qmax, dtype, axis_scale=True):
        amax = w.abs().amax(dim=1, keepdim=True) if axis_scale else w.abs().amax()
        scale = (amax / qmax).to(torch.bfloat16)
        q = w / scale.float()
        q = torch.round(q).clamp(-qmax, qmax).to(dtype) if dtype == torch.int8 else q.clamp(-qmax, qmax).to(dtype)
        return q, scale

    x = randn(M, K).to(torch.bfloat16)
    x_scale = None
    if kind == "qbytes_i8i8":  # per-tensor absmax activations (tensor/activations/quantization.py:24-31)
        x, x_scale = absmax_quantize(x.float(), 127, torch.int8, axis_scale=False)
    elif kind == "qbytes_f8f8":
        x, x_scale = absmax_quantize(x.float(), 448, torch.float8_e4m3fn, axis_scale=False)
    sets = []
    for _ in range(n_weights):
        w = (randn(N, K) * 0.02).to(torch.bfloat16).float()
        if kind == "qbits_i4":
            wg = w.reshape(N * K // 128, 128)
            lo, hi = wg.amin(dim=1, keepdim=True), wg.amax(dim=1, keepdim=True)
            scale = ((hi - lo) / 15).to(torch.bfloat16)
            shift = (-lo).to(torch.bfloat16)
            q = torch.round((wg + shift.float()) / scale.float()).clamp(0, 15).to(torch.uint8)
            half = q.shape[0] // 2
            packed = (q[:half] | (q[half:] << 4)).contiguous()
            sets.append((packed, scale, shift))
        elif kind in ("qbytes_i8", "qbytes_i8i8"):
            q, scale = absmax_quantize(w, 127, torch.int8)
            if x_scale is not None:
                scale = (scale.float() * x_scale.float()).to(torch.bfloat16)
            sets.append((q.contiguous(), scale))
        else:
            q, scale = absmax_quantize(w, 448, torch.float8_e4m3fn)
            if x_scale is not None:
                scale = (scale.float() * x_scale.float()).to(torch.bfloat16)
            sets.append((q.contiguous(), scale))
        del w
    return x.contiguous(), sets


def make_step(kind, x, sets, K, N):
    from optimum_quanto_amd.library.hip import quanto_hip

    lib = quanto_hip.lib
    state = {"i": 0}
    if kind == "qbits_i4":
        def step():
            packed, scale, shift = sets[state["i"] % len(sets)]
            state["i"] += 1
            return torch.ops.quanto.qbits_mm(x, packed, scale, shift, None, 4, 128, N, K)
    else:
        def step():
            w, scale = sets[state["i"] % len(sets)]
            state["i"] += 1
            return torch.ops.quanto.qbytes_mm(x, w, scale)
    return step, lib


def cpu_baseline(kind, M, K, N, budget_s=12.0):
    """Time the numpy oracle (a port of the reference's CPU path) on a bounded sample of the same workload."""
    from oracle import quanto_oracle as O

    rng = np.random.default_rng(0)
    cores = os.cpu_count() or 1
    if kind == "qbits_i4":
        # one full call of the generic reference path (unpack -> dequantize -> matmul): tensor/qbits.py:27-49 + function.py:41-47
        Ns = N
        packed = rng.integers(0, 256, size=(Ns * K // 256, 128), dtype=np.uint8)
        scale = O.round_to(rng.random((Ns * K // 128, 1)).astype(np.float32) * 0.01 + 0.005, "bf16")
        shift = O.round_to(rng.random((Ns * K // 128, 1)).astype(np.float32) * 0.05 + 0.05, "bf16")
        x = O.round_to(rng.standard_normal((M, K)).astype(np.float32), "bf16")
        fn = lambda: O.qbits_mm_ref(x, packed, 4, scale, shift, 128, Ns, K, "bf16")  # noqa: E731
        Ms, sample = M, f"full call (M,K,N)=({M},{K},{Ns}), generic unpack+dequantize+matmul"
    elif kind == "qbytes_i8i8":
        Ms = min(M, 512)
        a8 = rng.integers(-127, 128, size=(Ms, K), dtype=np.int8)
        b8 = rng.integers(-127, 128, size=(N, K), dtype=np.int8)
        scale = O.round_to(rng.random((N, 1)).astype(np.float32) * 1e-5 + 5e-6, "bf16")
        fn = lambda: O.qbytes_int_mm_ref(a8, b8, scale, "bf16")  # noqa: E731
        Ns, sample = N, f"first {Ms} of {M} activation rows, full weight (K,N)=({K},{N}): integer matmul + rescale per call"
    else:
        Ms = min(M, 512)
        kindf = "e4m3fn" if kind in ("qbytes_f8", "qbytes_f8f8") else None
        data = rng.integers(0, 256, size=(N, K), dtype=np.uint8) if kindf else rng.integers(-127, 128, size=(N, K), dtype=np.int8)
        if kindf:
            data[(data & 0x7F) == 0x7F] = 0  # avoid NaN codes
        scale = O.round_to(rng.random((N, 1)).astype(np.float32) * 1e-3 + 5e-4, "bf16")
        x = O.round_to(rng.standard_normal((Ms, K)).astype(np.float32), "bf16")
        if kind == "qbytes_f8f8":
            x = O.fp8_decode(O.fp8_encode(x, "e4m3fn"), "e4m3fn")  # activations on the fp8 grid
        fn = lambda: O.qbytes_mm_ref(x, data, scale, "bf16", kindf)  # noqa: E731
        Ns, sample = N, f"first {Ms} of {M} activation rows, full weight (K,N)=({K},{N}): dequantize + matmul per call"
    fn()  # warm-up
    times, t_start = [], time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 20):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    flops, nbytes = algorithmic_work(kind, Ms, K, Ns)
    return t, flops, nbytes, cores, sample, len(times)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="issue the timed steps one by one from Python instead of replaying a hipGraph")
    ap.add_argument("--ramp-ms", type=float, default=300.0,
                    help="untimed: keep the device busy with the same steps for this long before the timed region (clock ramp)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a ROCm device"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import optimum_quanto_amd  # noqa: F401  registers the quanto:: ops; raises if the HIP library cannot be loaded later

    kind, M, K, N, desc = WORKLOADS[args.workload]
    flops, nbytes = algorithmic_work(kind, M, K, N)
    weight_bytes = N * K // 2 if kind == "qbits_i4" else N * K
    # decode workloads: rotate over > 512 MB of weights so each launch reads HBM (SURVEY.md 8d "cache hygiene")
    n_weights = max(1, -(-(512 << 20) // weight_bytes)) if M <= 64 else 1
    x, sets = build_inputs(kind, M, K, N, device, n_weights, seed=1234 + rank)
    step, lib = make_step(kind, x, sets, K, N)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    kernel_name = lib.last_kernel()
    # The K timed steps are captured once into a hipGraph and replayed: a decode-shaped call lasts a few microseconds,
    # far less than the host needs to issue it through Python, so an eager loop would time the host, not the GPU.
    graph = None
    if not args.eager:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(args.steps):
                    step()
        torch.cuda.current_stream().wait_stream(side)
        graph.replay()  # untimed: uploads the executable graph
        torch.cuda.synchronize()
    # Untimed clock ramp.  An idle MI355X needs tens of milliseconds of sustained load to reach its steady power state: the
    # first 6 ms window of 4096^3 launches after process start averages 121 us per launch, the following ones 110, 105, 103,
    # 102 and from ~100 ms on 100 us (scripts/microbench_qbytes.py, same shape first / last in a process).  W warm-up
    # steps of a 0.1 ms kernel cannot cover that, so the same steps are replayed for --ramp-ms before the timed K steps.
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        if graph is not None:
            graph.replay()
        else:
            for _ in range(args.steps):
                step()
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    if graph is not None:
        graph.replay()
    else:
        for _ in range(args.steps):
            step()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        tt = torch.tensor([elapsed, dev_ms], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, dev_ms = float(tt[0]), float(tt[1])

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        launch_ms = dev_ms / args.steps  # device-side average per launch (events on the launch stream)
        compute_bound = M > 64
        if compute_bound:
            value = flops * world / (elapsed / args.steps) / 1e12
            metric, unit = "QLinear GEMM TFLOP/s (bf16 x int8 qbytes_mm)" if kind == "qbytes_i8" else "QLinear GEMM TFLOP/s", "TFLOP/s"
            achieved = flops / (launch_ms * 1e-3) / 1e12
            peak = MFMA_PEAK_8BIT_TOPS if kind in ("qbytes_i8i8", "qbytes_f8f8") else MFMA_PEAK_TFLOPS
            roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": None}
        else:
            value = nbytes * world / (elapsed / args.steps) / 1e9
            metric, unit = f"QLinear GEMM GB/s ({'bf16 x int4 qbits_mm' if kind == 'qbits_i4' else 'bf16 x int8 qbytes_mm'}, decode)", "GB/s"
            achieved = nbytes / (launch_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None}
        roof["kernel"] = kernel_name
        roof["launch_us"] = round(launch_ms * 1e3, 3)
        roof["algorithmic_bytes"] = nbytes
        roof["algorithmic_flops"] = flops
        pmc = os.path.join(ROOT, "profiles", f"pmc_{args.workload}.json")
        if os.path.exists(pmc):  # HBM bytes per launch from rocprofv3 --pmc passes (see profiles/README.md)
            roof["traffic"] = json.load(open(pmc)).get("hbm_bytes_per_launch")
        out = {
            "metric": metric, "value": round(value, 3), "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": ARITH_DTYPE[kind], "data": "synthetic",
            "config": {"workload": desc, "M": M, "K": K, "N": N, "weight_buffers_rotated": n_weights, "launch": "eager" if args.eager else "hipGraph replay of the K steps", "clock_ramp_ms": args.ramp_ms,
                       "parallelism": f"replicas x{world} (no data-path collective)"},
            "tflops": round(flops * world / (elapsed / args.steps) / 1e12, 3),
            "gbps": round(nbytes * world / (elapsed / args.steps) / 1e9, 1),
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:  # the host baseline is reported at N = 1 only
            t, cflops, cbytes, cores, sample, n = cpu_baseline(kind, M, K, N)
            cval = cflops / t / 1e12 if compute_bound else cbytes / t / 1e9
            out["cpu_baseline"] = {"value": round(cval, 5), "unit": unit, "cores": cores, "kind": "port", "sample": sample,
                                   "seconds_per_call": round(t, 4), "calls_timed": n}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
