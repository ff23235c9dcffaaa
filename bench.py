#!/usr/bin/env python3
"""Headline benchmark: QLinear GEMM throughput on MI355X (BASELINE.json metric: TFLOP/s and GB/s vs the roofline).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-sub] [--no-cpu-baseline] [--shard]

A *step* is one pass of the hot path (one ``quanto::qbytes_mm`` / ``quanto::qbits_mm`` call through the C ABI) over one
batch of synthetic input already resident in HBM.  The default run prints ONE JSON line whose ``value`` is BASELINE.json
``configs[1]`` (bf16 x int8, per-channel scale, (M,K,N) = (4096,4096,4096), TFLOP/s) and whose ``sub_results`` carry the
int4 half of the headline metric, measured in the same process with the same discipline: the north-star decode shape
(1,4096,4096), ``configs[2]`` (1,4096,11008), and the same decode GEMV over the fused q/k/v and gate/up projections of a
Llama-3-8B layer (``quanto::qbits_mm_multi``: one launch for Linears that share their input).  Every result has its own
``roofline`` (device events around the timed region, on the stream the kernels run on) and ``cpu_baseline``.

Method (same formulas as the reference's bench/kernels/benchmark_w4a16.py:44-71: TFLOP/s = 2MNK / t, GB/s = bytes / t):
  * the K timed steps are captured once in a hipGraph and replayed (a decode call is shorter than its Python issue time);
  * decode workloads rotate over > 512 MB of distinct weight buffers so every launch streams from HBM, not from the
    256 MB Infinity Cache (SURVEY.md 8d "cache hygiene");
  * before the timed replay the same graph is replayed, untimed, for ``--ramp-ms`` (default 300 ms): an idle MI355X takes
    ~100 ms of sustained load to reach its steady clock / power state (``config.clock_ramp_ms``; ``--ramp-ms 0`` = cold);
  * ``cpu_baseline`` (rank 0, N = 1 only) times the reference's own CPU path - the ATen kernels QLinear.forward reaches on
    CPU tensors, issued by ``oracle/reference_cpu_path.py`` exactly as the reference issues them (pinned ``torch.equal``
    against the imported reference in tests/test_reference_integration.py) - on the FULL shape and the SAME tensors the
    GPU side just multiplied, on the GPU box's host cores: median + IQR + thread count.

With N > 1 every rank runs the same workload on its own GPU: the path is embarrassingly parallel per Linear, no data-path
collective, weak scaling; ``value`` aggregates over ranks / max-over-ranks time.  The ranks come from the launcher
(``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``: RANK / LOCAL_RANK / WORLD_SIZE in the environment) or,
when ``--gpus N`` is given with no launcher around it, from bench.py itself (it re-executes under torch.distributed.run on
127.0.0.1).  At N > 1 the default run appends one more sub-result, ``cfg4_sharded``: ``configs[3]`` (fp8 weights, (512,8192,8192))
column-sharded over the ranks with ``ColumnParallelQLinear``, compute-only and compute + ``all_gather_into_tensor`` (RCCL over xGMI),
``"scaling": "strong"`` - SURVEY.md 8(e); ``--shard`` makes that the headline line instead.

Round 4 additions to the default line (all measured in the same process / on the same box as the headline):
  * ``roofline.frac`` is computed from the SAME host-bracketed time as ``value`` (``ms_per_step``); the device-event time of the same
    replay is ``event_us`` (r3 divided by the event time: 0.5544 next to a value that said 0.548);
  * ``kernel_us`` / ``kernel_us_min``: kernel-only duration from a ``rocprofv3 --kernel-trace`` pass of this same file (a child
    process, ``--trace-child``; r5: one steady-state hipGraph replay), next to the launch-inclusive ``us_per_step``.  The profiler stamps a
    dispatch from the start of its set-up, which an unprofiled replay overlaps with the tail of the previous kernel (~0.5 us): for kernels
    of a few us BOTH figures can therefore exceed ``us_per_step`` (r5 driver line, north-star: kernel_us_min 4.52 vs us_per_step 4.23; r6: no claim
    is made for us-scale rows, and ``trace_floor_us`` = the shortest marker dispatch of the same trace - a 3-byte kernel - shows what the profiler
    stamps for a kernel that does nothing); ``traffic`` = fabric bytes per step from two more child
    passes (``--pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE``, corrected as MI355X_MICROARCH.md prescribes); when rocprofv3 is not usable the
    committed ``profiles/pmc_*.json`` figure is reported with ``traffic_stale: true``;
  * ``ref_rocm_us``: the op sequence the UNMODIFIED reference issues for the same call on a ROCm device (elementwise dequantize +
    hipBLASLt: library/qbytes_mm.py:25-33,73-88; library/unpack.py:21-54 + tensor/qbits.py:27-49 + tensor/function.py:41-47), timed
    with the same graph / ramp discipline on the same tensors;
  * ``layer_decode`` records: the seven int4 QLinears of one Llama-3-8B layer as the FOUR launches the product issues at decode time
    (q/k/v in one, o, gate/up in one, down), > 512 MB of layer weights rotated, B = 1 and B = 32: sum of us, algorithmic bytes, layer-level
    HBM fraction, kernel-only sum, and - labelled as such - the same sequence over a working set that stays resident in the 256 MiB
    Infinity Cache (two layers' weights, 218 MB: what a perfect weight prefetcher running under the model's non-library kernels could
    deliver; a side-stream touch kernel inside the hipGraph was measured in r4 and is 5 x SLOWER: profiles/r04_side_stream_prefetch_negative.txt);
  * ``cfg5``: BASELINE configs[4] end to end - Llama-3-8B random-init bf16, weights=qint4 with lm_head excluded, the reference's
    method (bench/generation/metrics/latency.py:24-105: ``generate`` 512 prompt + 512 new tokens, greedy, eos disabled), tokens/s at
    batch 1 and batch 32 (``--no-cfg5`` skips it).

The printed line is kept under ~7.5 KB (the driver keeps an 8 KB stdout tail): sub-results are compact records (name, shape, us,
kernel, fraction, algorithmic bytes / flops, counter traffic, CPU seconds), the prose that explains the CPU paths is printed once
(``cpu_paths``); ``--verbose`` prints every sub-result in full on stderr.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec)
MFMA_PEAK_TFLOPS = 2500.0     # dense bf16/fp16 MFMA peak
MFMA_PEAK_8BIT_TOPS = 5000.0  # dense int8 / fp8 MFMA peak

LLAMA3_QKV = (4096, 1024, 1024)   # q / k / v out_features of a Llama-3-8B layer (hidden 4096, 8 kv heads of 128)
LLAMA3_GATE_UP = (14336, 14336)

WORKLOADS = {
    # name: (kind, M, K, N or tuple of N, description)
    "cfg2": ("qbytes_i8", 4096, 4096, 4096, "bf16 x int8 qbytes_mm, per-channel scale, (M,K,N)=(4096,4096,4096)"),
    "cfg3": ("qbits_i4", 1, 4096, 11008, "bf16 x int4 qbits_mm, group_size=128 scale+shift, (M,K,N)=(1,4096,11008)"),
    "northstar": ("qbits_i4", 1, 4096, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, (M,K,N)=(1,4096,4096)"),
    "cfg4": ("qbytes_f8", 512, 8192, 8192, "bf16 x fp8-e4m3fn qbytes_mm, per-channel scale, (M,K,N)=(512,8192,8192)"),
    "qkv_fused": ("qbits_i4_multi", 1, 4096, LLAMA3_QKV, "bf16 x int4 qbits_mm_multi, Llama-3-8B q/k/v in one launch, M=1, K=4096, N=4096+1024+1024"),
    "gateup_fused": ("qbits_i4_multi", 1, 4096, LLAMA3_GATE_UP, "bf16 x int4 qbits_mm_multi, Llama-3-8B gate/up in one launch, M=1, K=4096, N=2x14336"),
    "qkv_fused8": ("qbits_i4_multi", 8, 4096, LLAMA3_QKV, "bf16 x int4 qbits_mm_multi, Llama-3-8B q/k/v in one launch, batched decode M=8"),
    "qkv_fused32": ("qbits_i4_multi", 32, 4096, LLAMA3_QKV, "bf16 x int4 qbits_mm_multi, Llama-3-8B q/k/v in one launch, batched decode M=32"),
    "gateup_fused8": ("qbits_i4_multi", 8, 4096, LLAMA3_GATE_UP, "bf16 x int4 qbits_mm_multi, Llama-3-8B gate/up in one launch, batched decode M=8"),
    "gateup_fused32": ("qbits_i4_multi", 32, 4096, LLAMA3_GATE_UP, "bf16 x int4 qbits_mm_multi, Llama-3-8B gate/up in one launch, batched decode M=32"),
    # SURVEY.md 8f rank 1 (quantized activations) and the int4 prefill / batched-decode shapes of the same layer size
    "w8a8": ("qbytes_i8i8", 4096, 4096, 4096, "int8 x int8 qbytes_mm (quantized activations), int32 accumulate, (M,K,N)=(4096,4096,4096)"),
    "fp8a8": ("qbytes_f8f8", 4096, 4096, 4096, "fp8-e4m3fn x fp8-e4m3fn qbytes_mm (quantized activations), (M,K,N)=(4096,4096,4096)"),
    "cfg4_fp8a8": ("qbytes_f8f8", 512, 8192, 8192, "fp8-e4m3fn x fp8-e4m3fn qbytes_mm on the native fp8 MFMA (BASELINE configs[3] with quantized activations), (M,K,N)=(512,8192,8192)"),
    "cfg4_w8a8": ("qbytes_i8i8", 512, 8192, 8192, "int8 x int8 qbytes_mm (quantized activations), (M,K,N)=(512,8192,8192)"),
    # r6: a Llama-3 down-projection at 512 tokens with quantized activations - 128 output tiles for K = 14336: the K split of qmm_native8.hip
    "w8a8_down512": ("qbytes_i8i8", 512, 14336, 4096, "int8 x int8 qbytes_mm (quantized activations), Llama-3-8B down_proj at 512 tokens, (M,K,N)=(512,14336,4096)"),
    # r6: int4 weights x quantized activations on the 8-bit matrix instructions (quanto::qbits_mm_a8, csrc/qbits_a8_fused.hip)
    "w4a8": ("qbits_i4_a8i", 4096, 4096, 4096, "int8 activations x int4 qbits_mm_a8, group_size=128 scale+shift, (M,K,N)=(4096,4096,4096)"),
    "w4a8_512": ("qbits_i4_a8i", 512, 4096, 4096, "int8 activations x int4 qbits_mm_a8, group_size=128 scale+shift, (M,K,N)=(512,4096,4096)"),
    "w4afp8": ("qbits_i4_a8f", 4096, 4096, 4096, "fp8-e4m3fn activations x int4 qbits_mm_a8, group_size=128 scale+shift, (M,K,N)=(4096,4096,4096)"),
    "w4afp8_512": ("qbits_i4_a8f", 512, 4096, 4096, "fp8-e4m3fn activations x int4 qbits_mm_a8, group_size=128 scale+shift, (M,K,N)=(512,4096,4096)"),
    "int4_prefill": ("qbits_i4", 4096, 4096, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, prefill (M,K,N)=(4096,4096,4096)"),
    "int4_prefill512": ("qbits_i4", 512, 4096, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, (M,K,N)=(512,4096,4096)"),
    "int8_8k": ("qbytes_i8", 8192, 8192, 8192, "bf16 x int8 qbytes_mm, per-channel scale, (M,K,N)=(8192,8192,8192)"),
    "fp8_4k": ("qbytes_f8", 4096, 4096, 4096, "bf16 x fp8-e4m3fn qbytes_mm, per-channel scale, (M,K,N)=(4096,4096,4096)"),
    "int8_decode": ("qbytes_i8", 1, 4096, 4096, "bf16 x int8 qbytes_mm, per-channel scale, decode (M,K,N)=(1,4096,4096)"),
    "int4_decode32": ("qbits_i4", 32, 4096, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(32,4096,4096)"),
    "int4_decode32_down": ("qbits_i4", 32, 14336, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(32,14336,4096)"),
    "int4_decode8": ("qbits_i4", 8, 4096, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(8,4096,4096)"),
    "int4_decode16": ("qbits_i4", 16, 4096, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(16,4096,4096)"),
    "int4_decode8_up": ("qbits_i4", 8, 4096, 14336, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(8,4096,14336)"),
    "int4_decode8_down": ("qbits_i4", 8, 14336, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(8,14336,4096)"),
    "int4_decode16_up": ("qbits_i4", 16, 4096, 14336, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(16,4096,14336)"),
    "int4_decode16_down": ("qbits_i4", 16, 14336, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(16,14336,4096)"),
    "int8_qkv_fused": ("qbytes_i8_multi", 1, 4096, LLAMA3_QKV, "bf16 x int8 qbytes_mm_multi, Llama-3-8B q/k/v in one launch, M=1"),
    "int8_gateup_fused": ("qbytes_i8_multi", 1, 4096, LLAMA3_GATE_UP, "bf16 x int8 qbytes_mm_multi, Llama-3-8B gate/up in one launch, M=1"),
    "int8_qkv_fused32": ("qbytes_i8_multi", 32, 4096, LLAMA3_QKV, "bf16 x int8 qbytes_mm_multi, Llama-3-8B q/k/v in one launch, batched decode M=32"),
    "int8_gateup_fused32": ("qbytes_i8_multi", 32, 4096, LLAMA3_GATE_UP, "bf16 x int8 qbytes_mm_multi, Llama-3-8B gate/up in one launch, batched decode M=32"),
    "int4_decode8_kv": ("qbits_i4", 8, 4096, 1024, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(8,4096,1024)"),
    "int4_decode8_13b": ("qbits_i4", 8, 5120, 5120, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(8,5120,5120)"),
    "int4_decode8_70b": ("qbits_i4", 8, 8192, 8192, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(8,8192,8192)"),
    "int4_decode64": ("qbits_i4", 64, 4096, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(64,4096,4096)"),
    "int8_decode32": ("qbytes_i8", 32, 4096, 4096, "bf16 x int8 qbytes_mm, per-channel scale, batched decode (M,K,N)=(32,4096,4096)"),
    "int4_decode1_down": ("qbits_i4", 1, 14336, 4096, "bf16 x int4 qbits_mm, group_size=128 scale+shift, decode (M,K,N)=(1,14336,4096) (Llama-3-8B down_proj)"),
    "int4_decode32_up": ("qbits_i4", 32, 4096, 14336, "bf16 x int4 qbits_mm, group_size=128 scale+shift, batched decode (M,K,N)=(32,4096,14336)"),
}
DEFAULT_SUB = ["northstar", "cfg3", "cfg4", "cfg4_fp8a8", "w8a8", "fp8a8", "w8a8_down512", "w4a8", "w4a8_512", "int4_prefill", "int4_prefill512", "gateup_fused", "int4_decode32",
               "qkv_fused32", "layer_decode_b1", "layer_decode_b32"]  # (q/k/v at M = 1 and the int8 gate/up launch of r2-r4 are inside layer_decode_b1 / --sub)
# printed ONCE per JSON line (r2's line repeated this prose in every sub-result, grew past the driver's 8 KB stdout tail and lost
# its first two sub-results): what cpu_baseline.kind == "reference" and each cpu_baseline.path code stand for
CPU_BASELINE_NOTE = {
    "reference": "the ATen kernels the reference's CPU QLinear path executes, issued in its order by oracle/reference_cpu_path.py "
                 "(torch.equal to the imported reference in the build container); same tensors as the GPU leg, full shape, all host threads",
    "int8pack": "torch._weight_int8pack_mm, library/qbytes_mm.py:91-105",
    "int_mm": "torch._int_mm + fp32 rescale, library/qbytes_mm.py:36-50",
    "fp8_generic": "cast fp8->bf16, scale the weight, matmul, library/qbytes_mm.py:25-33",
    "int4_generic": "generic WeightQBitsTensor: unpack+dequantize+matmul per call, tensor/qbits.py:27-49, tensor/function.py:41-47",
    "tinygemm": "TinyGemmWeightQBitsTensor (create()'s CPU choice for bf16 scales): torch._weight_int4pack_mm_for_cpu, "
                "tensor/weights/tinygemm/qbits.py:51-58; lossy shift repack at load time",
}
ARITH_DTYPE = {"qbits_i4_a8i": "int8", "qbits_i4_a8f": "fp8", "qbytes_i8": "bf16", "qbytes_f8": "bf16", "qbits_i4": "bf16", "qbits_i4_multi": "bf16", "qbytes_i8_multi": "bf16", "qbytes_i8i8": "int8", "qbytes_f8f8": "fp8"}


def algorithmic_work(kind, M, K, N):
    """FLOPs and bytes per call, SURVEY.md section 8d (formulas of the reference's bench/kernels/benchmark_w4a16.py:44-71).
    A fused multi-Linear launch reads x once and is charged the sum of its Linears' weight / scale / output bytes."""
    Ns = N if isinstance(N, tuple) else (N,)
    Nt = sum(Ns)
    flops = 2.0 * M * Nt * K
    if kind in ("qbits_i4_a8i", "qbits_i4_a8f"):
        G = K // 128
        nbytes = Nt * K // 2 + 2 * (Nt * G * 2) + M * K + M * Nt * 2
    elif kind in ("qbits_i4", "qbits_i4_multi"):
        G = K // 128
        nbytes = Nt * K // 2 + 2 * (Nt * G * 2) + M * K * 2 + M * Nt * 2
    elif kind in ("qbytes_i8i8", "qbytes_f8f8"):
        nbytes = Nt * K + Nt * 2 + M * K + M * Nt * 2
    else:
        nbytes = Nt * K + Nt * 2 + M * K * 2 + M * Nt * 2
    return flops, float(nbytes)


def quantize_int4(w):
    """Per-group (128) max-min int4 with float shift (library/quantize.py:64-78 with max_optimizer.py:26-37), packed as
    tensor/packed.py:24-69.  w: [N, K] fp32 holding bf16 values."""
    N, K = w.shape
    wg = w.reshape(N * K // 128, 128)
    lo, hi = wg.amin(dim=1, keepdim=True), wg.amax(dim=1, keepdim=True)
    scale = ((hi - lo) / 15).to(torch.bfloat16)
    shift = (-lo).to(torch.bfloat16)
    q = torch.round((wg + shift.float()) / scale.float()).clamp(0, 15).to(torch.uint8)
    half = q.shape[0] // 2
    return (q[:half] | (q[half:] << 4)).contiguous(), scale, shift


def build_inputs(kind, M, K, N, device, n_weights, seed):
    """Synthetic inputs as SURVEY.md 8(d) prescribes: activations ``randn``, weights ``randn * 0.02`` pushed through the
    reference quantizer's arithmetic (per-row absmax int8 / fp8: library/quantize.py:26-56 + absmax_optimizer.py:26-36).
    Values matter to a compute-bound GEMM on this part: the matrix pipes draw data-dependent power (4096^3 bf16 x int8:
    101 us per launch on random operands, 79 us on constant ones), so the bench never uses constant or zero operands."""
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
    elif kind == "qbits_i4_a8i":
        x, x_scale = absmax_quantize(x.float(), 127, torch.int8, axis_scale=False)
        x = (x.contiguous(), x_scale.reshape(1))
    elif kind == "qbits_i4_a8f":
        x, x_scale = absmax_quantize(x.float(), 448, torch.float8_e4m3fn, axis_scale=False)
        x = (x.contiguous(), x_scale.reshape(1))
    sets = []
    for _ in range(n_weights):
        if kind == "qbits_i4_multi":
            sets.append([quantize_int4((randn(n, K) * 0.02).to(torch.bfloat16).float()) for n in N])
            continue
        if kind == "qbytes_i8_multi":
            sets.append([absmax_quantize((randn(n, K) * 0.02).to(torch.bfloat16).float(), 127, torch.int8) for n in N])
            continue
        w = (randn(N, K) * 0.02).to(torch.bfloat16).float()
        if kind in ("qbits_i4", "qbits_i4_a8i", "qbits_i4_a8f"):
            sets.append(quantize_int4(w))
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
    return (x if isinstance(x, tuple) else x.contiguous()), sets


def make_step(kind, x, sets, K, N):
    state = {"i": 0}
    if kind in ("qbits_i4_a8i", "qbits_i4_a8f"):
        xq, xs = x

        def step():
            packed, scale, shift = sets[state["i"] % len(sets)]
            state["i"] += 1
            return torch.ops.quanto.qbits_mm_a8(xq, xs, packed, scale, shift, None, 4, 128, N, K)
    elif kind == "qbits_i4":
        def step():
            packed, scale, shift = sets[state["i"] % len(sets)]
            state["i"] += 1
            return torch.ops.quanto.qbits_mm(x, packed, scale, shift, None, 4, 128, N, K)
    elif kind == "qbits_i4_multi":
        Ns = list(N)
        packs = [([p for p, _, _ in s], [sc for _, sc, _ in s], [sh for _, _, sh in s]) for s in sets]
        none = [None] * len(Ns)

        def step():
            p, sc, sh = packs[state["i"] % len(packs)]
            state["i"] += 1
            return torch.ops.quanto.qbits_mm_multi(x, p, sc, sh, none, 4, 128, Ns, K)
    elif kind == "qbytes_i8_multi":
        packs = [([q.contiguous() for q, _ in s], [sc for _, sc in s]) for s in sets]
        none = [None] * len(N)

        def step():
            ws, sc = packs[state["i"] % len(packs)]
            state["i"] += 1
            return torch.ops.quanto.qbytes_mm_multi(x, ws, sc, none)
    else:
        def step():
            w, scale = sets[state["i"] % len(sets)]
            state["i"] += 1
            return torch.ops.quanto.qbytes_mm(x, w, scale)
    return step


REF_ROCM_FOR = ("cfg2", "cfg3", "northstar", "cfg4", "int4_prefill", "int4_prefill512", "w8a8", "fp8a8", "cfg4_fp8a8", "cfg4_w8a8", "w8a8_down512", "w4a8", "w4a8_512", "w4afp8", "w4afp8_512")


def make_ref_rocm_step(kind, x, wset, K, N):
    """What the UNMODIFIED reference executes for this call on a ROCm device (it has no fused kernel there: only ``unpack`` is
    native, library/extensions/hip/__init__.py:25-36): dequantize the whole weight elementwise, then a dense hipBLASLt matmul.
    8-bit: library/qbytes_mm.py:25-33 (reached from :73-88 for float activations).  4-bit: quanto::unpack (library/unpack.py:21-54:
    mask, shift, cat - the reference's one-pass HIP unpack kernel yields the same tensor), ``scale * data``, ``-= shift``
    (tensor/qbits.py:41-45), ungroup = reshape for axis 0 (tensor/grouped.py:39-44), matmul (tensor/function.py:44)."""
    if kind in ("qbits_i4_a8i", "qbits_i4_a8f"):
        # quantized activation x int4 weight in the reference: dequantize the activation (tensor/weights/qbits.py:262-287 -> qfallback), then the
        # int4 sequence below
        packed, scale, shift = wset
        xq, xs = x

        def step():
            xd = xq.to(scale.dtype) * xs
            data = torch.cat([packed & 0x0F, (packed & 0xF0) >> 4]).to(torch.uint8)
            dqt = scale * data
            dqt -= shift
            return torch.matmul(xd, dqt.reshape(N, K).t())
    elif kind == "qbits_i4":
        packed, scale, shift = wset

        def step():
            data = torch.cat([packed & 0x0F, (packed & 0xF0) >> 4]).to(torch.uint8)
            dqt = scale * data
            dqt -= shift
            return torch.matmul(x, dqt.reshape(N, K).t())
    elif kind == "qbytes_i8i8":  # library/qbytes_mm.py:36-50, reached from :73-87 for int8 activations: hipBLASLt int8 GEMM + fp32 rescale
        w, scale = wset

        def step():
            out = torch._int_mm(x, w.t())
            return (out.to(torch.float32) * scale.t()).to(scale.dtype)
    else:
        w, scale = wset

        def step():
            a = x.to(scale.dtype)
            ww = w.to(scale.dtype) if w.dtype.is_floating_point else w
            return torch.matmul(a, (scale * ww).t())
    return step


def timed_replay(step, steps, args, dist, device, warmup=None):
    """W untimed warm-up steps, the K steps captured in ONE hipGraph (or issued eagerly with --eager), the clock ramp, then the
    timed region: barrier + synchronize on both sides, host clock around it and a device-event pair on the launch stream inside it;
    max over ranks.  Returns (elapsed_s, event_ms)."""
    for _ in range(args.warmup if warmup is None else warmup):
        step()
    torch.cuda.synchronize()
    graph = None
    if not args.eager:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(steps):
                    step()
        torch.cuda.current_stream().wait_stream(side)
        graph.replay()  # untimed: uploads the executable graph
        torch.cuda.synchronize()

    def run():
        if graph is not None:
            graph.replay()
        else:
            for _ in range(steps):
                step()

    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        run()
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run()
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
    del graph
    return elapsed, dev_ms


def cpu_baseline(kind, M, K, N, x, wset, budget_s):
    """The reference's CPU path (ATen kernels, reference order) on the full shape and the very tensors the GPU just used."""
    from oracle import reference_cpu_path as R

    xc = x.cpu()
    out = {"kind": "reference", "cores": torch.get_num_threads()}  # what "reference" means: CPU_BASELINE_NOTE, printed once per line
    if kind in ("qbits_i4", "qbits_i4_multi"):
        parts = wset if kind == "qbits_i4_multi" else [wset]
        Ns = list(N) if isinstance(N, tuple) else [N]
        cpu = [(p.cpu(), sc.cpu(), sh.cpu()) for p, sc, sh in parts]
        generic = lambda: [R.qbits_linear_generic(xc, p, sc, sh, 4, 128, n, K) for (p, sc, sh), n in zip(cpu, Ns)]  # noqa: E731
        tg = [R.tinygemm_pack(p, sc, sh, 128, n, K) for (p, sc, sh), n in zip(cpu, Ns)]  # load-time repack, not timed
        tiny = lambda: [R.tinygemm_linear(xc, d, 128, ss, n) for (d, ss), n in zip(tg, Ns)]  # noqa: E731
        t_gen = R.time_call(generic, budget_s)
        t_tiny = R.time_call(tiny, min(budget_s, 3.0), min_calls=20, max_calls=200)
        out["path"] = "int4_generic"
        t = t_gen
        out["tinygemm"] = {"seconds_per_call": round(t_tiny["median_s"], 6), "iqr_s": round(t_tiny["iqr_s"], 6), "calls": t_tiny["calls"]}
    else:
        parts = [(w.cpu(), sc.cpu()) for w, sc in (wset if kind == "qbytes_i8_multi" else [wset])]
        fn = lambda: [R.qbytes_mm_cpu(xc, wc, sc) for wc, sc in parts]  # noqa: E731
        t = R.time_call(fn, budget_s)
        out["path"] = {"qbytes_i8": "int8pack", "qbytes_i8_multi": "int8pack", "qbytes_i8i8": "int_mm"}.get(kind, "fp8_generic")
    flops, nbytes = algorithmic_work(kind, M, K, N)
    compute_bound = M > 64
    out["value"] = round(flops / t["median_s"] / 1e12, 5) if compute_bound else round(nbytes / t["median_s"] / 1e9, 4)
    out["unit"] = "TFLOP/s" if compute_bound else "GB/s"
    out["sample"] = f"full call, {t['calls']} timed calls after 3 warm-up"
    out["seconds_per_call"] = round(t["median_s"], 6)
    out["iqr_s"] = round(t["iqr_s"], 6)
    out["calls_timed"] = t["calls"]
    if "tinygemm" in out:
        out["tinygemm"]["value"] = round(nbytes / out["tinygemm"]["seconds_per_call"] / 1e9, 3)
        out["tinygemm"]["unit"] = "GB/s"
    return out


def run_workload(name, args, device, rank, world, dist, steps, with_cpu):
    from optimum_quanto_amd.library.hip import quanto_hip

    lib = quanto_hip.lib
    kind, M, K, N, desc = WORKLOADS[name]
    flops, nbytes = algorithmic_work(kind, M, K, N)
    Nt = sum(N) if isinstance(N, tuple) else N
    weight_bytes = Nt * K // 2 if kind.startswith("qbits") else Nt * K
    # decode workloads: rotate over > 512 MB of weights so each launch reads HBM (SURVEY.md 8d "cache hygiene")
    n_weights = max(1, -(-(512 << 20) // weight_bytes)) if M <= 64 else 1
    x, sets = build_inputs(kind, M, K, N, device, n_weights, seed=1234 + rank)
    step = make_step(kind, x, sets, K, N)

    step()
    torch.cuda.synchronize()
    kernel_name = lib.last_kernel()
    elapsed, dev_ms = timed_replay(step, steps, args, dist, device)
    if rank != 0:
        return None

    ms_per_step = elapsed * 1e3 / steps
    launch_ms = ms_per_step     # the roofline fraction uses the SAME time as `value` (host clock around barrier + synchronize) ...
    event_ms = dev_ms / steps   # ... the device-event pair inside that region is reported next to it (event_us)
    compute_bound = M > 64
    if compute_bound:
        value = flops * world / (elapsed / steps) / 1e12
        what = {"qbytes_i8": "bf16 x int8 qbytes_mm", "qbytes_f8": "bf16 x fp8 qbytes_mm", "qbits_i4": "bf16 x int4 qbits_mm",
                "qbytes_i8i8": "int8 x int8 qbytes_mm", "qbytes_f8f8": "fp8 x fp8 qbytes_mm", "qbits_i4_a8i": "int8 x int4 qbits_mm_a8",
                "qbits_i4_a8f": "fp8 x int4 qbits_mm_a8"}[kind]
        metric, unit = f"QLinear GEMM TFLOP/s ({what})", "TFLOP/s"
        achieved = flops / (launch_ms * 1e-3) / 1e12
        peak = MFMA_PEAK_8BIT_TOPS if kind in ("qbytes_i8i8", "qbytes_f8f8", "qbits_i4_a8i", "qbits_i4_a8f") else MFMA_PEAK_TFLOPS
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None}
    else:
        value = nbytes * world / (elapsed / steps) / 1e9
        metric, unit = f"QLinear GEMM GB/s ({'bf16 x int4 qbits_mm' if kind.startswith('qbits') else 'bf16 x int8 qbytes_mm'}, decode)", "GB/s"
        achieved = nbytes / (launch_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None}
    roof["kernel"] = kernel_name
    roof["launch_us"] = round(launch_ms * 1e3, 3)
    roof["event_us"] = round(event_ms * 1e3, 3)
    roof["kernel_us"] = roof["kernel_us_min"] = None  # kernel-only duration: filled from the rocprofv3 kernel-trace child pass
    roof["algorithmic_bytes"] = nbytes
    roof["algorithmic_flops"] = flops
    out = {
        "metric": metric, "value": round(value, 3), "unit": unit, "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ARITH_DTYPE[kind], "data": "synthetic",
        "config": {"workload": desc, "name": name, "M": M, "K": K, "N": list(N) if isinstance(N, tuple) else N,
                   "weight_buffers_rotated": n_weights, "launch": "eager" if args.eager else "hipGraph replay of the K steps",
                   "clock_ramp_ms": args.ramp_ms, "parallelism": f"replicas x{world} (no data-path collective)"},
        "tflops": round(flops * world / (elapsed / steps) / 1e12, 3),
        "gbps": round(nbytes * world / (elapsed / steps) / 1e9, 1),
        "roofline": roof,
    }
    if name in REF_ROCM_FOR and not args.no_ref_rocm:
        try:
            ref_step = make_ref_rocm_step(kind, x, sets[0], K, N)
            with torch.no_grad():
                r_elapsed, _ = timed_replay(ref_step, 10 if M > 64 else 20, args, None, device, warmup=2)
            out["ref_rocm_us"] = round(r_elapsed * 1e6 / (10 if M > 64 else 20), 2)
        except Exception as e:  # an ATen op the ROCm build lacks (torch._int_mm): say so instead of dropping the record
            out["ref_rocm_us"] = None
            out["ref_rocm_error"] = repr(e)[:120]
    if with_cpu and not isinstance(x, tuple):  # (quantized activation x int4: the reference has no CPU path of its own for it - it dequantizes the activation and runs the int4 path timed above)
        out["cpu_baseline"] = cpu_baseline(kind, M, K, N, x, sets[0], args.cpu_budget)
    del sets
    torch.cuda.empty_cache()
    return out


def run_sharded(args, device, rank, world, dist):
    """configs[3], ONE Linear column-sharded over the ranks (parallel.py): rank r multiplies x by its N/G output features,
    one all_gather_into_tensor rebuilds y.  Times compute only and compute + all_gather, both as max over ranks."""
    import optimum_quanto_amd as Q
    from optimum_quanto_amd.parallel import ColumnParallelQLinear, shard_qweight

    M, K, N = 512, 8192, 8192
    g = torch.Generator(device=device).manual_seed(1234)  # every rank builds the same weight, then keeps its shard
    w = (torch.randn((N, K), generator=g, device=device) * 0.02).to(torch.bfloat16)
    x = torch.randn((M, K), generator=g, device=device).to(torch.bfloat16)
    scale = Q.AbsmaxOptimizer()(w, Q.qfloat8_e4m3fn, 0)
    qw = Q.quantize_weight(w, Q.qfloat8_e4m3fn, 0, scale)
    del w
    layer = ColumnParallelQLinear(shard_qweight(qw, rank, world), None, 1, None)
    del qw
    results = {}
    for mode in ("compute_only", "with_all_gather"):
        fn = (lambda: torch.nn.functional.linear(x, layer.weight)) if mode == "compute_only" else (lambda: layer(x))
        with torch.no_grad():
            for _ in range(args.warmup):
                fn()
            torch.cuda.synchronize()
            t_ramp = time.perf_counter()
            while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
                for _ in range(args.steps):
                    fn()
                torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            elapsed = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt[0])
        results[mode] = elapsed / args.steps
    if rank != 0:
        return None
    flops = 2.0 * M * N * K
    t = results["with_all_gather"]
    return {"metric": "QLinear GEMM TFLOP/s (bf16 x fp8 qbytes_mm, one Linear column-sharded)", "value": round(flops / t / 1e12, 3),
            "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(t * 1e3, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "bf16 x fp8-e4m3fn qbytes_mm, (M,K,N)=(512,8192,8192), output features sharded over the ranks + all_gather",
                       "name": "cfg4_sharded", "M": M, "K": K, "N": N, "launch": "eager (the collective is issued by torch.distributed)",
                       "parallelism": f"column shard x{world}, one all_gather_into_tensor of [M, N/{world}] per call"},
            "compute_only_us": round(results["compute_only"] * 1e6, 2), "with_all_gather_us": round(t * 1e6, 2),
            "compute_only_tflops": round(flops / results["compute_only"] / 1e12, 3)}


def run_stub(args, rank, world, dist):
    """Test-only (--stub): the launch / barrier / max-over-ranks / rank-0-prints skeleton of this file on CPU tensors over gloo, so
    that `bench.py --gpus 2` can be exercised where there is no GPU (tests/test_bench_cli.py).  Measures nothing of the product."""
    x = torch.randn(64, 64)
    for _ in range(args.warmup):
        x @ x
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x @ x
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
    if rank != 0:
        return None
    return {"metric": "stub", "value": round(world * args.steps / elapsed, 3), "unit": "calls/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "stub", "config": {"workload": "CPU stub over gloo (test only)"}}


# ------------------------------------------------------------------------------------------------------------------------
# One Llama-3-8B decoder layer at decode time: the seven int4 QLinears as the FOUR launches the product issues
# ------------------------------------------------------------------------------------------------------------------------
LAYER_LAUNCHES = (("qkv", "qbits_i4_multi", 4096, LLAMA3_QKV), ("o", "qbits_i4", 4096, 4096),
                  ("gate_up", "qbits_i4_multi", 4096, LLAMA3_GATE_UP), ("down", "qbits_i4", 14336, 4096))
LAYER_WORKLOADS = {"layer_decode_b1": 1, "layer_decode_b32": 32}


def layer_algorithmic_bytes(B):
    return sum(algorithmic_work(kind, B, K, N)[1] for _, kind, K, N in LAYER_LAUNCHES)


def build_layer(B, device, seed):
    """Weights of ONE layer per rotation set (109 MB of int4 + scales), enough sets for > 512 MB; the four inputs are independent
    random activations (the ops between the Linears - attention, norms, SiLU - are the model's, not this library's)."""
    g = torch.Generator(device=device).manual_seed(seed)
    rnd = lambda *shape: torch.randn(shape, generator=g, device=device, dtype=torch.float32)  # noqa: E731
    layer_bytes = sum((sum(N) if isinstance(N, tuple) else N) * K // 2 for _, _, K, N in LAYER_LAUNCHES)
    nsets = max(1, -(-(512 << 20) // layer_bytes))
    xs = [rnd(B, K).to(torch.bfloat16).contiguous() for _, _, K, _ in LAYER_LAUNCHES]
    sets = []
    for _ in range(nsets):
        one = []
        for _, kind, K, N in LAYER_LAUNCHES:
            Ns = N if isinstance(N, tuple) else (N,)
            one.append([quantize_int4((rnd(n, K) * 0.02).to(torch.bfloat16).float()) for n in Ns])
        sets.append(one)
    return xs, sets


def make_layer_step(xs, sets):
    """One decode step of the layer: q/k/v (one launch), o, gate/up (one launch), down - the calls the product issues."""
    state = {"i": 0}
    none = {1: [None], 2: [None] * 2, 3: [None] * 3}

    def step():
        cur = sets[state["i"] % len(sets)]
        state["i"] += 1
        for j, (_, kind, K, N) in enumerate(LAYER_LAUNCHES):
            m = cur[j]
            if kind == "qbits_i4_multi":
                torch.ops.quanto.qbits_mm_multi(xs[j], [p for p, _, _ in m], [sc for _, sc, _ in m], [sh for _, _, sh in m], none[len(m)], 4, 128,
                                                list(N), K)
            else:
                p, sc, sh = m[0]
                torch.ops.quanto.qbits_mm(xs[j], p, sc, sh, None, 4, 128, N, K)
    return step


def run_layer_decode(name, args, device, rank, world, dist, steps):
    B = LAYER_WORKLOADS[name]
    xs, sets = build_layer(B, device, seed=4321 + rank)
    nbytes = layer_algorithmic_bytes(B)
    with torch.no_grad():
        elapsed, dev_ms = timed_replay(make_layer_step(xs, sets), steps, args, dist, device)
        # the same launches over TWO layers' weights (218 MB): the working set stays in the 256 MiB Infinity Cache - the rate a perfect
        # prefetcher (next layer's weights pulled in while the model's own kernels run) would give these launches.  NOT an HBM number.
        m_elapsed, _ = timed_replay(make_layer_step(xs, sets[:2]), steps, args, dist, device)
    if rank != 0:
        return None
    us, mall_us = elapsed * 1e6 / steps, m_elapsed * 1e6 / steps
    # (launches of a layer: q/k/v in one, o, gate/up in one, down - LAYER_LAUNCHES; r6: the list, the event time and the rotation count left the line)
    out = {"name": name, "B": B, "us_per_layer": round(us, 3),
           "alg_bytes": int(nbytes), "GBs": round(nbytes * world / us / 1e3, 1), "bound": "hbm", "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4),
           "kernel_us": None, "traffic": None, "cache_resident_us_per_layer": round(mall_us, 3)}
    del sets
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------------------------------
# QConv2d.forward as an implicit GEMM (csrc/qconv_mfma.hip) beside the reference's behaviour on the same device
# ------------------------------------------------------------------------------------------------------------------------
QCONV = {"B": 8, "C": 128, "H": 28, "W": 28, "OC": 128, "k": 3, "stride": 1, "pad": 1}


def build_qconv(wq, device):
    """The frozen QConv2d of the record (weights ``wq``) on the device and its input."""
    import optimum_quanto_amd as Q

    c = QCONV
    torch.manual_seed(0)
    x = torch.randn(c["B"], c["C"], c["H"], c["W"], device=device).to(torch.bfloat16)
    conv = torch.nn.Conv2d(c["C"], c["OC"], c["k"], stride=c["stride"], padding=c["pad"]).to(torch.bfloat16)
    q = Q.QConv2d.from_module(conv, weights=getattr(Q, wq))
    Q.freeze(q)
    return q.to(device), x


def run_qconv2d(args, device, steps=50):
    """One ResNet-style 3x3 layer, bf16 activations: ``QConv2d`` with qint8 and qint4 weights through this library (quanto::qbytes_conv2d /
    quanto::qbits_conv2d: im2col gathered inside the kernel, K split + deterministic reduce) and what the reference computes for the same
    module on a ROCm device (nn/qconv2d.py:54-55 -> qfallback: dequantize the weight, aten convolution = MIOpen).  hipGraph-timed like every
    other record; the weight is small (147 KB / 74 KB), so nothing is rotated.  Algorithmic bytes: input + weight + scales + output, each once."""
    from optimum_quanto_amd.library.hip import quanto_hip

    c = QCONV
    OH = (c["H"] + 2 * c["pad"] - c["k"]) // c["stride"] + 1
    M, K, N = c["B"] * OH * OH, c["C"] * c["k"] * c["k"], c["OC"]
    flops = 2 * M * K * N
    io_bytes = c["B"] * c["C"] * c["H"] * c["W"] * 2 + M * N * 2
    out = {"name": CONV_NAME, "shape": f"({c['B']},{c['C']},{c['H']},{c['W']})->{c['OC']} 3x3 pad 1", "M": M, "K": K, "N": N, "alg_flops": flops,
           "kernel_us": None, "traffic": None}
    with torch.no_grad():
        for wq in ("qint8", "qint4"):
            q, x = build_qconv(wq, device)
            w, bias = q.weight, q.bias
            el, _ = timed_replay(lambda: q(x), steps, args, None, device, warmup=3)
            us = el * 1e6 / steps
            nbytes = io_bytes + (N * K + N * 2 if wq == "qint8" else N * K // 2 + 2 * (N * (K // w._group_size) * 2))
            out[f"{wq[1:]}_us"] = round(us, 2)
            out[f"{wq[1:]}_kernel"] = quanto_hip.lib.last_kernel()
            out[f"{wq[1:]}_alg_bytes"] = nbytes
            out[f"{wq[1:]}_frac_mfma"] = round(flops / us / 1e6 / MFMA_PEAK_TFLOPS, 4)
            out[f"{wq[1:]}_frac_hbm"] = round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)
            el, _ = timed_replay(lambda: torch.nn.functional.conv2d(x, w.dequantize(), bias, c["stride"], c["pad"]), steps, args, None, device, warmup=3)
            out[f"ref_rocm_{wq[1:]}_us"] = round(el * 1e6 / steps, 2)
    # r6: a depthwise layer of the same network family - (8,144,56,56) 3x3 pad 1, groups = 144, int8 weight - on the stencil kernel of csrc/qconv_depthwise.hip
    # against the reference's sequence (dequantize the weight, grouped float convolution); HBM-bound: input + output once
    with torch.no_grad():
        import optimum_quanto_amd as Q
        torch.manual_seed(1)
        dconv = torch.nn.Conv2d(144, 144, 3, padding=1, groups=144).to(torch.bfloat16)
        dq = Q.QConv2d.from_module(dconv, weights=Q.qint8)
        Q.freeze(dq)
        dq = dq.to(device)
        dx = torch.randn(8, 144, 56, 56, device=device).to(torch.bfloat16)
        el, _ = timed_replay(lambda: dq(dx), steps, args, None, device, warmup=3)
        dw_us = el * 1e6 / steps
        out["dw_kernel"] = quanto_hip.lib.last_kernel()
        dw_w, dw_b = dq.weight, dq.bias
        el, _ = timed_replay(lambda: torch.nn.functional.conv2d(dx, dw_w.dequantize(), dw_b, 1, 1, 1, 144), steps, args, None, device, warmup=3)
        out["dw_us"], out["ref_rocm_dw_us"] = round(dw_us, 2), round(el * 1e6 / steps, 2)
        out["dw_frac_hbm"] = round(2 * dx.numel() * 2 / dw_us / 1e3 / HBM_PEAK_GBS, 4)
    # bound: the serial phases of a K-tile inside a workgroup - both roofline fractions are given; ref_rocm_* = dequantize + MIOpen convolution;
    # kernel_us / traffic belong to the int8 launch
    return out


# ------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: Llama-3-8B random-init, weights=qint4 with lm_head excluded, end-to-end tokens/s at batch 1 and 32
# ------------------------------------------------------------------------------------------------------------------------
def run_cfg5(args, device, batches=(1, 32), prompt=512, new=512):
    """The reference's method, call for call (bench/generation/metrics/latency.py:24-105): ``model.generate`` with
    ``max_new_tokens = min_new_tokens = 512``, greedy, eos disabled, a random 512-token prompt and an all-ones mask; device events
    around the whole call (prefill inside the figure, as in the reference); one timed call per batch size after a short warm-up call.
    The model is created from a config on the device (no network), quantized and frozen by this package
    (``QuantizedModelForCausalLM.quantize(weights=qint4, exclude="lm_head")``), sibling projections share launches
    (``fuse_decode_projections``)."""
    from transformers import GenerationConfig, LlamaConfig, LlamaForCausalLM

    import optimum_quanto_amd as Q

    t0 = time.perf_counter()
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
                      vocab_size=128256, max_position_embeddings=8192, rope_theta=500000.0, tie_word_embeddings=False)
    torch.manual_seed(0)
    with torch.device(device):
        model = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    Q.QuantizedModelForCausalLM.quantize(model, weights="qint4", exclude="lm_head")
    linked = Q.fuse_decode_projections(model)
    torch.cuda.synchronize()
    # model: Llama-3-8B random-init bf16, qint4 g128, lm_head excluded; *_tok_s = generate() per latency.py:24-105, graph_* = static-cache decode step replayed
    # from a hipGraph
    out = {"name": "cfg5", "prompt": prompt, "new_tokens": new, "fused_groups": linked,
           "build_s": round(time.perf_counter() - t0, 1), "int4_bytes_per_token": 32 * sum((sum(N) if isinstance(N, tuple) else N) * (K // 2 + K // 128 * 4) for _, _, K, N in LAYER_LAUNCHES)}
    if getattr(model, "generation_config", None) is not None:
        model.generation_config.eos_token_id = None
    with torch.no_grad():
        for b in batches:
            try:
                ids = torch.randint(1, cfg.vocab_size - 1, size=(b, prompt)).to(device)
                mask = torch.ones(b, prompt, dtype=torch.int32).to(device)
                warm = GenerationConfig(max_new_tokens=4, min_new_tokens=4, use_cache=True, pad_token_id=0, num_beams=1, do_sample=False, eos_token_id=None)
                model.generate(ids, attention_mask=mask, generation_config=warm)
                gen = GenerationConfig(max_new_tokens=new, min_new_tokens=new, use_cache=True, pad_token_id=0, num_beams=1, do_sample=False, eos_token_id=None)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                res = model.generate(ids, attention_mask=mask, generation_config=gen)
                e1.record()
                torch.cuda.synchronize()
                assert res.shape[1] == prompt + new
                ms = e0.elapsed_time(e1)
                out[f"b{b}_tok_s"] = round(b * new / (ms * 1e-3), 1)
                out[f"b{b}_ms_per_token"] = round(ms / new, 3)
            except Exception as e:
                out[f"b{b}_error"] = repr(e)[:160]
        # r6: the same model with the Python decode loop out of the way - the single-token forward on a static KV cache captured ONCE in a hipGraph and
        # replayed per token (scripts/bench_generate.py, driver "graph": decode only, prefill reported separately): what the library's kernels are worth
        # once the transformers loop no longer sets the pace
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import bench_generate as BG

            for b in batches:
                try:
                    prefill_ms, decode_ms = BG.run(model, cfg, b, prompt, new, "graph", device)
                    out[f"graph_b{b}_tok_s"] = round(b * new / (decode_ms * 1e-3), 1)
                    out[f"graph_b{b}_ms_per_token"] = round(decode_ms / new, 3)
                    out[f"graph_b{b}_prefill_warmup_capture_ms"] = round(prefill_ms, 1)
                except Exception as e:
                    out[f"graph_b{b}_error"] = repr(e)[:160]
                torch.cuda.empty_cache()
        except Exception as e:
            out["graph_error"] = repr(e)[:160]
        # the binding's own share of a decode step: host time per eager QLinear.forward at the decode shape (2000 back-to-back calls: the
        # loop is host-bound, so wall time / calls = host cost per call), split by layer of the Python stack (scripts/host_overhead.py)
        try:
            out["host_us_per_call"] = host_cost_per_call(model, device)
        except Exception as e:
            out["host_us_per_call_error"] = repr(e)[:120]
    del model
    torch.cuda.empty_cache()
    return out


def host_cost_per_call(model, device, calls=2000):
    """us of host time per call at M = 1 for one of the model's int4 QLinears (o_proj): the module, F.linear on the quantized weight
    (__torch_function__), the torch.ops.quanto op, the ctypes binding, and the raw C entry with pre-marshalled arguments."""
    import ctypes

    from optimum_quanto_amd.library.hip import quanto_hip

    lin = model.model.layers[0].self_attn.o_proj
    w = lin.weight
    x = torch.randn(1, 1, 4096, dtype=torch.bfloat16, device=device)
    y = torch.empty((1, 4096), dtype=torch.bfloat16, device=device)
    c = quanto_hip.lib._c
    raw = (x.data_ptr(), w._data._data.data_ptr(), w._scale.data_ptr(), w._shift.data_ptr(), 0, y.data_ptr(), 1, 4096, 4096, 4, 128, 2, 2, 2, 0, 0,
           ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream))
    legs = {"module": lambda: lin(x), "F_linear": lambda: torch.nn.functional.linear(x, w),
            "op": lambda: torch.ops.quanto.qbits_mm(x, w._data._data, w._scale, w._shift, None, 4, 128, 4096, 4096),
            "binding": lambda: quanto_hip.lib.qbits_mm(x, w._data._data, w._scale, w._shift, None, 4, 128, 4096, 4096),
            "c_entry": lambda: c.quanto_hip_qbits_mm(*raw)}
    res = {}
    for k, fn in legs.items():
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            fn()
        torch.cuda.synchronize()
        res[k] = round((time.perf_counter() - t0) / calls * 1e6, 1)
    return res


# ------------------------------------------------------------------------------------------------------------------------
# rocprofv3 child passes: kernel-only durations and fabric traffic measured by THIS run (not copied from profiles/)
# ------------------------------------------------------------------------------------------------------------------------
MARKER = "unpack_scalar_kernel"  # one 3-byte quanto::unpack launch (a kernel no bench workload uses) separates the segments of a trace
TRACE_STEPS = 12
CONV_NAME = "qconv2d_3x3"


def _child_step(name, device):
    """(step, keep-alive) of a workload inside a rocprofv3 child process - the very step the timed region replays."""
    if name in LAYER_WORKLOADS:
        xs, sets = build_layer(LAYER_WORKLOADS[name], device, seed=4321)
        return make_layer_step(xs, sets), sets
    if name == CONV_NAME:
        q, x = build_qconv("qint8", device)
        return (lambda: q(x)), q
    kind, M, K, N, _ = WORKLOADS[name]
    Nt = sum(N) if isinstance(N, tuple) else N
    wb = Nt * K // 2 if kind.startswith("qbits") else Nt * K
    n_weights = max(1, -(-(512 << 20) // wb)) if M <= 64 else 1
    x, sets = build_inputs(kind, M, K, N, device, min(n_weights, 3 + TRACE_STEPS), seed=1234)
    return make_step(kind, x, sets, K, N), sets


def trace_child(names, args, device):
    """``--trace-child``: run under rocprofv3.  Two segments per workload, each opened by a marker launch:
      * ``graph`` mode (kernel trace): [capture TRACE_STEPS steps in a hipGraph, replay it for the clock ramp] [ONE more replay] - the second
        segment is the steady state the timed region of the parent measures, so its kernel durations can be laid beside ``us_per_step``
        (r4 traced cold, eager launches: a kernel that took LONGER than the step containing it);
      * ``eager`` mode (counter passes, where rocprofv3 serialises the dispatches anyway): [3 warm-up steps] [TRACE_STEPS steps]."""
    from optimum_quanto_amd.library.hip import quanto_hip

    lib = quanto_hip.lib
    tag = torch.zeros(3, dtype=torch.uint8, device=device)  # 3 bytes: the scalar unpack kernel (the vectorised one needs multiples of 16)
    graph_mode = args.trace_mode == "graph"
    for name in names:
        step, keep = _child_step(name, device)
        with torch.no_grad():
            torch.cuda.synchronize()
            lib.unpack(tag, 4)  # opens the warm-up segment (the first call of a workload belongs to it, not to the previous workload's measured one)
            step()
            torch.cuda.synchronize()
            if graph_mode:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        for _ in range(TRACE_STEPS):
                            step()
                torch.cuda.current_stream().wait_stream(side)
                compute_bound = name == CONV_NAME or (name in WORKLOADS and WORKLOADS[name][1] > 64)
                t0, n = time.perf_counter(), 0
                while (time.perf_counter() - t0) * 1e3 < (120.0 if compute_bound else 15.0) or n < 2:
                    g.replay()
                    torch.cuda.synchronize()
                    n += 1
                lib.unpack(tag, 4)
                g.replay()
                torch.cuda.synchronize()
                del g
            else:
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                lib.unpack(tag, 4)
                for _ in range(TRACE_STEPS):
                    step()
                torch.cuda.synchronize()
        del keep, step
        torch.cuda.empty_cache()
    lib.unpack(tag, 4)
    torch.cuda.synchronize()


def _rocprof_pass(names, mode, timeout_s):
    """One child run of this file under rocprofv3 (``mode``: "trace" = --kernel-trace, or a counter name for --pmc).  Returns the
    per-dispatch rows [(kernel_name, start_ns, end_ns, counter_value or None)] in dispatch order, or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    out_dir = tempfile.mkdtemp(prefix="qh_bench_prof_", dir="/tmp")
    flags = ["--kernel-trace"] if mode == "trace" else ["--pmc", mode]
    cmd = [exe, *flags, "--output-format", "csv", "-d", out_dir, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--trace-mode",
           "graph" if mode == "trace" else "eager", "--trace-child", *names]
    env = dict(os.environ, TMPDIR="/tmp", QH_BENCH_CHILD="1")
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, capture_output=True)
        pat = "*kernel_trace.csv" if mode == "trace" else "*counter_collection.csv"
        files = glob.glob(os.path.join(out_dir, "**", pat), recursive=True)
        if not files:
            return None
        if mode == "trace" and os.environ.get("QH_BENCH_KEEP_TRACE"):  # the visit scripts keep the trace the line was computed from
            shutil.copy(files[0], os.environ["QH_BENCH_KEEP_TRACE"])
        rows = []
        for r in csv.DictReader(open(files[0])):
            val = float(r["Counter_Value"]) if mode != "trace" else None
            start = int(r["Start_Timestamp"])
            rows.append((start, int(r.get("Dispatch_Id") or 0), r["Kernel_Name"], start, int(r["End_Timestamp"]), val))
        rows.sort()  # one in-order queue: start time = dispatch order
        return [(k, a, b, v) for _, _, k, a, b, v in rows]
    except Exception:
        return None
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def _segments(rows, names):
    """Split the dispatch list at the marker launches.  Every workload opens TWO segments (warm-up / ramp, then the measured one): the
    measured segment of names[i] is segment 2 i + 1; only the library's kernels (qh::) are kept."""
    segs, cur = [], None
    for k, a, b, v in rows:
        if MARKER in k:
            if cur is not None:
                segs.append(cur)
            cur = []
        elif cur is not None and "qh::" in k:
            cur.append((k, a, b, v))
    return dict(zip(names, segs[1::2])) if len(segs) == 2 * len(names) else None


def collect_profiles(names, timeout_s=300):
    """kernel-only us per step (average and sum-of-minima over the kernels a step launches, from ONE steady-state graph replay) and fabric
    bytes per step for every workload in ``names``; {} when rocprofv3 cannot be used here."""
    if os.environ.get("QH_BENCH_CHILD"):
        return {}
    res = {}
    trace = _rocprof_pass(names, "trace", timeout_s)
    segs = _segments(trace, names) if trace else None
    if trace:
        marks = [(b - a) / 1e3 for k, a, b, _ in trace if MARKER in k]
        if marks:
            res["_trace_floor_us"] = round(min(marks), 3)  # what the profiler stamps for a kernel that does nothing (3-byte unpack launch)
    if segs:
        for n, seg in segs.items():
            per_step = len(seg) // TRACE_STEPS
            if per_step == 0 or len(seg) != per_step * TRACE_STEPS:
                continue
            by_slot = [[(b - a) / 1e3 for (_, a, b, _) in seg[j::per_step]] for j in range(per_step)]  # slot j of a step over the steps
            res[n] = {"kernel_us": round(sum(sum(v) / len(v) for v in by_slot), 3), "kernel_us_min": round(sum(min(v) for v in by_slot), 3),
                      "kernels_per_step": per_step}
    traffic = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = _rocprof_pass(names, counter, timeout_s)
        segs = _segments(rows, names) if rows else None
        if not segs:
            return res
        for n, seg in segs.items():
            if len(seg) >= TRACE_STEPS:
                traffic.setdefault(n, {})[counter] = sum(v for *_, v in seg) / TRACE_STEPS  # KB per step
    for n, t in traffic.items():
        if len(t) == 2:
            # gfx950: FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes -> x 2 (MI355X_MICROARCH.md, HBM section)
            res.setdefault(n, {})["traffic"] = int(t["FETCH_SIZE"] * 1024 * 2 + t["WRITE_SIZE"] * 1024)
    return res


def apply_profile(rec, prof, compacted):
    """Fill kernel_us / traffic of a result from this run's own rocprofv3 passes."""
    if not prof:
        return
    roof = rec if compacted else rec["roofline"]
    for k in (("kernel_us",) if compacted else ("kernel_us", "kernel_us_min")):  # compact records carry the mean only (line budget)
        if k in prof:
            roof[k] = prof[k]
    if "traffic" in prof:
        roof["traffic"] = prof["traffic"]


def compact(r):
    """A sub-result in ~300 bytes: what the reader needs to recompute the roofline fraction (frac = alg_flops or alg_bytes / us_per_step /
    peak; alg_flops = 2 M sum(N) K), nothing repeated.  ``kernel_us`` (steady-state graph replay under rocprofv3) <= ``event_us`` (device
    events around the timed replay) <= ``us_per_step`` (host clock around barrier + synchronize: the time ``value`` uses)."""
    roof, cfg = r["roofline"], r["config"]
    # r6 diet (the default line had reached 8,019 bytes against the driver's 8 KiB stdout tail): device-event time, the minimum of the traced durations,
    # the rotation count and the CPU record's unit / path strings are in --verbose (stderr) only; "cpu_s" = seconds per call of the reference's CPU path
    out = {"name": cfg["name"], "M": cfg["M"], "K": cfg["K"], "N": cfg["N"], "value": r["value"], "unit": r["unit"],
           "us_per_step": round(r["ms_per_step"] * 1e3, 3), "kernel_us": None,
           "kernel": roof["kernel"], "bound": roof["bound"], "frac": roof["frac"], "alg_bytes": int(roof["algorithmic_bytes"]),
           "traffic": roof["traffic"]}
    if "ref_rocm_us" in r:
        out["ref_rocm_us"] = r["ref_rocm_us"]
    if "cpu_baseline" in r:
        out["cpu_s"] = r["cpu_baseline"]["seconds_per_call"]
    return out


# ------------------------------------------------------------------------------------------------------------------------
# Batched decode: where the time of the split-K streaming kernel goes (its built-in ablation switches, csrc/qbits_skinny.hip)
# ------------------------------------------------------------------------------------------------------------------------
# r6: bits 4 / 8 (DMA re-reads the first tile: L2 hits, no HBM) now reach the two-wave-set form the (32,4096,4096) call takes - r5's "no_hbm" rung still
# streamed the weights - and the skeleton is taken apart further: 32 / 64 drop the activation / weight DMA instructions altogether, 128 returns at entry
# (short keys: the line has to stay under the driver's 8 KB tail.  Each rung removes one more part: "-tail" = no split-K tail, "-mfma" = also no MFMA /
# LDS-read work, "-table" = also no scale table, "-hbm" = also L2-resident DMA sources, "-xdma" / "-dma" = also no activation / no DMA instruction at all)
ABLATIONS = (("full", 0), ("-tail", 1), ("-mfma", 3), ("-table", 19), ("-hbm", 31), ("-xdma", 31 + 32), ("-dma", 31 + 32 + 64), ("launch", 128))


def ablate_child(name, args, device):
    """``--ablate-child``: a child process with the library's experiment knobs on (they are read only behind QUANTO_HIP_EXPERIMENT, which the
    parent - and the driver's command - does not set).  Times the same hipGraph replay with QUANTO_HIP_SKINNY_ABLATE = 0 (the product), 1 (no
    partial-sum store / arrival counter / reduce: every block but split 0 returns after its K loop), 3 (+ no MFMA / LDS-read work: the DMA stream
    alone), 19 (+ no scale / shift table), 31 (+ every DMA re-reads tile 0: L2 hits, no HBM traffic), 63 (+ no activation DMA instruction), 127 (+ no
    weight DMA either: prologue, barriers and loop control only), 128 (the kernel returns at entry: launch + grid dispatch).  Differences between
    neighbouring rungs = the ``skeleton_breakdown`` the r5 review asked for.  Results are WRONG by construction."""
    kind, M, K, N, _ = WORKLOADS[name]
    Nt = sum(N) if isinstance(N, tuple) else N
    n_weights = max(1, -(-(512 << 20) // (Nt * K // 2)))
    x, sets = build_inputs(kind, M, K, N, device, n_weights, seed=1234)
    res = {}
    for label, bits in ABLATIONS:
        os.environ["QUANTO_HIP_SKINNY_ABLATE"] = str(bits)
        step = make_step(kind, x, sets, K, N)
        with torch.no_grad():
            elapsed, _ = timed_replay(step, 200, args, None, device, warmup=3)
        res[label] = round(elapsed * 1e6 / 200, 3)
    os.environ.pop("QUANTO_HIP_SKINNY_ABLATE", None)
    print(json.dumps({"ablate": res}), flush=True)


def run_ablation(name, timeout_s=120):
    """Parent side: one child run of this file with the experiment switch set; {} when it fails."""
    import subprocess

    env = dict(os.environ, QUANTO_HIP_EXPERIMENT="1", QH_BENCH_CHILD="1")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--ablate-child", name], env=env, capture_output=True, text=True, timeout=timeout_s)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        return json.loads(line)["ablate"]
    except Exception:
        return {}


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` with no launcher around it: start N ranks of this same command, one per GPU."""
    import subprocess

    # --standalone: torchrun picks (and holds) a free rendezvous port itself - probing one here and closing the socket left a window
    # for another process to take it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n}",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: WORLD_SIZE when a launcher set it, else 1")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))  # sub-results may also name layer_decode_b1 / _b32
    ap.add_argument("--sub", nargs="*", default=None, help="workloads reported under sub_results (default: the int4 decode half of the metric)")
    ap.add_argument("--no-sub", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=6.0, help="seconds of host time per timed CPU path")
    ap.add_argument("--shard", action="store_true", help="configs[3] column-sharded over the ranks (ColumnParallelQLinear + all_gather) as the headline line")
    ap.add_argument("--eager", action="store_true", help="issue the timed steps one by one from Python instead of replaying a hipGraph")
    ap.add_argument("--ramp-ms", type=float, default=300.0,
                    help="untimed: keep the device busy with the same steps for this long before the timed region (clock ramp)")
    ap.add_argument("--verbose", action="store_true", help="also print every sub-result in full on stderr")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)  # test-only: CPU + gloo skeleton run (run_stub)
    ap.add_argument("--no-ref-rocm", action="store_true", help="skip timing the reference's ROCm op sequence (ref_rocm_us)")
    ap.add_argument("--no-cfg5", action="store_true", help="skip BASELINE configs[4] (Llama-3-8B end-to-end tokens/s, ~1 min)")
    ap.add_argument("--no-profile", action="store_true", help="skip the rocprofv3 child passes (kernel_us, traffic) of the default run")
    ap.add_argument("--profile", action="store_true", help="run the rocprofv3 child passes for a non-default selection of workloads too")
    ap.add_argument("--trace-child", nargs="+", default=None, help=argparse.SUPPRESS)  # the child run of collect_profiles
    ap.add_argument("--trace-mode", default="graph", choices=["graph", "eager"], help=argparse.SUPPRESS)
    ap.add_argument("--ablate-child", default=None, help=argparse.SUPPRESS)  # the child run of run_ablation
    args = ap.parse_args()

    if args.gpus is None:  # an external launcher's WORLD_SIZE is enough (torchrun --nproc-per-node 8 bench.py --shard)
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    use_gpu = not args.stub
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_gpu and torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.stub:
        out = run_stub(args, rank, world, dist)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import optimum_quanto_amd  # noqa: F401  registers the quanto:: ops; raises if the HIP library cannot be loaded later

    if args.trace_child is not None:
        trace_child(args.trace_child, args, device)
        return
    if args.ablate_child is not None:
        ablate_child(args.ablate_child, args, device)
        return
    if args.shard:
        out = run_sharded(args, device, rank, world, dist)
    else:
        with_cpu = not args.no_cpu_baseline and world == 1  # the host baseline is reported at N = 1 only
        out = run_workload(args.workload, args, device, rank, world, dist, args.steps, with_cpu)
        default_run = args.workload == "cfg2" and args.sub is None and not args.no_sub
        subs = [] if args.no_sub else (args.sub if args.sub is not None else (DEFAULT_SUB if args.workload == "cfg2" else []))
        sub_results = []
        for name in subs:
            # decode launches last a few microseconds: time at least 200 of them so that the event pair brackets milliseconds
            if name in LAYER_WORKLOADS:
                r = run_layer_decode(name, args, device, rank, world, dist, max(args.steps, 100))
                if r is not None:
                    sub_results.append(r)
                continue
            r = run_workload(name, args, device, rank, world, dist, max(args.steps, 200), with_cpu)
            if r is not None:
                if args.verbose:
                    print(json.dumps(r), file=sys.stderr, flush=True)
                sub_results.append(compact(r))
        if world > 1 and not args.no_sub and args.workload == "cfg2":
            # N > 1: the one place the path has a collective - configs[3] column-sharded over the ranks, strong scaling
            r = run_sharded(args, device, rank, world, dist)
            if r is not None:
                sub_results.append({"name": "cfg4_sharded", "scaling": "strong", "n_gpus": world, "value": r["value"], "unit": r["unit"],
                                    "compute_only_us": r["compute_only_us"], "with_all_gather_us": r["with_all_gather_us"],
                                    "compute_only_tflops": r["compute_only_tflops"], "parallelism": r["config"]["parallelism"]})
        if world == 1 and rank == 0 and default_run and out is not None:
            dec = next((sr for sr in sub_results if sr.get("name") == "int4_decode32"), None)
            if dec is not None:  # us per launch with parts of the kernel switched off: tail = full - no_tail, MFMA work = no_tail - no_tail_no_mfma, ...
                dec["ablate_us"] = run_ablation("int4_decode32")
        conv_rec = None
        if world == 1 and rank == 0 and default_run and out is not None:
            try:
                conv_rec = run_qconv2d(args, device)
            except Exception as e:
                conv_rec = {"name": CONV_NAME, "error": repr(e)[:200]}
            sub_results.append(conv_rec)
        if world == 1 and rank == 0 and (default_run or args.profile) and not args.no_profile and out is not None:
            # this run's own kernel-only durations and counter traffic (child processes under rocprofv3; nothing here is timed)
            names = [args.workload] + [sr["name"] for sr in sub_results if sr["name"] in WORKLOADS or sr["name"] in LAYER_WORKLOADS]
            if conv_rec is not None and "error" not in conv_rec:
                names.append(CONV_NAME)
            t_prof = time.perf_counter()
            prof = collect_profiles(names)
            apply_profile(out, prof.get(args.workload), compacted=False)
            for sr in sub_results:
                apply_profile(sr, prof.get(sr["name"]), compacted=True)
            out["profile_passes"] = {"ok": bool(prof), "seconds": round(time.perf_counter() - t_prof, 1), "trace_floor_us": prof.get("_trace_floor_us"),
                                     "what": "bench.py docstring (Round 4 additions): rocprofv3 child passes of this run"}
        if world == 1 and rank == 0 and default_run and not args.no_cfg5 and out is not None:
            try:
                rec = run_cfg5(args, device)
                # the library's kernels in one decoded token = 32 layers x the four launches of layer_decode_b1 / _b32 (same shapes, same kernels)
                for b, lname in ((1, "layer_decode_b1"), (32, "layer_decode_b32")):
                    lay = next((sr for sr in sub_results if sr.get("name") == lname), None)
                    if lay is not None and lay.get("kernel_us") is not None:
                        rec[f"b{b}_qh_kernel_ms_per_token"] = round(32 * lay["kernel_us"] / 1e3, 3)
                sub_results.append(rec)
            except Exception as e:  # transformers missing / out of memory: the GEMM records stand on their own
                sub_results.append({"name": "cfg5", "error": repr(e)[:200]})
        if out is not None and sub_results:
            out["sub_results"] = sub_results
        if out is not None and "cpu_baseline" in out:
            out["cpu_baseline"]["how"] = "the reference's CPU QLinear path (its ATen kernels in its order, oracle/reference_cpu_path.py; path codes: README.md, Benchmark)"
    if rank == 0 and out is not None:
        print(json.dumps(out, separators=(",", ":")), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
