"""TEST / BENCH INFRASTRUCTURE ONLY - the reference's CPU QLinear path restated as the ATen call sequence it executes.

The reference is a Python package whose arithmetic lives in torch: on CPU tensors ``QLinear.forward`` bottoms out in a
handful of ATen kernels (``torch._weight_int8pack_mm``, ``torch._int_mm``, ``torch._weight_int4pack_mm_for_cpu``, or
elementwise ops + ``torch.matmul``).  ``/root/reference`` does not exist on the GPU box, torch (the same 2.10 build) does:
this module issues exactly those kernels, in the reference's order, so that ``bench.py::cpu_baseline`` times the
reference's own CPU arithmetic on the GPU box's host cores.  Every function cites the reference lines it follows
(paths relative to /root/reference/optimum/quanto).

Pinned: ``tests/test_reference_integration.py`` (``tests/reference_subprocess.py cpu_path``) runs the real reference (scratch copy,
subprocess) in the build container and
asserts ``torch.equal`` between its outputs and these functions' on the same tensors, for every path below.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product never does.
"""
import time

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------------------------------
# 8-bit weights: quanto::qbytes_mm on CPU tensors
# ---------------------------------------------------------------------------------------------------------------------
def qbytes_mm_generic(activations, weights, output_scales):
    """library/qbytes_mm.py:25-33 - scale the weights in the output dtype, then a dense matmul."""
    activations = activations.to(output_scales.dtype)
    if weights.dtype.is_floating_point:
        weights = weights.to(output_scales.dtype)
    weights = output_scales * weights
    return torch.matmul(activations, weights.t())


def qbytes_int_mm(activations, weights, output_scales):
    """library/qbytes_mm.py:36-50 - int8 x int8 -> int32 (torch._int_mm), fp32 rescale, cast."""
    in_features, out_features = activations.shape[-1], weights.shape[0]
    out = torch._int_mm(activations.reshape(-1, in_features), weights.t()).reshape(activations.shape[:-1] + (out_features,))
    return (out.to(torch.float32) * output_scales.t()).to(output_scales.dtype)


def qbytes_int8pack_mm(activations, weights, output_scales):
    """library/qbytes_mm.py:53-64 - torch._weight_int8pack_mm (bf16 activations, int8 weights, vector of scales)."""
    in_features, out_features = activations.shape[-1], weights.shape[0]
    out = torch._weight_int8pack_mm(activations.reshape(-1, in_features), weights, output_scales.flatten())
    return out.reshape(activations.shape[:-1] + (out_features,))


def qbytes_mm_cpu(activations, weights, output_scales):
    """The CPU registration, library/qbytes_mm.py:91-105 (torch >= 2.6 branch)."""
    if activations.dtype == torch.int8 and weights.dtype == torch.int8:
        return qbytes_int_mm(activations, weights, output_scales)
    if activations.dtype == torch.bfloat16 and weights.dtype == torch.int8 and activations.shape[-1] % 4 == 0:
        return qbytes_int8pack_mm(activations, weights, output_scales)
    return qbytes_mm_generic(activations, weights, output_scales)


def qbytes_linear(x, data, scale, bias=None):
    """WeightQBytesLinearFunction.forward, tensor/weights/qbytes.py:68-82 (float activations)."""
    in_features, out_features = x.shape[-1], data.shape[0]
    out = qbytes_mm_cpu(x.reshape(-1, in_features), data, scale).view(x.shape[:-1] + (out_features,))
    return out if bias is None else out + bias


# ---------------------------------------------------------------------------------------------------------------------
# sub-byte weights, generic class: unpack -> dequantize -> matmul on every call
# ---------------------------------------------------------------------------------------------------------------------
def unpack(packed, bits):
    """library/unpack.py:21-54 (the python implementation; the reference JIT-builds an equivalent C++ loop for CPU tensors,
    library/extensions/cpp/unpack.cpp:19-47)."""
    planes = []
    for i in range(8 // bits):
        mask = 2 ** (bits * (i + 1)) - 1
        planes.append((packed & mask) >> (bits * i))
    return torch.cat(planes).to(torch.uint8)


def dequantize_qbits(packed, scale, shift, bits, group_size, out_features, in_features):
    """PackedTensor.unpack (tensor/packed.py:101-104) + QBitsDequantizer.forward (tensor/qbits.py:27-49) + ungroup for
    axis 0 (tensor/grouped.py:39-44: a reshape)."""
    rows = out_features * in_features // group_size if group_size is not None else out_features
    data = unpack(packed, bits)[:rows]
    if not shift.dtype.is_floating_point:
        data = data.to(torch.int8) - shift.to(torch.int8)
    dqt = scale * data
    if shift.dtype.is_floating_point:
        dqt -= shift
    return dqt.reshape(out_features, in_features)


def qbits_linear_generic(x, packed, scale, shift, bits, group_size, out_features, in_features, bias=None):
    """QuantizedLinearFunction.forward through qfallback, tensor/function.py:41-47: dequantize the whole weight, matmul."""
    w = dequantize_qbits(packed, scale, shift, bits, group_size, out_features, in_features)
    out = torch.matmul(x, w.t())
    return out if bias is None else out + bias


# ---------------------------------------------------------------------------------------------------------------------
# int4 weights with bf16 scale/shift on CPU: the TinyGemm subclass WeightQBitsTensor.create() selects
# (tensor/weights/qbits.py:119-136)
# ---------------------------------------------------------------------------------------------------------------------
def tinygemm_pack(packed, scale, shift, group_size, out_features, in_features):
    """TinyGemmWeightQBitsTensor.__init__ (tensor/weights/tinygemm/qbits.py:84-107) + TinyGemmPackedTensor.pack
    (tensor/weights/tinygemm/packed.py:41-61): generic layout -> (int4pack data, [G, N, 2] scale / mid-point shift).
    One-off at load time in the reference; not part of the timed call."""
    rows = out_features * in_features // group_size
    ungrouped = unpack(packed, 4)[:rows].reshape(out_features, in_features)
    data = torch._convert_weight_to_int4pack_for_cpu(ungrouped.to(torch.int32).contiguous(), innerKTiles=2)
    scale = scale.reshape(out_features, in_features // group_size, 1)
    shift = shift.reshape(out_features, in_features // group_size, 1)
    if not shift.dtype.is_floating_point:
        shift = scale * shift
    shift = -shift + 2 ** 3 * scale  # mid-point of the quantization range (lossy in bf16, as the reference notes)
    scale_shift = torch.cat([scale, shift], 2).transpose(0, 1).contiguous()
    return data, scale_shift


def tinygemm_linear(x, data, group_size, scale_shift, out_features, bias=None):
    """TinyGemmQBitsLinearFunction.forward, tensor/weights/tinygemm/qbits.py:42-62 (CPU branch :51-54)."""
    in_features = x.shape[-1]
    out = torch._weight_int4pack_mm_for_cpu(x.reshape(-1, in_features), data, group_size, scale_shift)
    out = out.reshape(x.shape[:-1] + (out_features,))
    return out if bias is None else out + bias


# ---------------------------------------------------------------------------------------------------------------------
# timing protocol of SURVEY.md 8(d): no_grad, warm-up 3, >= min_calls timed calls, median + IQR, thread count recorded
# ---------------------------------------------------------------------------------------------------------------------
def time_call(fn, budget_s=10.0, warmup=3, min_calls=5, max_calls=50):
    with torch.no_grad():
        for _ in range(warmup):
            fn()
        times, t_start = [], time.perf_counter()
        while len(times) < min_calls or (time.perf_counter() - t_start < budget_s and len(times) < max_calls):
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
    q1, med, q3 = np.percentile(times, [25, 50, 75])
    return {"median_s": float(med), "iqr_s": float(q3 - q1), "calls": len(times), "threads": torch.get_num_threads()}
