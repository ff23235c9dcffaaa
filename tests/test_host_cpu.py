"""CPU-only checks of the host-side mirror (no GPU needed):

* the torch host code reproduces the reference's golden vectors bit for bit (it is the data path that builds the
  packed weights the kernels read);
* libquanto_hip.so builds/loads and exports every symbol include/quanto_hip.h declares (no compute calls);
* the product path refuses to run device work without the library (no silent CPU fallback).
"""
import ctypes
import io
import os
import re

import numpy as np
import pytest
import torch

import optimum_quanto_amd as Q
from optimum_quanto_amd.library.hip import quanto_hip
from optimum_quanto_amd.tensor.packing import PackedTensor, pack_weights

from helpers import TORCH_DT, to_numpy, to_torch
from oracle import quanto_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "quanto_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(quanto_hip_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 8
    path = quanto_hip.build()  # no-op when up to date; hipcc cross-compiles without a GPU
    lib = ctypes.CDLL(path)
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/quanto_hip.h but not exported"
    lib.quanto_hip_abi_version.restype = ctypes.c_int
    assert lib.quanto_hip_abi_version() == 1
    lib.quanto_hip_status_string.restype = ctypes.c_char_p
    assert lib.quanto_hip_status_string(-1) == b"invalid argument"


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    lib = quanto_hip.cdll
    lib.quanto_hip_unpack.restype = ctypes.c_int
    lib.quanto_hip_unpack.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    assert lib.quanto_hip_unpack(None, None, 16, 3, None) == -1  # bits must be 2 or 4
    assert lib.quanto_hip_unpack(None, None, -1, 4, None) == -1
    assert lib.quanto_hip_unpack(None, None, 16, 4, None) == -1  # null pointers
    lib.quanto_hip_qbits_mm_workspace_size.restype = ctypes.c_int64
    lib.quanto_hip_qbits_mm_workspace_size.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int] * 4
    assert lib.quanto_hip_qbits_mm_workspace_size(4096, 4096, 4096, 4, 128, 2, 0) == 4096 * 4096 * 2  # prefill: dequantized weight
    assert lib.quanto_hip_qbits_mm_workspace_size(4096, 4096, 4096, 4, 128, 2, 3) == 32 * 4096 * 4  # kernel MFMA: per-group row sums of x
    # above the streaming kernel's range, while one round of 128 x 128 tiles covers the problem: the fused int4 GEMM - no scratch when
    # the scale table of all groups fits the LDS, split-K scratch (counter region + fp32 partial tiles of 512 lanes x 128 B) otherwise
    assert lib.quanto_hip_qbits_mm_workspace_size(1024, 4096, 4096, 4, 128, 2, 0) == 0
    # few tiles (5 x 2 of 64 tokens): K split 8 ways, counter region + fp32 partial tiles of 512 lanes x 64 B
    assert lib.quanto_hip_qbits_mm_workspace_size(300, 256, 4096, 4, 128, 2, 0) == 4096 + 10 * 8 * 512 * 64
    assert lib.quanto_hip_qbits_mm_workspace_size(128, 4096, 4096, 4, 128, 2, 0) == 4096 + 64 * 4 * 512 * 64
    # K = 14336: split 2 ways (128 tiles of 64 tokens), counter region + fp32 partial tiles
    assert lib.quanto_hip_qbits_mm_workspace_size(256, 4096, 14336, 4, 128, 2, 0) == 4096 + 128 * 2 * 512 * 64
    assert lib.quanto_hip_qbits_mm_pick(300, 256, 4096, 4, 128, 2) == 8
    assert lib.quanto_hip_qbits_mm_workspace_size(300, 256, 4096, 4, 128, 2, 7) == 256 * 4096 * 2  # DEQUANT_MFMA: the dequantized weight
    assert lib.quanto_hip_qbits_mm_workspace_size(40, 200, 512, 4, 64, 2, 0) == 8 * 128 * 4  # group size 64, N not in 64-feature blocks: 128x128 kernel (row sums of x)
    assert lib.quanto_hip_qbits_mm_workspace_size(40, 256, 512, 4, 64, 2, 0) == 0  # group size 64: streaming kernel, too few tiles to split
    # streaming MFMA kernel, N = 4096: 256 waves -> K split 4 ways; the fixed 4 KiB counter region (QUANTO_HIP_WS_COUNTER_BYTES) + fp32 partials (TF = 4)
    assert lib.quanto_hip_qbits_mm_workspace_size(64, 4096, 4096, 4, 128, 2, 0) == 4096 + 256 * 4 * 64 * 4 * 16
    # N = 14336: 224 blocks of 64 features -> split 2 (448 blocks, two to three per CU)
    assert lib.quanto_hip_qbits_mm_workspace_size(64, 14336, 4096, 4, 128, 2, 0) == 4096 + 896 * 2 * 64 * 4 * 16
    assert lib.quanto_hip_qbits_mm_workspace_size(64, 32768, 4096, 4, 128, 2, 0) == 0  # wide enough: not split
    lib.quanto_hip_qbits_mm_pick.restype = ctypes.c_int
    lib.quanto_hip_qbits_mm_pick.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int] * 3
    assert lib.quanto_hip_qbits_mm_pick(64, 4096, 4096, 4, 128, 2) == 5 and lib.quanto_hip_qbits_mm_pick(1, 4096, 4096, 4, 128, 2) == 2
    # streaming kernel up to 64 rows, fused int4 GEMM while its modelled time stays within 1.2 rounds of 128-token tiles, dequantize + dense GEMM beyond
    assert [lib.quanto_hip_qbits_mm_pick(m, 4096, 4096, 4, 128, 2) for m in (64, 65, 192, 256, 1024, 2048)] == [5, 8, 8, 8, 8, 7]
    assert lib.quanto_hip_qbits_mm_pick(256, 4096, 14336, 4, 128, 2) == 8  # K = 14336: fused with K split 2 ways (56 groups' scale table per workgroup)
    # small decode batches: the register-streaming kernel where one block per CU covers N, the LDS-streaming one elsewhere
    assert [lib.quanto_hip_qbits_mm_pick(m, 4096, 4096, 4, 128, 2) for m in (4, 5, 16, 17)] == [2, 9, 9, 5]
    assert [lib.quanto_hip_qbits_mm_pick(8, n, k, 4, 128, 2) for n, k in ((1024, 4096), (14336, 4096), (4096, 14336), (5120, 5120))] == [9, 5, 5, 5]
    assert lib.quanto_hip_qbits_mm_workspace_size(2048, 4096, 4096, 4, 128, 2, 0) == 4096 * 4096 * 2
    # r3: group sizes 64 / 32 (64-feature blocks) and per-channel scales run the streaming kernel from 5 rows up to 192 (r4: group size 96 and
    # qint2 with group size 128 as well); what it does not take keeps the GEMV passes (<= 24 rows), everything beyond goes through dequantize + dense GEMM
    SKINNY, GEMV, DEQUANT, MFMA128 = 5, 2, 7, 3
    assert [lib.quanto_hip_qbits_mm_pick(m, 4096, 4096, 4, 64, 2) for m in (4, 5, 64, 192, 193)] == [GEMV, SKINNY, SKINNY, SKINNY, DEQUANT]
    assert [lib.quanto_hip_qbits_mm_pick(m, 4096, 4096, 4, 32, 2) for m in (8, 128)] == [SKINNY, SKINNY]
    assert [lib.quanto_hip_qbits_mm_pick(m, 4096, 4096, 4, 0, 2) for m in (8, 128)] == [SKINNY, SKINNY]       # per-channel
    assert lib.quanto_hip_qbits_mm_pick(40, 200, 512, 4, 64, 2) == MFMA128                                      # N not in 64-feature blocks
    assert [lib.quanto_hip_qbits_mm_pick(m, 4096, 1152, 4, 96, 2) for m in (4, 8, 25, 192, 193)] == [GEMV, SKINNY, SKINNY, SKINNY, DEQUANT]  # group size 96 (r4)
    assert [lib.quanto_hip_qbits_mm_pick(8, n, k, 4, 96, 2) for n, k in ((4000, 1152), (4096, 96))] == [GEMV, GEMV]  # N not in 64-feature blocks; one group = per-channel
    assert [lib.quanto_hip_qbits_mm_pick(m, 4096, 4096, 2, 128, 2) for m in (4, 8, 25, 192, 193)] == [GEMV, SKINNY, SKINNY, SKINNY, DEQUANT]  # qint2 (r4)
    assert lib.quanto_hip_qbits_mm_pick(8, 4096, 4096, 2, 64, 2) == GEMV                                          # qint2, group size 64: GEMV passes
    LARGE4 = 10
    # r5: with a workspace dequantize + dense GEMM at every prefill size (the 128-byte-row dense kernel); the large-tile int4 GEMM only without one
    assert [lib.quanto_hip_qbits_mm_pick(m, 8192, 8192, 4, 128, 2) for m in (2048, 4096, 8192)] == [DEQUANT, DEQUANT, DEQUANT]
    assert lib.quanto_hip_qbits_mm_pick(4096, 28672, 8192, 4, 128, 2) == DEQUANT and lib.quanto_hip_qbits_mm_pick(8192, 14336, 4096, 4, 128, 2) == DEQUANT
    assert lib.quanto_hip_qbits_mm_workspace_size(8192, 8192, 8192, 4, 128, 2, 0) == 8192 * 8192 * 2                   # the dense weight
    assert lib.quanto_hip_qbits_mm_workspace_size(8192, 8192, 8192, 4, 128, 2, LARGE4) == 0                          # forced: it needs no workspace
    # int8, M = 96 off the fitted grid (r4): the tile kernel from 40 tiles on while K is short
    assert [lib.quanto_hip_qbytes_mm_pick(96, n, k, 2, 3, 2) for n, k in ((5120, 5120), (2048, 2048), (5120, 11008), (8192, 8192))] == [4, 5, 5, 4]
    assert lib.quanto_hip_qbits_mm_pick(64, 192, 14336, 4, 32, 2) == DEQUANT  # 448 groups' tables do not fit next to the ring
    assert lib.quanto_hip_qbits_mm_workspace_size(4, 4096, 4096, 4, 128, 2, 0) == 0
    assert lib.quanto_hip_qbits_mm_workspace_size(1, 4096, 4096, 4, 128, 2, 0) == 0  # GEMV needs none
    assert lib.quanto_hip_qbits_mm_workspace_size(1, 4096, 4096, 3, 128, 2, 0) == -1
    lib.quanto_hip_quantize_affine_packed.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 2 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    assert lib.quanto_hip_quantize_affine_packed(None, None, None, None, 8, 128, 3, 128, 2, 2, None) == -1   # bits
    assert lib.quanto_hip_quantize_affine_packed(None, None, None, None, 8, 100, 4, 64, 2, 2, None) == -1   # K % group
    assert lib.quanto_hip_quantize_affine_packed(None, None, None, None, 8, 128, 4, 128, 2, 2, None) == -1  # null pointers
    # entry points added for the "next" rows: argument validation happens before any device call
    vp, i64, ci, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
    lib.quanto_hip_quantize_symmetric.argtypes = [vp, vp, vp, i64, i64, ci, ci, ci, vp]
    assert lib.quanto_hip_quantize_symmetric(None, None, None, 16, 1, 0, 2, 3, None) == -1   # null pointers
    assert lib.quanto_hip_quantize_symmetric(None, None, None, 16, 1, 7, 2, 3, None) == -1   # unknown scale mode
    assert lib.quanto_hip_quantize_symmetric(None, None, None, 16, 5, 1, 2, 3, None) == -1   # numel not a multiple of inner
    assert lib.quanto_hip_quantize_symmetric(None, None, None, 16, 1, 0, 2, 7, None) == -2   # e4m3fnuz target: not supported
    assert lib.quanto_hip_quantize_symmetric(None, None, None, 0, 1, 0, 2, 3, None) == 0     # empty tensor
    lib.quanto_hip_dequantize_symmetric.argtypes = [vp, vp, vp, i64, ci, ci, vp]
    assert lib.quanto_hip_dequantize_symmetric(None, None, None, 16, 3, 2, None) == -1     # null pointers
    assert lib.quanto_hip_dequantize_symmetric(None, None, None, 16, 2, 2, None) == -2     # bf16 "quantized" data
    assert lib.quanto_hip_dequantize_symmetric(None, None, None, 16, 3, 3, None) == -2     # int8 output
    assert lib.quanto_hip_dequantize_symmetric(None, None, None, 0, 5, 1, None) == 0       # empty tensor
    lib.quanto_hip_quantize_affine.argtypes = [vp, vp, vp, vp, i64, i64, ci, ci, ci, ci, vp]
    assert lib.quanto_hip_quantize_affine(None, None, None, None, 8, 128, 4, 128, 2, 2, None) == -1
    assert lib.quanto_hip_quantize_affine(None, None, None, None, 8, 100, 4, 128, 2, 2, None) == -1  # K % group_size
    lib.quanto_hip_pack.argtypes = [vp, vp, i64, i64, ci, vp]
    assert lib.quanto_hip_pack(None, None, 8, 8, 3, None) == -1 and lib.quanto_hip_pack(None, None, 0, 8, 4, None) == 0
    lib.quanto_hip_qbytes_mm_ws.argtypes = [vp] * 5 + [i64] * 3 + [ci] * 4 + [vp, sz, vp]
    assert lib.quanto_hip_qbytes_mm_ws(None, None, None, None, None, 4, 4, 64, 2, 3, 2, 0, None, 0, None) == -1
    assert lib.quanto_hip_qbytes_mm_ws(None, None, None, None, None, 0, 4, 64, 2, 3, 2, 0, None, 0, None) == 0   # M == 0
    lib.quanto_hip_qbytes_mm_workspace_size.restype = i64
    lib.quanto_hip_qbytes_mm_workspace_size.argtypes = [i64] * 3 + [ci] * 4
    assert lib.quanto_hip_qbytes_mm_workspace_size(32, 14336, 4096, 2, 3, 2, 0) == 4096 + 224 * 2 * 256 * 2 * 16  # 224 blocks -> split 2
    assert lib.quanto_hip_qbytes_mm_workspace_size(32, 32768, 4096, 2, 3, 2, 0) == 0           # wide N: not split
    assert lib.quanto_hip_qbytes_mm_workspace_size(32, 4096, 4096, 2, 3, 2, 0) == 4096 + 64 * 4 * 256 * 2 * 16  # fixed counter region, split 4, TF = 2
    assert lib.quanto_hip_qbytes_mm_workspace_size(4096, 4096, 4096, 2, 3, 2, 0) == 0          # 256-tiles: no workspace
    lib.quanto_hip_qbytes_mm_pick.argtypes = [i64] * 3 + [ci] * 3
    picks = {m: lib.quanto_hip_qbytes_mm_pick(m, 4096, 4096, 2, 3, 2) for m in (1, 8, 64, 256, 4096)}
    assert picks == {1: 2, 8: 5, 64: 5, 256: 4, 4096: 4}, picks                                  # gemv, skinny, skinny, large, large
    # one 128-row tile band beats the streaming kernel's passes of 64 rows from ~100 rows on, even on a handful of tiles
    assert [lib.quanto_hip_qbytes_mm_pick(m, 1024, 4096, 2, 3, 2) for m in (96, 128)] == [5, 4]
    # split-K of the 128-tile grid while its tiles cover at most half of the CUs: 2 ways, 4 ways for a single row of tiles
    assert lib.quanto_hip_qbytes_mm_workspace_size(512, 4096, 4096, 2, 3, 2, 0) == 4096 + 128 * 2 * 128 * 128 * 4
    assert lib.quanto_hip_qbytes_mm_workspace_size(128, 4096, 4096, 2, 3, 2, 0) == 4096 + 32 * 4 * 128 * 128 * 4
    assert lib.quanto_hip_qbytes_mm_workspace_size(1024, 4096, 4096, 2, 3, 2, 0) == 0  # 256 tiles: every CU has one
    assert lib.quanto_hip_qbytes_mm_workspace_size(512, 4096, 14336, 2, 3, 2, 0) == 4096 + 128 * 2 * 128 * 128 * 4  # fixed counter region + fp32 partials
    assert lib.quanto_hip_qbytes_mm_pick(4096, 4096, 4096, 3, 3, 2) == 6                         # int8 activations: native8
    # native8 (r6): 128-tiles split over K where K is long for the output it feeds - (256,8192,8192) = 128 tiles x 2, (512,4096,14336) = 128 tiles x 4,
    # (128,4096,4096) = 32 tiles x 4; int32 / fp32 partial tiles of 64 KiB
    assert lib.quanto_hip_qbytes_mm_workspace_size(256, 8192, 8192, 5, 5, 2, 0) == 4096 + 128 * 2 * 128 * 128 * 4
    assert lib.quanto_hip_qbytes_mm_workspace_size(512, 8192, 8192, 3, 3, 2, 0) == 0              # 256 tiles: two split workgroups per CU lose to none
    assert lib.quanto_hip_qbytes_mm_workspace_size(512, 4096, 14336, 3, 3, 2, 0) == 4096 + 128 * 4 * 128 * 128 * 4
    assert lib.quanto_hip_qbytes_mm_workspace_size(128, 4096, 4096, 5, 5, 2, 0) == 4096 + 32 * 4 * 128 * 128 * 4
    assert lib.quanto_hip_qbytes_mm_workspace_size(1024, 4096, 4096, 3, 3, 2, 0) == 0             # 256 tiles, K = 4096: not split
    assert lib.quanto_hip_qbytes_mm_workspace_size(4096, 4096, 4096, 3, 3, 2, 0) == 0             # 256-tiles: not split
    assert lib.quanto_hip_qbytes_mm_workspace_size(512, 8192, 8128, 3, 3, 2, 0) == 0              # K % 128 != 0: the 64-byte-row kernel has no split


def test_conv2d_entries_reject_bad_arguments_without_a_gpu():
    """quanto_hip_q{bytes,bits}_conv2d check their geometry and pointers before anything is launched: inconsistent output sizes, zero strides,
    null tensors -> EINVAL; non-float outputs -> ENOTSUP; an empty batch is a no-op."""
    lib = quanto_hip.cdll
    vp, i64, ci, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
    f = lib.quanto_hip_qbytes_conv2d
    f.restype, f.argtypes = ci, [vp] * 5 + [i64] * 9 + [ci] * 9 + [vp, sz, vp]
    geom = (1, 8, 8, 8, 8, 3, 3, 8, 8)  # B cin H W OC KH KW OH OW: 3 x 3, padding 1
    assert f(None, None, None, None, None, *geom, 1, 1, 1, 1, 1, 1, 2, 3, 2, None, 0, None) == -1           # null tensors
    assert f(None, None, None, None, None, 1, 8, 8, 8, 8, 3, 3, 7, 8, 1, 1, 1, 1, 1, 1, 2, 3, 2, None, 0, None) == -1  # OH does not follow from the geometry
    assert f(None, None, None, None, None, *geom, 0, 1, 1, 1, 1, 1, 2, 3, 2, None, 0, None) == -1           # stride 0
    assert f(None, None, None, None, None, *geom, 1, 1, 1, 1, 1, 0, 2, 3, 2, None, 0, None) == -1           # dilation 0
    assert f(None, None, None, None, None, *geom, 1, 1, 1, 1, 1, 1, 2, 3, 3, None, 0, None) == -2           # int8 output
    assert f(None, None, None, None, None, 0, 8, 8, 8, 8, 3, 3, 8, 8, 1, 1, 1, 1, 1, 1, 2, 3, 2, None, 0, None) == 0   # empty batch
    # depthwise entry (r6): the same checks; OC must be a multiple of the input channels
    h = lib.quanto_hip_qbytes_conv2d_depthwise
    h.restype, h.argtypes = ci, [vp] * 5 + [i64] * 9 + [ci] * 9 + [vp]
    assert h(None, None, None, None, None, *geom, 1, 1, 1, 1, 1, 1, 2, 3, 2, None) == -1                     # null tensors
    assert h(None, None, None, None, None, 1, 8, 8, 8, 12, 3, 3, 8, 8, 1, 1, 1, 1, 1, 1, 2, 3, 2, None) == -1  # OC = 12 on 8 input channels
    assert h(None, None, None, None, None, 1, 8, 8, 8, 8, 3, 3, 7, 8, 1, 1, 1, 1, 1, 1, 2, 3, 2, None) == -1   # OH does not follow from the geometry
    assert h(None, None, None, None, None, *geom, 1, 1, 1, 1, 1, 1, 2, 3, 3, None) == -2                     # int8 output
    assert h(None, None, None, None, None, 0, 8, 8, 8, 8, 3, 3, 8, 8, 1, 1, 1, 1, 1, 1, 2, 3, 2, None) == 0    # empty batch
    g = lib.quanto_hip_qbits_conv2d
    g.restype, g.argtypes = ci, [vp] * 6 + [i64] * 9 + [ci] * 10 + [vp, sz, vp]
    assert g(None, None, None, None, None, None, *geom, 1, 1, 1, 1, 1, 1, 4, 0, 2, 2, None, 0, None) == -1  # null tensors
    assert g(None, None, None, None, None, None, *geom, 1, 1, 1, 1, 1, 1, 3, 0, 2, 2, None, 0, None) == -1  # bits
    assert g(None, None, None, None, None, None, *geom, 1, 1, 1, 1, 1, 1, 4, 50, 2, 2, None, 0, None) == -1  # K = 72 is not a multiple of the group size
    assert g(None, None, None, None, None, None, 1, 8, 8, 8, 8, 3, 3, 8, 9, 1, 1, 1, 1, 1, 1, 4, 0, 2, 2, None, 0, None) == -1  # OW
    assert g(None, None, None, None, None, None, 0, 8, 8, 8, 8, 3, 3, 8, 8, 1, 1, 1, 1, 1, 1, 4, 0, 2, 2, None, 0, None) == 0   # empty batch


def test_conv2d_k_split_plan_without_a_gpu():
    """quanto_hip_conv2d_workspace_size is the host-side statement of the convolution kernel's K split (csrc/qconv_mfma.hip: pick_split): up to
    ~2 workgroups per CU, at least 4 K-tiles per split (r5; 3 until the gather got cheaper), no split beyond 128 output tiles; one 128 x 128 fp32 tile per (split, tile)."""
    f = quanto_hip.cdll.quanto_hip_conv2d_workspace_size
    f.restype, f.argtypes = ctypes.c_int64, [ctypes.c_int64] * 5
    tile = 128 * 128 * 4
    assert f(8, 7, 7, 512, 4608) == 18 * (4 * 4) * tile        # 16 tiles, 72 K-tiles: 18 splits of 4
    assert f(8, 28, 28, 128, 1152) == 4 * 49 * tile            # 49 tiles, 18 K-tiles: 4 splits of 4 or 5
    assert f(1, 28, 28, 256, 2304) == 9 * (7 * 2) * tile       # 14 tiles, 36 K-tiles
    assert f(8, 28, 28, 256, 2304) == 5 * (49 * 2) * tile      # 98 tiles: 5 splits keep the grid under 512 workgroups
    assert f(8, 56, 56, 64, 576) == 0 and f(32, 56, 56, 64, 576) == 0   # 196 / 784 tiles: not split
    assert f(8, 32, 32, 320, 2880) == 2 * (64 * 3) * tile      # 129 .. 256 tiles: two splits once K is at least 32 K-tiles deep (45 here)
    assert f(32, 28, 28, 128, 1152) == 0                       # 196 tiles, 18 K-tiles: not split
    assert f(8, 56, 56, 256, 64) == 0                          # one K-tile
    assert f(8, 28, 28, 128, 1100) == 4 * 49 * tile            # a ragged last K-tile counts as a K-tile (r5): 18 of them, as for K = 1152
    assert f(8, 112, 112, 64, 147) == 0                        # an RGB 7x7 stem: 3 K-tiles, 784 tiles
    assert f(0, 28, 28, 128, 1152) == 0                        # empty batch
    assert f(-1, 28, 28, 128, 1152) == -1 and f(8, 28, 28, 0, 1152) == -1
    # sub-byte weights (r5): the same partial tiles plus room for the dense 16-bit weight of the row form, a multiple of 256 bytes
    g = quanto_hip.cdll.quanto_hip_qbits_conv2d_workspace_size
    g.restype, g.argtypes = ctypes.c_int64, [ctypes.c_int64] * 5
    assert g(8, 28, 28, 128, 1152) == 4 * 49 * tile + 128 * 1152 * 2
    assert g(8, 56, 56, 64, 576) == 64 * 576 * 2
    assert g(8, 15, 13, 44, 180) == f(8, 15, 13, 44, 180) + (44 * 180 * 2 + 255) // 256 * 256
    assert g(0, 28, 28, 128, 1152) == 0 and g(-1, 28, 28, 128, 1152) == -1


def test_multi_linear_plans_without_a_gpu():
    """Host logic of quanto_hip_q{bits,bytes}_mm_multi_plan: which single launch serves a group of Linears, and with what scratch."""
    lib = quanto_hip.cdll
    i64, ci = ctypes.c_int64, ctypes.c_int
    lib.quanto_hip_qbits_mm_multi_plan.restype = ci
    lib.quanto_hip_qbits_mm_multi_plan.argtypes = [ci, ctypes.POINTER(i64), i64, i64, ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(i64)]
    lib.quanto_hip_qbytes_mm_multi_plan.restype = ci
    lib.quanto_hip_qbytes_mm_multi_plan.argtypes = [ci, ctypes.POINTER(i64), i64, i64, ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(i64)]

    def qbits(Ns, M, K, group=128):
        k, ws = ci(-1), i64(-1)
        assert lib.quanto_hip_qbits_mm_multi_plan(len(Ns), (i64 * len(Ns))(*Ns), M, K, 4, group, 2, ctypes.byref(k), ctypes.byref(ws)) == 0
        return k.value, ws.value

    def qbytes(Ns, M, K, a=2, b=3):
        k, ws = ci(-1), i64(-1)
        assert lib.quanto_hip_qbytes_mm_multi_plan(len(Ns), (i64 * len(Ns))(*Ns), M, K, a, b, 2, ctypes.byref(k), ctypes.byref(ws)) == 0
        return k.value, ws.value

    qkv, gate_up = [4096, 1024, 1024], [14336, 14336]
    assert qbits(qkv, 1, 4096) == (2, 0) and qbits(gate_up, 4, 4096) == (2, 0)               # decode: one GEMV launch, no scratch
    # batched decode: one streaming launch; q/k/v = 96 feature blocks -> K split 4 ways (counter region + fp32 partials, two fragments)
    assert qbits(qkv, 32, 4096) == (5, 4096 + 96 * 4 * 256 * 2 * 16)
    assert qbits(gate_up, 32, 4096) == (5, 0)                                                 # 448 feature blocks: no split, no scratch
    assert qbits(qkv, 65, 4096)[0] == 0 and qbits([4096, 1000], 8, 4096)[0] == 0              # prefill-sized / ragged N: separate calls
    assert qbits(qkv, 8, 4096, group=64)[0] == 0                                              # other group sizes: separate calls
    assert qbytes(qkv, 1, 4096) == (2, 0) and qbytes(qkv, 2, 4096, b=5) == (2, 0)             # int8 / fp8 GEMV launch
    assert qbytes(qkv, 32, 4096) == (5, 4096 + 96 * 4 * 256 * 2 * 16) and qbytes(gate_up, 16, 4096) == (5, 0)
    assert qbytes(qkv, 32, 4096, a=3)[0] == 0                                                 # quantized activations: the native8 GEMM per member
    assert qbytes([4096, 1000], 8, 4096)[0] == 0 and qbytes([4096], 8, 4096)[0] == 0          # ragged N / a single member
    k, ws = ci(-1), i64(-1)
    assert lib.quanto_hip_qbits_mm_multi_plan(5, (i64 * 5)(64, 64, 64, 64, 64), 1, 128, 4, 128, 2, ctypes.byref(k), ctypes.byref(ws)) == -1  # > QUANTO_HIP_MAX_MULTI


def test_extension_registry_matches_reference_contract():
    # tests/library/test_extensions.py:19-39 in the reference: on ROCm the extension is called quanto_hip
    assert Q.is_extension_available("quanto_hip") == (torch.version.hip is not None)
    if torch.version.hip is not None:
        assert Q.get_extension("quanto_hip") is quanto_hip
    assert not Q.is_extension_available("quanto_cuda")


def test_device_ops_have_no_cpu_fallback():
    """A CPU tensor never reaches the HIP binding, and the binding refuses CPU tensors."""
    with pytest.raises(Q.QuantoHipError):
        quanto_hip.lib.unpack(torch.zeros(16, dtype=torch.uint8), 4)


@pytest.mark.parametrize("bits", [2, 4])
def test_pack_unpack_golden(golden, bits):
    cases = sorted({k[: k.rfind("/")] for k in golden if k.startswith(f"pack/b{bits}/")})
    assert cases
    for c in cases:
        a = torch.from_numpy(golden[c + "/a"])
        packed = pack_weights(a, bits)
        assert np.array_equal(packed.numpy(), golden[c + "/packed"])
        assert np.array_equal(torch.ops.quanto.unpack(packed, bits).numpy(), golden[c + "/unpacked"])
        pt = PackedTensor.pack(a, bits)
        assert torch.equal(pt.unpack(), a)


def test_packed_tensor_serialization():
    t = torch.randint(0, 16, (10, 32), dtype=torch.uint8)
    packed = PackedTensor.pack(t, 4)
    buf = io.BytesIO()
    torch.save(packed, buf)
    buf.seek(0)
    again = torch.load(buf, weights_only=False)
    assert isinstance(again, PackedTensor) and again.bits == 4 and again.shape == packed.shape
    assert torch.equal(again._data, packed._data) and torch.equal(again.unpack(), t)



def _assert_reference_cpu_output(y, here, golden_y, dt):
    """`y` came out of this package's CPU path, `here` out of the torch CPU ops the reference runs for the same call, restated in
    the test and executed on THIS machine: bit for bit.  Against the output the reference produced where the fixture was
    generated: equal on the same CPU family; on another ISA torch's CPU GEMM blocks / accumulates the K sum differently (the
    16-bit GEMMs of oneDNN in particular), which moves an output by an ulp or two of its dtype - so that comparison is made in
    units of the last place."""
    assert torch.equal(y, here)
    if np.array_equal(to_numpy(y), golden_y):
        return
    if dt == "fp32":
        assert O.rel_max(to_numpy(y), golden_y) < 1e-5
    else:
        ulps = O.ulp_distance(to_numpy(y), golden_y, dt)
        assert ulps.max() <= 2 and (ulps > 0).mean() < 0.2, (int(ulps.max()), float((ulps > 0).mean()))


QBITS = ["int4_g128_fp32", "int4_g128_fp16", "int4_g128_bf16", "int4_g128_fp16_zp", "int4_g64_fp32",
         "int4_perchannel_fp32", "int4_oddrows_fp32", "int2_g128_fp32", "int2_g128_bf16", "int4_g128_bf16_small_w"]


def _dt_of(tag):
    return next(d for d in ("fp32", "fp16", "bf16") if d in tag)


@pytest.mark.parametrize("tag", QBITS)
def test_quantize_weight_qbits_golden(golden, tag):
    k, dt = f"qbits/{tag}", _dt_of(tag)
    N, K, bits, gs, zp = [int(v) for v in golden[k + "/meta"]]
    gs = gs or None
    qt = Q.qint4 if bits == 4 else Q.qint2
    w = to_torch(golden[k + "/w"], dt)
    scale, shift = Q.MaxOptimizer()(w, qtype=qt, axis=0, group_size=gs, zeropoint=bool(zp))
    qw = Q.quantize_weight(w, qtype=qt, axis=0, scale=scale, shift=shift, group_size=gs)
    assert isinstance(qw, Q.WeightQBitsTensor) and qw.dtype == TORCH_DT[dt]
    assert np.array_equal(to_numpy(qw._scale), golden[k + "/scale"])
    assert np.array_equal(to_numpy(qw._shift), golden[k + "/shift"])
    assert np.array_equal(qw._data._data.numpy(), golden[k + "/packed"])
    assert np.array_equal(to_numpy(qw.dequantize()), golden[k + "/dequantized"])
    for key in [x for x in golden if x.startswith(k + "/x")]:
        M = key.rsplit("/x", 1)[1]
        x = to_torch(golden[key], dt)
        y = torch.nn.functional.linear(x, qw)
        # tensor/weights/qbits.py:262-287 off-device: dequantize (bit-equal to the reference's, asserted above) + torch's linear
        _assert_reference_cpu_output(y, torch.nn.functional.linear(x, qw.dequantize()), golden[k + f"/y{M}"], dt)


QBYTES = ["int8_fp32", "int8_fp16", "int8_bf16", "e4m3fn_fp32", "e4m3fn_fp16", "e4m3fn_bf16", "e4m3fnuz_fp16", "e5m2_fp16",
          "cfg1_int8_fp32_1x1024x1024"]


@pytest.mark.parametrize("tag", QBYTES)
def test_quantize_weight_qbytes_golden(golden, tag):
    k, dt = f"qbytes/{tag}", _dt_of(tag)
    qt = {"int8": Q.qint8, "e4m3fn": Q.qfloat8_e4m3fn, "e4m3fnuz": Q.qfloat8_e4m3fnuz, "e5m2": Q.qfloat8_e5m2}[
        "int8" if "int8" in tag else tag.split("_")[0]]
    if k + "/w" in golden:
        w = to_torch(golden[k + "/w"], dt)
        scale = Q.AbsmaxOptimizer()(w, qtype=qt, axis=0)
        qw = Q.quantize_weight(w, qtype=qt, axis=0, scale=scale)
        assert np.array_equal(to_numpy(qw._scale), golden[k + "/scale"])
        assert np.array_equal(to_numpy(qw._data), golden[k + "/data"])
    else:  # cfg1: the 1024x1024 float weight is not stored; rebuild the QTensor from its golden integers
        data = torch.from_numpy(golden[k + "/data"])
        scale = to_torch(golden[k + "/scale"], dt)
        qw = Q.WeightQBytesTensor(qt, 0, data.shape, data.stride(), data, scale, None)
    if k + "/dequantized" in golden:
        assert np.array_equal(to_numpy(qw.dequantize()), golden[k + "/dequantized"])
    for key in [x for x in golden if x.startswith(k + "/x")]:
        M = key.rsplit("/x", 1)[1]
        x = to_torch(golden[key], dt)
        y = torch.nn.functional.linear(x, qw)
        # library/qbytes_mm.py:91-105 on the CPU: bf16 x int8 -> _weight_int8pack_mm, everything else (scale * weight) then matmul
        if dt == "bf16" and qt == Q.qint8 and x.shape[-1] % 4 == 0:
            here = torch._weight_int8pack_mm(x.reshape(-1, x.shape[-1]), qw._data, qw._scale.flatten()).reshape(y.shape)
        else:
            wf = qw._data.to(qw._scale.dtype) if qw._data.dtype.is_floating_point else qw._data
            here = torch.matmul(x, (qw._scale * wf).t())
        _assert_reference_cpu_output(y, here, golden[k + f"/y{M}"], dt)


def test_qlinear_quantize_freeze_state_dict_roundtrip():
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(256, 128), torch.nn.ReLU(), torch.nn.Linear(128, 64, bias=False))
    x = torch.randn(4, 256)
    Q.quantize(model, weights=Q.qint4, exclude="2")
    assert isinstance(model[0], Q.QLinear) and not isinstance(model[2], Q.QLinear)
    assert model[0].weight_group_size == 128 and not model[0].frozen
    y_dyn = model(x)
    Q.freeze(model)
    assert model[0].frozen and isinstance(model[0].weight, Q.WeightQBitsTensor)
    y = model(x)
    assert torch.equal(y, y_dyn)  # dynamic and frozen quantization agree
    qmap = Q.quantization_map(model)
    assert qmap == {"0": {"weights": "qint4", "activations": "none"}}
    sd = model.state_dict()
    assert {"0.weight._data._data", "0.weight._scale", "0.weight._shift", "0.bias"} <= set(sd)
    fresh = torch.nn.Sequential(torch.nn.Linear(256, 128), torch.nn.ReLU(), torch.nn.Linear(128, 64, bias=False))
    Q.requantize(fresh, sd, qmap)
    assert torch.equal(fresh(x), y)
    # from the meta device (how a large checkpoint is opened): the float weight of a quantized module is never materialised
    # - it goes from meta straight to the quantized tensor -, and a bf16 model keeps its dtype for what is not quantized
    allocated = []
    real_empty_like = torch.empty_like

    def spy(t, *a, **kw):
        out = real_empty_like(t, *a, **kw)
        if out.device.type != "meta":
            allocated.append(tuple(t.shape))
        return out

    with torch.device("meta"):
        lazy = torch.nn.Sequential(torch.nn.Linear(256, 128), torch.nn.ReLU(), torch.nn.Linear(128, 64, bias=False)).to(torch.bfloat16)
    torch.empty_like = spy
    try:
        with pytest.warns(UserWarning, match="cast to the model dtype"):  # fp32 checkpoint, bf16 skeleton: announced, not silent
            Q.requantize(lazy, sd, qmap, device=torch.device("cpu"))
    finally:
        torch.empty_like = real_empty_like
    assert (128, 256) not in allocated and (64, 128) in allocated, allocated
    assert isinstance(lazy[0].weight, Q.WeightQBitsTensor) and lazy[2].weight.dtype == torch.bfloat16 and lazy[0].bias.dtype == torch.bfloat16
    assert not any(p.device.type == "meta" for p in lazy.parameters())
    # the checkpoint above is fp32, the model bf16: the rebuilt weight's scale / shift follow the MODEL's dtype (as its bias
    # does), and the module runs - r2 left scale fp32 next to a bf16 bias and forward raised a dtype mismatch
    assert lazy[0].weight._scale.dtype == torch.bfloat16 and lazy[0].weight._shift.dtype == torch.bfloat16
    assert lazy[0].weight.dtype == torch.bfloat16
    assert torch.equal(lazy[0].weight._data._data, model[0].weight._data._data)  # integers untouched
    out = lazy(x.to(torch.bfloat16))
    assert out.dtype == torch.bfloat16 and (out.float() - y).abs().max() < 0.05 * y.abs().max()


def test_requantize_says_what_it_could_not_rebuild():
    """A reference checkpoint may quantize module types this package has no counterpart for (the reference's QLayerNorm) and carry
    their input / output scales: requantize must not drop them silently."""
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(64, 32), torch.nn.LayerNorm(32))
    Q.quantize(model, weights=Q.qint8, exclude="1")
    Q.freeze(model)
    sd, qmap = model.state_dict(), Q.quantization_map(model)
    qmap["1"] = {"weights": "none", "activations": "qint8"}  # what the reference writes for a quantized LayerNorm ...
    sd["1.input_scale"], sd["1.output_scale"] = torch.ones(()), torch.ones(())  # ... with its calibrated scales
    qmap["ghost"] = {"weights": "qint8", "activations": "none"}
    fresh = torch.nn.Sequential(torch.nn.Linear(64, 32), torch.nn.LayerNorm(32))
    with pytest.warns(UserWarning) as rec:
        Q.requantize(fresh, sd, qmap)
    text = " ".join(str(w.message) for w in rec)
    assert "1 (LayerNorm)" in text and "ghost" in text and "1.input_scale" in text
    assert isinstance(fresh[0], Q.QLinear) and torch.equal(fresh[0].weight._data, model[0].weight._data)


def test_conv2d_geometry_gate_mirrors_the_kernel_limits():
    """library/hip.py asks conv2d_geometry_ok before routing a QConv2d to the implicit-GEMM kernels; the limits are the kernel's
    (31-bit offsets, one grid dimension of 65535 tiles of 128 pixels): beyond them the call keeps the im2col / reference path
    instead of surfacing ENOTSUP as an error."""
    from optimum_quanto_amd.library.hip import _Bindings

    ok = _Bindings.conv2d_geometry_ok
    one = ((1, 1), (0, 0), (1, 1))
    assert ok((8, 128, 28, 28), (128, 128, 3, 3), (1, 1), (1, 1), (1, 1))
    assert ok((8, 3, 224, 224), (64, 3, 7, 7), (2, 2), (3, 3), (1, 1))          # K = 147: a ragged last K-tile (r5)
    assert ok((1, 64, 32, 32), (64, 64, 9, 9), *one)                            # 81 taps: two mask words (r5)
    assert ok((1, 8, 32, 32), (8, 8, 11, 11), *one) and not ok((1, 8, 32, 32), (8, 8, 12, 12), *one)  # 121 taps / 144 taps
    assert ok((8, 64, 1024, 1023), (64, 64, 1, 1), *one)                        # 65472 tiles of 128 pixels ...
    assert not ok((8, 64, 1024, 1024), (64, 64, 1, 1), *one)                    # ... 65536: grid.y
    assert not ok((16, 64, 1024, 1024), (64, 64, 1, 1), (8, 8), (0, 0), (1, 1))  # 2^30 input elements
    assert not ok((1, 64, 8, 8), (64, 64, 3, 3), (0, 1), (1, 1), (1, 1))         # stride 0
    assert not ok((1, 64, 2, 2), (64, 64, 3, 3), *one)                          # empty output


def test_conv_pair_normalisation():
    from optimum_quanto_amd.tensor.weights import _pair

    assert _pair(2) == [2, 2] and _pair((2,)) == [2, 2] and _pair([1, 3]) == [1, 3] and _pair(torch.Size([4])) == [4, 4]


def test_group_size_selection_follows_reference():
    from optimum_quanto_amd.nn.module import select_group_size
    assert [select_group_size(k) for k in (4096, 11008, 14336, 160, 96, 128, 200, 192)] == [128, 128, 128, 32, None, None, None, 96]


def test_quantize_weight_argument_errors():
    w = torch.randn(8, 8)
    with pytest.raises(ValueError):
        Q.quantize_weight(w, Q.qint8, axis=1, scale=torch.ones(8, 1))
    with pytest.raises(ValueError):
        Q.quantize_weight(w, Q.qint8, axis=0, scale=torch.ones(8, 1), shift=torch.zeros(8, 1))
    with pytest.raises(ValueError):
        Q.quantize_weight(w, Q.qint4, axis=0, scale=torch.ones(8, 1))
    with pytest.raises(ValueError):
        torch.ops.quanto.quantize_symmetric(torch.randn(8), torch.int8, 0, torch.ones(8))


@pytest.mark.parametrize("kind", ["qbits_i4", "qbytes_i8", "qbytes_f8", "qbytes_i8i8", "qbytes_f8f8"])
def test_bench_inputs_are_quantizer_outputs(kind):
    """bench.py builds its synthetic operands with the reference quantizer's arithmetic, on the device the
    bench runs on; here on the CPU: the dequantized weight must reproduce the bf16 weight it came from to within half a
    quantization step, int4 codes must span 0..15 per group, int8 rows must reach +-127."""
    import bench
    from oracle import quanto_oracle as O

    M, K, N = 8, 256, 64
    x, sets = bench.build_inputs(kind, M, K, N, torch.device("cpu"), 2, seed=3)
    assert len(sets) == 2 and tuple(x.shape) == (M, K)
    if kind == "qbits_i4":
        packed, scale, shift = sets[0]
        assert tuple(packed.shape) == (N * K // 256, 128) and tuple(scale.shape) == (N * K // 128, 1) == tuple(shift.shape)
        q = O.unpacked_rows(to_numpy(packed), 4, N * K // 128)
        assert q.min() == 0 and q.max() == 15
        assert (q.min(axis=1) == 0).all() and (q.max(axis=1) == 15).all()  # max-min affine: every group spans the range
    else:
        q, scale = sets[0]
        assert tuple(q.shape) == (N, K) and tuple(scale.shape) == (N, 1) and scale.dtype == torch.bfloat16
        if q.dtype == torch.int8:
            a = q.to(torch.int32).abs().amax(dim=1)
            assert int(a.max()) <= 127 and int(a.min()) >= 126  # absmax scaling (scale rounded to bf16: 126 or 127)
        else:
            assert q.dtype == torch.float8_e4m3fn and float(q.float().abs().max()) <= 448.0
    if kind == "qbytes_i8i8":
        assert x.dtype == torch.int8 and int(x.to(torch.int32).abs().max()) >= 126
    elif kind == "qbytes_f8f8":
        assert x.dtype == torch.float8_e4m3fn
    else:
        assert x.dtype == torch.bfloat16
