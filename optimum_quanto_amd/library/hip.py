"""``quanto_hip``: ctypes binding of ``libquanto_hip.so`` (the MI355X backend, ``include/quanto_hip.h``).

This is the counterpart of the reference's ``library/extensions/hip/__init__.py:18-36`` (which binds a
single ``unpack`` kernel through pybind11): it exposes ``ext.lib.unpack(t, bits)`` with the same call
shape plus the fused products the reference lacks on ROCm.  The wrappers take torch tensors, allocate the
output with the input's options and return it by value - the reference's C++ convention
(library/extensions/cuda/unpack.cu:26-56) - and launch on torch's *current* stream under a device guard.
"""
import ctypes
import os
import re

import torch

from .extension import NativeLibrary, register_extension

__all__ = ["quanto_hip", "QuantoHipError"]

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# quanto_hip_dtype (include/quanto_hip.h)
F32, F16, BF16, I8, U8, F8_E4M3FN, F8_E5M2, F8_E4M3FNUZ = range(8)
WS_COUNTER_BYTES = 4096  # QUANTO_HIP_WS_COUNTER_BYTES
def _c_atoi(v: str) -> int:
    """The value C's ``atoi`` reads from an environment string (csrc/qh_common.h parses the knobs with it): optional blanks, a sign, leading
    digits; anything else is 0 - so that "01", "1 " or "true" mean the same thing on both sides of the binding."""
    m = re.match(r"\s*([+-]?\d+)", v or "")
    return int(m.group(1)) if m else 0


_EXPERIMENT = _c_atoi(os.environ.get("QUANTO_HIP_EXPERIMENT", "0")) != 0  # the library's knobs are live: plans are not cached
KERNEL_AUTO, KERNEL_NAIVE, KERNEL_GEMV, KERNEL_MFMA, KERNEL_MFMA_LARGE, KERNEL_SKINNY, KERNEL_NATIVE8, KERNEL_DEQUANT_MFMA, KERNEL_MFMA_FUSED4, KERNEL_MMV, KERNEL_MFMA_LARGE4 = range(11)
KERNELS = {"auto": KERNEL_AUTO, "naive": KERNEL_NAIVE, "gemv": KERNEL_GEMV, "mfma": KERNEL_MFMA, "mfma_large": KERNEL_MFMA_LARGE, "skinny": KERNEL_SKINNY,
           "mfma_native8": KERNEL_NATIVE8, "dequant_mfma": KERNEL_DEQUANT_MFMA, "mfma_fused4": KERNEL_MFMA_FUSED4, "mmv": KERNEL_MMV, "mfma_large4": KERNEL_MFMA_LARGE4}

_DTYPES = {
    torch.float32: F32,
    torch.float16: F16,
    torch.bfloat16: BF16,
    torch.int8: I8,
    torch.uint8: U8,
    torch.float8_e4m3fn: F8_E4M3FN,
    torch.float8_e5m2: F8_E5M2,
    torch.float8_e4m3fnuz: F8_E4M3FNUZ,
}


class QuantoHipError(RuntimeError):
    pass


def _dt(t: torch.Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise QuantoHipError(f"quanto_hip: unsupported dtype {t.dtype}") from None


def _ptr(t):
    """Device address as a plain int (ctypes converts it through the entry's argtypes; building a c_void_p object per argument costs more
    than the rest of the marshalling)."""
    return 0 if t is None else t.data_ptr()


try:  # the stream handle without constructing a torch.cuda.Stream per call (what torch's own inductor / triton launchers use)
    _raw_stream = torch._C._cuda_getCurrentRawStream
    _current_device = torch._C._cuda_getDevice
except AttributeError:  # pragma: no cover - other torch builds
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream

    def _current_device():
        return torch.cuda.current_device()


class _DeviceGuard:
    """``with torch.cuda.device(d)`` only when ``d`` is not already current: the context manager costs ~3 us per call, the check 0.2."""

    __slots__ = ("ctx",)

    def __init__(self, device: torch.device):
        self.ctx = None if device.index is None or device.index == _current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


class _Bindings:
    """Typed entry points; one instance per loaded library."""

    def __init__(self, cdll: ctypes.CDLL):
        c = cdll
        vp, i64, ci, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
        c.quanto_hip_abi_version.restype = ci
        c.quanto_hip_status_string.restype = ctypes.c_char_p
        c.quanto_hip_status_string.argtypes = [ci]
        c.quanto_hip_last_kernel.restype = ctypes.c_char_p
        c.quanto_hip_stream_capture_id.restype = i64
        c.quanto_hip_stream_capture_id.argtypes = [vp]
        c.quanto_hip_unpack.restype = ci
        c.quanto_hip_unpack.argtypes = [vp, vp, i64, ci, vp]
        c.quanto_hip_dequantize_qbits.restype = ci
        c.quanto_hip_dequantize_qbits.argtypes = [vp, vp, vp, vp, i64, i64, ci, ci, ci, ci, vp]
        c.quanto_hip_qbits_mm.restype = ci
        c.quanto_hip_qbits_mm.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, ci, ci, ci, ci, ci, vp, sz, vp]
        c.quanto_hip_qbits_mm_multi.restype = ci
        c.quanto_hip_qbits_mm_multi.argtypes = [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                                ctypes.POINTER(vp), ctypes.POINTER(i64), i64, i64, ci, ci, ci, ci, vp]
        c.quanto_hip_qbits_mm_multi_ws.restype = ci
        c.quanto_hip_qbits_mm_multi_ws.argtypes = [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                                   ctypes.POINTER(vp), ctypes.POINTER(i64), i64, i64, ci, ci, ci, ci, vp, sz, vp]
        c.quanto_hip_qbits_mm_multi_plan.restype = ci
        c.quanto_hip_qbits_mm_multi_plan.argtypes = [ci, ctypes.POINTER(i64), i64, i64, ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(i64)]
        c.quanto_hip_qbytes_mm_multi_ws.restype = ci
        c.quanto_hip_qbytes_mm_multi_ws.argtypes = [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                                    ctypes.POINTER(i64), i64, i64, ci, ci, ci, vp, sz, vp]
        c.quanto_hip_qbytes_mm_multi_plan.restype = ci
        c.quanto_hip_qbytes_mm_multi_plan.argtypes = [ci, ctypes.POINTER(i64), i64, i64, ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(i64)]
        c.quanto_hip_qbits_mm_workspace_size.restype = i64
        c.quanto_hip_qbits_mm_workspace_size.argtypes = [i64, i64, i64, ci, ci, ci, ci]
        c.quanto_hip_qbits_mm_plan.restype = ci
        c.quanto_hip_qbits_mm_plan.argtypes = [i64, i64, i64, ci, ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(i64)]
        c.quanto_hip_qbytes_mm_plan.restype = ci
        c.quanto_hip_qbytes_mm_plan.argtypes = [i64, i64, i64, ci, ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(i64)]
        c.quanto_hip_qbits_mm_pick.restype = ci
        c.quanto_hip_qbits_mm_pick.argtypes = [i64, i64, i64, ci, ci, ci]
        c.quanto_hip_qbytes_mm.restype = ci
        c.quanto_hip_qbytes_mm.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, ci, ci, ci, ci, vp]
        c.quanto_hip_qbits_mm_a8.restype = ci
        c.quanto_hip_qbits_mm_a8.argtypes = [vp] * 7 + [i64] * 3 + [ci] * 5 + [vp, sz, vp]
        c.quanto_hip_qbits_mm_a8_workspace_size.restype = i64
        c.quanto_hip_qbits_mm_a8_workspace_size.argtypes = [i64] * 3 + [ci] * 4
        c.quanto_hip_qbytes_mm_ws.restype = ci
        c.quanto_hip_qbytes_mm_ws.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, ci, ci, ci, ci, vp, sz, vp]
        c.quanto_hip_qbytes_mm_workspace_size.restype = i64
        c.quanto_hip_qbytes_mm_workspace_size.argtypes = [i64, i64, i64, ci, ci, ci, ci]
        c.quanto_hip_qbytes_mm_pick.restype = ci
        c.quanto_hip_qbytes_mm_pick.argtypes = [i64, i64, i64, ci, ci, ci]
        c.quanto_hip_quantize_symmetric.restype = ci
        c.quanto_hip_quantize_symmetric.argtypes = [vp, vp, vp, i64, i64, ci, ci, ci, vp]
        c.quanto_hip_quantize_affine.restype = ci
        c.quanto_hip_quantize_affine.argtypes = [vp, vp, vp, vp, i64, i64, ci, ci, ci, ci, vp]
        c.quanto_hip_quantize_affine_packed.restype = ci
        c.quanto_hip_quantize_affine_packed.argtypes = [vp, vp, vp, vp, i64, i64, ci, ci, ci, ci, vp]
        c.quanto_hip_pack.restype = ci
        c.quanto_hip_pack.argtypes = [vp, vp, i64, i64, ci, vp]
        c.quanto_hip_qbytes_conv2d.restype = ci
        c.quanto_hip_qbytes_conv2d.argtypes = [vp, vp, vp, vp, vp] + [i64] * 9 + [ci] * 9 + [vp, ctypes.c_size_t, vp]
        c.quanto_hip_dequantize_symmetric.restype = ci
        c.quanto_hip_dequantize_symmetric.argtypes = [vp, vp, vp, i64, ci, ci, vp]
        c.quanto_hip_qbytes_conv2d_depthwise.restype = ci
        c.quanto_hip_qbytes_conv2d_depthwise.argtypes = [vp, vp, vp, vp, vp] + [i64] * 9 + [ci] * 9 + [vp]
        c.quanto_hip_conv2d_workspace_size.restype = i64
        c.quanto_hip_conv2d_workspace_size.argtypes = [i64] * 5
        c.quanto_hip_qbits_conv2d_workspace_size.restype = i64
        c.quanto_hip_qbits_conv2d_workspace_size.argtypes = [i64] * 5
        c.quanto_hip_qbits_conv2d_workspace_size_geom.restype = i64
        c.quanto_hip_qbits_conv2d_workspace_size_geom.argtypes = [i64] * 8 + [ci] * 2
        c.quanto_hip_qbits_conv2d.restype = ci
        c.quanto_hip_qbits_conv2d.argtypes = [vp] * 6 + [i64] * 9 + [ci] * 10 + [vp, ctypes.c_size_t, vp]
        self._c = c
        if c.quanto_hip_abi_version() != 1:
            raise QuantoHipError("libquanto_hip.so ABI version mismatch: rebuild with __graft_entry__.build()")

    # -- helpers ----------------------------------------------------------------------------------
    def _check(self, status: int, what: str):
        if status != 0:
            msg = self._c.quanto_hip_status_string(status).decode()
            raise QuantoHipError(f"quanto_hip.{what} failed: {msg} (status {status})")

    @staticmethod
    def _stream(t: torch.Tensor):
        return ctypes.c_void_p(_raw_stream(t.device.index if t.device.index is not None else _current_device()))

    @staticmethod
    def _require_cuda(*tensors):
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise QuantoHipError("quanto_hip kernels only accept tensors on a ROCm device")

    def _zeroed_workspace(self, device: torch.device, nbytes: int, stream) -> torch.Tensor:
        """Split-K workspace: [QUANTO_HIP_WS_COUNTER_BYTES of arrival counters | fp32 partial sums] (include/quanto_hip.h).
        One buffer per (device, stream, capture): its counter region zero-filled once when (re)allocated and only ever handed to kernels that
        restore the counter words they use.  Launches on one stream reuse it in stream order; launches on different streams
        of one device may overlap, so each stream gets its own counters.  A buffer allocated while the stream is being
        captured lives in that graph's memory pool and its zero-fill is a node of that graph (re-run on every replay): it is
        keyed by the capture id so that neither eager launches nor another capture ever see it."""
        cache = self.__dict__.setdefault("_zero_ws", {})
        capture = self._c.quanto_hip_stream_capture_id(ctypes.c_void_p(stream))
        if capture < 0:
            self._check(int(capture), "stream_capture_id")
        key = (device, stream, capture)
        buf = cache.get(key)
        if buf is None or buf.numel() < nbytes:
            if len(cache) > 64:  # stream handles / capture ids come and go: do not keep dead buffers alive forever
                for k in [k for k in cache if k[2] != 0 and k != key]:
                    del cache[k]
            # only the counter region has to be zero (include/quanto_hip.h: QUANTO_HIP_WS_COUNTER_BYTES; the kernels restore what they
            # use, the partial sums behind it are never read before they are written): 4 KiB of fill - under capture a 4 KiB memset
            # node per replay instead of one over the whole buffer (8-17 MB)
            buf = torch.empty((max(nbytes, 8 << 20),), dtype=torch.uint8, device=device)
            buf[:WS_COUNTER_BYTES].zero_()
            cache[key] = buf
        return buf

    def _scratch(self, device: torch.device, nbytes: int, stream) -> torch.Tensor:
        """Uninitialised scratch (the dequantized weight of the prefill path, the row sums of the 128x128 kernel): one growing
        buffer per (device, stream, capture) instead of an allocation per call; calls on one stream use it in stream order."""
        cache = self.__dict__.setdefault("_scratch_ws", {})
        capture = self._c.quanto_hip_stream_capture_id(ctypes.c_void_p(stream))
        if capture < 0:
            self._check(int(capture), "stream_capture_id")
        key = (device, stream, capture)
        buf = cache.get(key)
        if buf is None or buf.numel() < nbytes:
            if len(cache) > 64:
                for k in [k for k in cache if k[2] != 0 and k != key]:
                    del cache[k]
            buf = cache[key] = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        return buf

    def last_kernel(self) -> str:
        return self._c.quanto_hip_last_kernel().decode()

    # -- quanto::quantize_symmetric -----------------------------------------------------------------
    QUANTIZE_TARGETS = (torch.int8, torch.float8_e4m3fn, torch.float8_e5m2)

    def quantize_symmetric(self, base: torch.Tensor, dtype: torch.dtype, axis, scale: torch.Tensor) -> torch.Tensor:
        """One-pass clamp(round(base / scale)).to(dtype); ``axis`` in (None, 0, -1) as validated by the op wrapper."""
        self._require_cuda(base, scale)
        if dtype not in self.QUANTIZE_TARGETS or base.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            raise QuantoHipError(f"quantize_symmetric: unsupported dtypes {base.dtype} -> {dtype}")
        base = base.contiguous()
        scale = scale.to(base.dtype).contiguous()
        out = torch.empty(base.shape, dtype=dtype, device=base.device)
        if axis is None:
            mode, inner = 0, 1
        elif axis == 0:
            mode, inner = 1, (base.numel() // base.shape[0] if base.numel() else 1)
        else:
            mode, inner = 2, base.shape[-1]
        with torch.cuda.device(base.device):
            st = self._c.quanto_hip_quantize_symmetric(_ptr(base), _ptr(scale), _ptr(out), base.numel(), inner, mode, _dt(base),
                                                       _dt(out), self._stream(base))
        self._check(st, "quantize_symmetric")
        return out

    def dequantize_symmetric(self, data: torch.Tensor, scale: torch.Tensor):
        """``scale * data.to(scale.dtype)`` for a per-tensor scale in one pass (tensor/qbytes.py:23-36); None when the view is not one the kernel takes (the
        caller keeps the two-kernel expression)."""
        if not (data.is_cuda and scale.is_cuda and scale.numel() == 1 and data.is_contiguous() and data.dtype in (torch.int8, torch.float8_e4m3fn, torch.float8_e5m2)
                and scale.dtype in (torch.float32, torch.float16, torch.bfloat16)):
            return None
        out = torch.empty(data.shape, dtype=scale.dtype, device=data.device)
        if data.numel() == 0:
            return out
        if (data.data_ptr() | out.data_ptr()) % 16:
            return None
        with torch.cuda.device(data.device):
            st = self._c.quanto_hip_dequantize_symmetric(_ptr(data), _ptr(scale), _ptr(out), data.numel(), _dt(data), _dt(out), self._stream(data))
        self._check(st, "dequantize_symmetric")
        return out

    def quantize_affine(self, base: torch.Tensor, bits: int, group_size, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
        """Axis-0 2-D weights only: uint8 grouped matrix [N*K/C, C] (C = group_size or K)."""
        self._require_cuda(base, scale, shift)
        N, K = base.shape
        base = base.contiguous()
        scale = scale.to(base.dtype).contiguous()
        shift = shift.contiguous() if not shift.dtype.is_floating_point else shift.to(base.dtype).contiguous()
        C = group_size or K
        out = torch.empty((N * K // C, C), dtype=torch.uint8, device=base.device)
        with torch.cuda.device(base.device):
            st = self._c.quanto_hip_quantize_affine(_ptr(base), _ptr(scale), _ptr(shift), _ptr(out), N, K, bits, group_size or 0,
                                                    _dt(base), _dt(shift), self._stream(base))
        self._check(st, "quantize_affine")
        return out

    def quantize_affine_packed(self, base: torch.Tensor, bits: int, group_size, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
        """quantize_affine + pack_weights in one pass: the packed bytes [ceil(R / (8/bits)), C] of the grouped matrix [R, C]."""
        self._require_cuda(base, scale, shift)
        N = base.shape[0]
        K = base.numel() // N
        base = base.contiguous()
        scale = scale.to(base.dtype).contiguous()
        shift = shift.contiguous() if not shift.dtype.is_floating_point else shift.to(base.dtype).contiguous()
        C = group_size or K
        rows = N * K // C
        vpi = 8 // bits
        out = torch.empty(((rows + vpi - 1) // vpi, C), dtype=torch.uint8, device=base.device)
        with torch.cuda.device(base.device):
            st = self._c.quanto_hip_quantize_affine_packed(_ptr(base), _ptr(scale), _ptr(shift), _ptr(out), N, K, bits,
                                                           group_size or 0, _dt(base), _dt(shift), self._stream(base))
        self._check(st, "quantize_affine_packed")
        return out

    def pack(self, t: torch.Tensor, bits: int) -> torch.Tensor:
        self._require_cuda(t)
        if t.dtype not in (torch.uint8, torch.int8):
            raise QuantoHipError("pack expects an 8-bit integer tensor")
        t = t.contiguous().view(torch.uint8)
        rows = t.shape[0]
        cols = t.numel() // rows if rows else 0
        row_dim = (rows + 8 // bits - 1) // (8 // bits)
        out = torch.empty((row_dim,) + tuple(t.shape[1:]), dtype=torch.uint8, device=t.device)
        with torch.cuda.device(t.device):
            self._check(self._c.quanto_hip_pack(_ptr(t), _ptr(out), rows, cols, bits, self._stream(t)), "pack")
        return out

    # -- quanto::qbytes_conv2d (implicit GEMM) -----------------------------------------------------------
    @staticmethod
    def conv2d_out_size(size, k, stride, pad, dil):
        return (size + 2 * pad - dil * (k - 1) - 1) // stride + 1

    def _conv2d_scratch(self, x, B, OH, OW, OC, K, geom=None):
        """(buffer, bytes) for the convolution kernels' K split - plain scratch, nothing to zero; (None, 0) when the problem needs none.  ``geom`` =
        (cin, W, KH, KW, stride_w, dil_w) of a sub-byte weight: plus the dense weight of the row form, when THIS geometry can take it."""
        if geom is None:
            nbytes = int(self._c.quanto_hip_conv2d_workspace_size(B, max(OH, 0), max(OW, 0), OC, K))
        else:
            cin, W, KH, KW, sw, dw = geom
            nbytes = int(self._c.quanto_hip_qbits_conv2d_workspace_size_geom(B, cin, W, OC, KH, KW, max(OH, 0), max(OW, 0), sw, dw))
        if nbytes <= 0:
            return None, 0
        return self._scratch(x.device, nbytes, self._stream(x).value), nbytes

    @classmethod
    def conv2d_geometry_ok(cls, x_shape, w_shape, stride, padding, dilation) -> bool:
        """Python mirror of ``conv::geometry_ok`` (csrc/qconv_mfma.hip): what the implicit-GEMM kernels index with 31-bit offsets and one grid
        dimension.  Beyond these limits the C entry returns ENOTSUP; the callers ask here first and keep the im2col / reference path instead."""
        B, C, H, W = x_shape
        OC, _, KH, KW = w_shape
        if min(stride) <= 0 or min(dilation) <= 0 or min(padding) < 0:
            return False
        OH = cls.conv2d_out_size(H, KH, stride[0], padding[0], dilation[0])
        OW = cls.conv2d_out_size(W, KW, stride[1], padding[1], dilation[1])
        K = C * KH * KW
        return (B >= 1 and OH >= 1 and OW >= 1 and K >= 1 and K < (1 << 24) and KH * KW <= 127 and B * C * H * W < (1 << 30) and B * OC * OH * OW < (1 << 31)
                and OC * K < (1 << 31) and (B * OH * OW + 127) // 128 <= 65535)

    def qbytes_conv2d_supported(self, x, w, stride=(1, 1), padding=(0, 0), dilation=(1, 1)) -> bool:
        """What the kernel takes: NCHW 16-bit activations, an 8-bit OCP weight, and a geometry within ``conv2d_geometry_ok`` (windows of up
        to 127 taps, 31-bit offsets; any C * KH * KW: the last K-tile may be ragged, any weight alignment)."""
        return (x.is_cuda and x.dim() == 4 and w.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16) and
                w.dtype in (torch.int8, torch.float8_e4m3fn, torch.float8_e5m2) and
                self.conv2d_geometry_ok(tuple(x.shape), tuple(w.shape), stride, padding, dilation))

    def qbytes_conv2d_depthwise_supported(self, x, w, stride=(1, 1), padding=(0, 0), dilation=(1, 1)) -> bool:
        """Depthwise layers (r6): weight [OC, 1, KH, KW] on an input of C > 1 channels with OC a multiple of C; NCHW 16-bit activations, 8-bit OCP weight."""
        if not (x.is_cuda and x.dim() == 4 and w.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16) and
                w.dtype in (torch.int8, torch.float8_e4m3fn, torch.float8_e5m2)):
            return False
        B, C, H, W = x.shape
        OC, wc, KH, KW = w.shape
        if wc != 1 or C < 2 or OC % C != 0 or min(stride) <= 0 or min(dilation) <= 0 or min(padding) < 0:
            return False
        OH = self.conv2d_out_size(H, KH, stride[0], padding[0], dilation[0])
        OW = self.conv2d_out_size(W, KW, stride[1], padding[1], dilation[1])
        return (B >= 1 and OH >= 1 and OW >= 1 and B * C * H * W < (1 << 31) and B * OC * OH * OW < (1 << 31) and KH * KW <= 4096)

    def qbytes_conv2d(self, x, w, scales, bias, stride, padding, dilation):
        """Dense convolution with an 8-bit weight [OC, C, KH, KW] and per-channel scales: im2col happens inside the kernel's staging loads.  A weight
        [OC, 1, KH, KW] on an input of C > 1 channels is the depthwise layer (groups = C): the stencil kernel of csrc/qconv_depthwise.hip."""
        self._require_cuda(x, w, scales, bias)
        B, C, H, W = x.shape
        OC, wc, KH, KW = w.shape
        if wc == 1 and C > 1:
            return self._qbytes_conv2d_depthwise(x, w, scales, bias, stride, padding, dilation)
        OH = self.conv2d_out_size(H, KH, stride[0], padding[0], dilation[0])
        OW = self.conv2d_out_size(W, KW, stride[1], padding[1], dilation[1])
        x, w = x.contiguous(), w.contiguous()
        s = scales.reshape(-1).to(x.dtype).contiguous()
        if s.numel() == 1:
            s = s.expand(OC).contiguous()
        if bias is not None:
            bias = bias.to(x.dtype).contiguous()
        y = torch.empty((B, OC, max(OH, 0), max(OW, 0)), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            ws, ws_bytes = self._conv2d_scratch(x, B, OH, OW, OC, C * KH * KW)
            st = self._c.quanto_hip_qbytes_conv2d(_ptr(x), _ptr(w), _ptr(s), _ptr(bias), _ptr(y), B, C, H, W, OC, KH, KW, OH, OW, stride[0], stride[1],
                                                  padding[0], padding[1], dilation[0], dilation[1], _dt(x), _dt(w), _dt(y), _ptr(ws), ws_bytes,
                                                  self._stream(x))
        self._check(st, "qbytes_conv2d")
        return y

    def _qbytes_conv2d_depthwise(self, x, w, scales, bias, stride, padding, dilation):
        B, C, H, W = x.shape
        OC, _, KH, KW = w.shape
        OH = self.conv2d_out_size(H, KH, stride[0], padding[0], dilation[0])
        OW = self.conv2d_out_size(W, KW, stride[1], padding[1], dilation[1])
        x, w = x.contiguous(), w.contiguous()
        s = scales.reshape(-1).to(x.dtype).contiguous()
        if s.numel() == 1:
            s = s.expand(OC).contiguous()
        if bias is not None:
            bias = bias.to(x.dtype).contiguous()
        y = torch.empty((B, OC, max(OH, 0), max(OW, 0)), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            st = self._c.quanto_hip_qbytes_conv2d_depthwise(_ptr(x), _ptr(w), _ptr(s), _ptr(bias), _ptr(y), B, C, H, W, OC, KH, KW, OH, OW, stride[0],
                                                            stride[1], padding[0], padding[1], dilation[0], dilation[1], _dt(x), _dt(w), _dt(y),
                                                            self._stream(x))
        self._check(st, "qbytes_conv2d_depthwise")
        return y

    # -- quanto::qbits_conv2d (implicit GEMM, int4 dequantized while staged) -----------------------------------
    def qbits_conv2d_supported(self, x, weight_size, bits: int, group_size, stride=(1, 1), padding=(0, 0), dilation=(1, 1)) -> bool:
        """NCHW 16-bit activations, generic packed int4 / int2 weight [OC, C, KH, KW] with OC a multiple of the values per byte and groups of a
        multiple of 8 (or per-channel scales), geometry within ``conv2d_geometry_ok``."""
        oc, c, kh, kw = weight_size
        k = c * kh * kw
        return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16) and bits in (2, 4) and oc % (8 // bits) == 0 and
                (not group_size or (group_size % 8 == 0 and k % group_size == 0)) and oc * (k // (group_size or k)) < (1 << 31) and
                self.conv2d_geometry_ok(tuple(x.shape), tuple(weight_size), stride, padding, dilation))

    def qbits_conv2d(self, x, packed, scale, shift, bias, bits: int, group_size, weight_size, stride, padding, dilation):
        """Dense convolution with a generic packed int4 weight (its [OC, C, KH, KW] shape in ``weight_size``): im2col inside the staging loads,
        the weight dequantized there with the reference's roundings (r5: three-tap-wide windows at stride 1 dequantize the weight once into the
        scratch buffer and take the row form of the kernel on it)."""
        self._require_cuda(x, packed, scale, shift, bias)
        if x.dtype != scale.dtype:
            x = x.to(scale.dtype)
        B, C, H, W = x.shape
        OC, _, KH, KW = weight_size
        OH = self.conv2d_out_size(H, KH, stride[0], padding[0], dilation[0])
        OW = self.conv2d_out_size(W, KW, stride[1], padding[1], dilation[1])
        x, packed, scale, shift = x.contiguous(), packed.contiguous(), scale.contiguous(), shift.contiguous()
        if bias is not None:
            bias = bias.to(x.dtype).contiguous()
        y = torch.empty((B, OC, max(OH, 0), max(OW, 0)), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            ws, ws_bytes = self._conv2d_scratch(x, B, OH, OW, OC, C * KH * KW, geom=(C, W, KH, KW, stride[1], dilation[1]))
            st = self._c.quanto_hip_qbits_conv2d(_ptr(x), _ptr(packed), _ptr(scale), _ptr(shift), _ptr(bias), _ptr(y), B, C, H, W, OC, KH, KW, OH, OW,
                                                 stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1], bits, group_size or 0, _dt(x),
                                                 _dt(shift), _ptr(ws), ws_bytes, self._stream(x))
        self._check(st, "qbits_conv2d")
        return y

    # -- quanto::unpack ---------------------------------------------------------------------------
    def unpack(self, t: torch.Tensor, bits: int) -> torch.Tensor:
        self._require_cuda(t)
        if t.dtype != torch.uint8:
            raise QuantoHipError("unpack expects a torch.uint8 tensor")
        if bits not in (2, 4):
            raise ValueError(f"Can only unpack 2-bit or 4-bit tensors, got bits={bits}")
        t = t.contiguous()
        vpi = 8 // bits
        out_shape = (t.shape[0] * vpi,) + tuple(t.shape[1:]) if t.ndim > 0 else (vpi,)
        out = torch.empty(out_shape, dtype=torch.uint8, device=t.device)
        with torch.cuda.device(t.device):
            self._check(self._c.quanto_hip_unpack(_ptr(t), _ptr(out), t.numel(), bits, self._stream(t)), "unpack")
        return out

    # -- fused unpack + dequantize ------------------------------------------------------------------
    def dequantize_qbits(self, packed, scale, shift, bits: int, group_size, out_features: int, in_features: int):
        self._require_cuda(packed, scale, shift)
        packed, scale, shift = packed.contiguous(), scale.contiguous(), shift.contiguous()
        out = torch.empty((out_features, in_features), dtype=scale.dtype, device=packed.device)
        with torch.cuda.device(packed.device):
            st = self._c.quanto_hip_dequantize_qbits(
                _ptr(packed), _ptr(scale), _ptr(shift), _ptr(out), out_features, in_features, bits, group_size or 0,
                _dt(scale), _dt(shift), self._stream(packed))
        self._check(st, "dequantize_qbits")
        return out

    # -- quanto::qbits_mm ---------------------------------------------------------------------------
    def _plan(self, which: str, key, call):
        """(kernel, workspace bytes) of a call shape, asked from the library once per (shape, format, dtype, forced kernel) and kept: the
        choice is a pure function of those (csrc/c_api.hip: pick_q*_kernel; with QUANTO_HIP_EXPERIMENT the knobs may change between calls, so
        nothing is kept then)."""
        cache = self.__dict__.setdefault("_plans", {})
        hit = cache.get((which, key))
        if hit is not None:
            return hit
        k_out, ws_out = ctypes.c_int(0), ctypes.c_int64(0)
        st = call(ctypes.byref(k_out), ctypes.byref(ws_out))
        if st != 0:
            self._check(st, which + "_plan")
        plan = (k_out.value, ws_out.value)
        if not _EXPERIMENT:
            if len(cache) > 4096:
                cache.clear()
            cache[(which, key)] = plan
        return plan

    def qbits_mm(self, x, packed, scale, shift, bias, bits: int, group_size, out_features: int, in_features: int,
                 kernel: str = "auto"):
        if not (x.is_cuda and packed.is_cuda and scale.is_cuda and shift.is_cuda and (bias is None or bias.is_cuda)):
            raise QuantoHipError("quanto_hip kernels only accept tensors on a ROCm device")
        if x.dim() == 0 or x.shape[-1] != in_features:  # the kernel reads M * in_features elements: never from a smaller buffer
            raise QuantoHipError(f"qbits_mm: input of shape {tuple(x.shape)} does not end in in_features = {in_features}")
        sdt = scale.dtype
        if x.dtype != sdt:
            x = x.to(sdt)
        x2 = x if x.dim() == 2 and x.is_contiguous() else x.reshape(-1, in_features).contiguous()
        if not packed.is_contiguous():
            packed = packed.contiguous()
        if not scale.is_contiguous():
            scale = scale.contiguous()
        if not shift.is_contiguous():
            shift = shift.contiguous()
        if bias is not None and (bias.dtype != sdt or not bias.is_contiguous()):
            bias = bias.to(sdt).contiguous()
        M = x2.shape[0]
        gs = group_size or 0
        dt, zdt = _DTYPES.get(sdt), _DTYPES.get(shift.dtype)
        if dt is None or zdt is None:
            raise QuantoHipError(f"quanto_hip: unsupported dtype {sdt if dt is None else shift.dtype}")
        y = torch.empty((M, out_features), dtype=sdt, device=x.device)
        c = self._c
        k, ws_bytes = self._plan("qbits_mm", (M, out_features, in_features, bits, gs, dt, kernel),
                                 lambda ko, wo: c.quanto_hip_qbits_mm_plan(M, out_features, in_features, bits, gs, dt, KERNELS[kernel], ko, wo))
        xp, pp = x2.data_ptr(), packed.data_ptr()
        if kernel == "auto" and (xp | pp) % 16:
            k, ws_bytes = KERNEL_NAIVE, 0  # misaligned view: the kernel without an alignment requirement (what AUTO does in C)
        index = x.device.index
        with _DeviceGuard(x.device):
            stream = _raw_stream(index if index is not None else _current_device())
            if ws_bytes == 0:
                wp = 0
            elif k in (KERNEL_SKINNY, KERNEL_MFMA_FUSED4):
                wp = self._zeroed_workspace(x.device, ws_bytes, stream).data_ptr()  # split-K arrival counters: zero on entry, left zero by the kernel
            else:
                wp = self._scratch(x.device, ws_bytes, stream).data_ptr()
            st = c.quanto_hip_qbits_mm(xp, pp, scale.data_ptr(), shift.data_ptr(), 0 if bias is None else bias.data_ptr(), y.data_ptr(), M,
                                       out_features, in_features, bits, gs, dt, zdt, k, wp, ws_bytes, stream)
        if st != 0:
            self._check(st, "qbits_mm")
        return y if x.dim() == 2 else y.reshape(*x.shape[:-1], out_features)

    # -- quanto::qbits_mm_a8 (quantized activations x int4 weights) ---------------------------------------
    A8_DTYPES = (torch.int8, torch.float8_e4m3fn)

    def qbits_mm_a8_workspace(self, M: int, out_features: int, in_features: int, bits: int, group_size, a_dtype, dtype) -> int:
        """Split-K scratch bytes of the W4A8 kernel for this call shape, or a negative status when the format is not served (the caller then
        dequantizes the activation, as the reference does)."""
        adt, dt = _DTYPES.get(a_dtype), _DTYPES.get(dtype)
        if adt is None or dt is None:
            return -2
        return self._plan("qbits_mm_a8", (M, out_features, in_features, bits, group_size or 0, adt, dt),
                          lambda ko, wo: self._a8_plan(M, out_features, in_features, bits, group_size or 0, adt, dt, ko, wo))[1]

    def _a8_plan(self, M, N, K, bits, gs, adt, dt, kernel_out, ws_out):
        ws_out._obj.value = int(self._c.quanto_hip_qbits_mm_a8_workspace_size(M, N, K, bits, gs, adt, dt))
        return 0

    def qbits_mm_a8(self, a, a_scale, packed, scale, shift, bias, bits: int, group_size, out_features: int, in_features: int):
        """F.linear(quantized activation, int4 weight) on the 8-bit matrix instructions: ``a`` int8 / float8_e4m3fn [..., K], ``a_scale`` its
        per-tensor scale (one element).  Raises QuantoHipError(ENOTSUP) for formats the kernel does not take."""
        if not (a.is_cuda and a_scale.is_cuda and packed.is_cuda and scale.is_cuda and shift.is_cuda and (bias is None or bias.is_cuda)):
            raise QuantoHipError("quanto_hip kernels only accept tensors on a ROCm device")
        if a.dim() == 0 or a.shape[-1] != in_features:
            raise QuantoHipError(f"qbits_mm_a8: input of shape {tuple(a.shape)} does not end in in_features = {in_features}")
        if a_scale.numel() != 1:
            raise QuantoHipError("qbits_mm_a8 expects a per-tensor activation scale")
        sdt = scale.dtype
        a2 = a if a.dim() == 2 and a.is_contiguous() else a.reshape(-1, in_features).contiguous()
        a_scale = a_scale.reshape(1).to(sdt).contiguous()
        packed, scale, shift = packed.contiguous(), scale.contiguous(), shift.contiguous()
        if bias is not None:
            bias = bias.to(sdt).contiguous()
        M = a2.shape[0]
        ws_bytes = self.qbits_mm_a8_workspace(M, out_features, in_features, bits, group_size, a2.dtype, sdt)
        if ws_bytes < 0:
            self._check(int(ws_bytes), "qbits_mm_a8")
        y = torch.empty((M, out_features), dtype=sdt, device=a.device)
        index = a.device.index
        with _DeviceGuard(a.device):
            stream = _raw_stream(index if index is not None else _current_device())
            wp = self._zeroed_workspace(a.device, ws_bytes, stream).data_ptr() if ws_bytes > 0 else 0
            st = self._c.quanto_hip_qbits_mm_a8(a2.data_ptr(), a_scale.data_ptr(), packed.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                0 if bias is None else bias.data_ptr(), y.data_ptr(), M, out_features, in_features, bits,
                                                group_size or 0, _DTYPES[a2.dtype], _DTYPES[sdt], _dt(shift), wp, ws_bytes, stream)
        if st != 0:
            self._check(st, "qbits_mm_a8")
        return y if a.dim() == 2 else y.reshape(*a.shape[:-1], out_features)

    # -- quanto::qbits_mm_multi ---------------------------------------------------------------------
    MAX_MULTI = 4  # QUANTO_HIP_MAX_MULTI

    def qbits_mm_multi(self, x, packed, scale, shift, bias, bits: int, group_size, out_features, in_features: int):
        """Several qbits_mm products sharing ``x`` (q/k/v, gate/up).  One kernel launch when every product is eligible for
        the decode GEMV (M <= 4, int4, group size 128: bit-identical to the separate calls) or, for batched decode (4 < M <= 64,
        every out_features a multiple of 64), for one launch of the streaming MFMA kernel over all members; otherwise the
        separate ops (each with its own kernel choice and workspace).  Returns the list of outputs."""
        n = len(packed)
        bias = list(bias) if bias is not None else [None] * n
        if not (len(scale) == len(shift) == len(bias) == len(out_features) == n) or n < 1:
            raise QuantoHipError("qbits_mm_multi: inconsistent argument lists")
        self._require_cuda(x, *packed, *scale, *shift, *bias)
        sdt = scale[0].dtype
        if x.dtype != sdt:
            x = x.to(sdt)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, in_features).contiguous()
        M = x2.shape[0]
        kernel, ws_bytes = KERNEL_AUTO, 0
        nfs = (ctypes.c_int64 * n)(*out_features)
        if 1 <= M <= 64 and n <= self.MAX_MULTI and all(s.dtype == sdt for s in scale) and len({sh.dtype for sh in shift}) == 1:
            k_out, ws_out = ctypes.c_int(0), ctypes.c_int64(0)
            with torch.cuda.device(x.device):
                st = self._c.quanto_hip_qbits_mm_multi_plan(n, nfs, M, in_features, bits, group_size or 0, _dt(scale[0]), ctypes.byref(k_out),
                                                            ctypes.byref(ws_out))
            if st == 0:
                kernel, ws_bytes = k_out.value, ws_out.value
        if kernel == KERNEL_AUTO:
            return [self.qbits_mm(x, packed[i], scale[i], shift[i], bias[i], bits, group_size, out_features[i], in_features)
                    for i in range(n)]
        packed = [t.contiguous() for t in packed]
        scale = [t.contiguous() for t in scale]
        shift = [t.contiguous() for t in shift]
        bias = [None if b is None else b.to(sdt).contiguous() for b in bias]
        ys = [torch.empty((M, nf), dtype=sdt, device=x.device) for nf in out_features]
        arr = lambda ts: (ctypes.c_void_p * n)(*[0 if t is None else t.data_ptr() for t in ts])  # noqa: E731
        with torch.cuda.device(x.device):
            ws = self._zeroed_workspace(x.device, ws_bytes, self._stream(x).value) if ws_bytes else None  # split-K arrival counters
            st = self._c.quanto_hip_qbits_mm_multi_ws(_ptr(x2), n, arr(packed), arr(scale), arr(shift), arr(bias), arr(ys), nfs, M, in_features,
                                                      bits, group_size or 0, _dt(scale[0]), _dt(shift[0]), _ptr(ws), ws_bytes, self._stream(x))
        self._check(st, "qbits_mm_multi")
        return [y.reshape(*lead, nf) for y, nf in zip(ys, out_features)]

    # -- quanto::qbytes_mm_multi --------------------------------------------------------------------
    def qbytes_mm_multi(self, a, weights, scales, bias):
        """Several qbytes_mm products sharing the activation (q/k/v, gate/up of an int8 / fp8 model): one GEMV launch for M <= 2,
        one streaming-MFMA launch for M <= 64 when every out_features is a multiple of 64, otherwise the separate ops."""
        n = len(weights)
        bias = list(bias) if bias is not None else [None] * n
        if not (len(scales) == len(bias) == n) or n < 1:
            raise QuantoHipError("qbytes_mm_multi: inconsistent argument lists")
        self._require_cuda(a, *weights, *scales, *bias)
        K = weights[0].shape[1]
        sdt = scales[0].dtype
        one_call = (n <= self.MAX_MULTI and a.dtype.is_floating_point and a.dtype.itemsize > 1 and all(w.shape[1] == K for w in weights)
                    and all(w.dtype == weights[0].dtype for w in weights) and all(s.dtype == sdt for s in scales)
                    and all(s.numel() == w.shape[0] for s, w in zip(scales, weights)))
        kernel, ws_bytes = KERNEL_AUTO, 0
        if one_call:
            if a.dtype != sdt:
                a = a.to(sdt)
            lead = a.shape[:-1]
            a2 = a.reshape(-1, K).contiguous()
            M = a2.shape[0]
            nfs = (ctypes.c_int64 * n)(*[w.shape[0] for w in weights])
            if 1 <= M <= 64:
                k_out, ws_out = ctypes.c_int(0), ctypes.c_int64(0)
                with torch.cuda.device(a.device):
                    st = self._c.quanto_hip_qbytes_mm_multi_plan(n, nfs, M, K, _dt(a2), _dt(weights[0]), _dt(scales[0]), ctypes.byref(k_out),
                                                                 ctypes.byref(ws_out))
                if st == 0:
                    kernel, ws_bytes = k_out.value, ws_out.value
        if kernel == KERNEL_AUTO:
            return [self.qbytes_mm(a, weights[i], scales[i], bias[i]) for i in range(n)]
        weights = [w.contiguous() for w in weights]
        scales = [s.reshape(-1).contiguous() for s in scales]
        bias = [None if b is None else b.to(sdt).contiguous() for b in bias]
        ys = [torch.empty((M, w.shape[0]), dtype=sdt, device=a.device) for w in weights]
        arr = lambda ts: (ctypes.c_void_p * n)(*[0 if t is None else t.data_ptr() for t in ts])  # noqa: E731
        with torch.cuda.device(a.device):
            ws = self._zeroed_workspace(a.device, ws_bytes, self._stream(a).value) if ws_bytes else None  # split-K arrival counters
            st = self._c.quanto_hip_qbytes_mm_multi_ws(_ptr(a2), n, arr(weights), arr(scales), arr(bias), arr(ys), nfs, M, K, _dt(a2),
                                                       _dt(weights[0]), _dt(scales[0]), _ptr(ws), ws_bytes, self._stream(a))
        self._check(st, "qbytes_mm_multi")
        return [y.reshape(*lead, w.shape[0]) for y, w in zip(ys, weights)]

    # -- quanto::qbytes_mm --------------------------------------------------------------------------
    def qbytes_mm(self, a, b, scales, bias=None, kernel: str = "auto"):
        if not (a.is_cuda and b.is_cuda and scales.is_cuda and (bias is None or bias.is_cuda)):
            raise QuantoHipError("quanto_hip kernels only accept tensors on a ROCm device")
        N, K = b.shape
        if scales.numel() != N:
            raise QuantoHipError(f"qbytes_mm expects one scale per output feature ({N}), got {tuple(scales.shape)}")
        if a.dim() == 0 or a.shape[-1] != K:  # torch.matmul's shape error in the reference; here the kernel would read past the buffer
            raise QuantoHipError(f"qbytes_mm: input of shape {tuple(a.shape)} does not end in in_features = {K}")
        sdt = scales.dtype
        if a.dtype.is_floating_point and a.dtype.itemsize > 1 and a.dtype != sdt:
            a = a.to(sdt)  # library/qbytes_mm.py:26
        a2 = a if a.dim() == 2 and a.is_contiguous() else a.reshape(-1, K).contiguous()
        if not b.is_contiguous():
            b = b.contiguous()
        s = scales if scales.dim() == 1 and scales.is_contiguous() else scales.reshape(-1).contiguous()
        if bias is not None and (bias.dtype != sdt or not bias.is_contiguous()):
            bias = bias.to(sdt).contiguous()
        M = a2.shape[0]
        adt, bdt, odt = _DTYPES.get(a2.dtype), _DTYPES.get(b.dtype), _DTYPES.get(sdt)
        if adt is None or bdt is None or odt is None:
            raise QuantoHipError(f"quanto_hip: unsupported dtype in qbytes_mm({a2.dtype}, {b.dtype}, {sdt})")
        y = torch.empty((M, N), dtype=sdt, device=a.device)
        c = self._c
        k, ws_bytes = self._plan("qbytes_mm", (M, N, K, adt, bdt, odt, kernel),
                                 lambda ko, wo: c.quanto_hip_qbytes_mm_plan(M, N, K, adt, bdt, odt, KERNELS[kernel], ko, wo))
        ap, bp = a2.data_ptr(), b.data_ptr()
        if kernel == "auto" and (ap | bp) % 16:
            k, ws_bytes = KERNEL_NAIVE, 0  # misaligned view: the kernel without an alignment requirement (what AUTO does in C)
        index = a.device.index
        with _DeviceGuard(a.device):
            stream = _raw_stream(index if index is not None else _current_device())
            wp = self._zeroed_workspace(a.device, ws_bytes, stream).data_ptr() if ws_bytes > 0 else 0  # split-K arrival counters: zero on entry, left zero
            st = c.quanto_hip_qbytes_mm_ws(ap, bp, s.data_ptr(), 0 if bias is None else bias.data_ptr(), y.data_ptr(), M, N, K, adt, bdt, odt, k, wp,
                                           max(ws_bytes, 0), stream)
        if st != 0:
            self._check(st, "qbytes_mm")
        return y if a.dim() == 2 else y.reshape(*a.shape[:-1], N)


class QuantoHipExtension(NativeLibrary):
    """The ``quanto_hip`` extension (name expected by the reference's tests/library/test_extensions.py:23-24)."""

    def __init__(self):
        csrc = os.path.join(_PKG_DIR, "csrc")
        super().__init__(
            "quanto_hip",
            root_dir=csrc,
            lib_path=os.path.join(_PKG_DIR, "lib", "libquanto_hip.so"),
            sources=["c_api.hip", "unpack.hip", "naive_mm.hip", "qbits_gemv.hip", "qbytes_gemv.hip", "qmm_mfma.hip", "qconv_mfma.hip", "qconv_depthwise.hip", "qmm_mfma_large.hip", "qmm_large_common.h", "qbits_skinny.hip", "qbits_mmv.hip", "qbits_mfma_fused.hip", "qbits_a8_fused.hip", "qbits_mfma_large.hip", "qbytes_skinny.hip", "qmm_native8.hip", "qmm_f32.hip", "quantize.hip",
                     "qh_common.h", os.path.join("..", "..", "include", "quanto_hip.h")],
        )
        self._bindings = None

    @property
    def lib(self) -> _Bindings:
        if self._bindings is None:
            try:
                self._bindings = _Bindings(self.cdll)
            except OSError as e:
                raise QuantoHipError(
                    f"quanto_hip: cannot load {self.lib_path} ({e}). Build it with `python -c 'import __graft_entry__ as g; "
                    "g.build()'` (hipcc --offload-arch=gfx950). There is no fallback for ROCm tensors.") from e
        return self._bindings


quanto_hip = QuantoHipExtension()
# The reference registers its HIP extension only when a ROCm device is visible
# (library/extensions/__init__.py:24-28); same gate here, plus "the binary exists" so that a GPU box without
# the library fails loudly at first use instead of silently running something else.
if torch.version.hip is not None:
    register_extension(quanto_hip)
