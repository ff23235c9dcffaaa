// quanto::quantize_symmetric as ONE pass: out8 = cast(clamp(round?(base / scale))).
//
// Replaces the 4-5 elementwise torch kernels of library/quantize.py:26-55 (div, round, clamp, cast - each a full HBM
// round trip over the activation) on the per-forward path of quantized activations (nn/qmodule.py:281-291 ->
// tensor/activations/qbytes.py:31-39).  Algorithmic traffic: sizeof(T) + 1 bytes per element; HBM-bound.
//
// Bit-exactness with the torch sequence: the quotient is computed in fp32 with a correctly rounded divide and rounded
// to the tensor dtype T (what aten's div does through its opmath type); integer targets are then rounded half-to-even
// *in T* (exact: |q| <= 256 after the clamp matters only) and clamped to [-128, 127]; float8 targets are clamped to the
// finite range and converted with the hardware's round-to-nearest-even OCP converters (v_cvt_pk_fp8_f32 / bf8).
#include "qh_common.h"

namespace qh {
namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

enum { SCALE_TENSOR = QUANTO_HIP_SCALE_PER_TENSOR, SCALE_FIRST = QUANTO_HIP_SCALE_AXIS_FIRST, SCALE_LAST = QUANTO_HIP_SCALE_AXIS_LAST };

template <int ODT>
__device__ __forceinline__ float clamp_target(float q) {
  if constexpr (ODT == QUANTO_HIP_I8) {
    q = __builtin_rintf(q);
    return __builtin_fminf(__builtin_fmaxf(q, -128.f), 127.f);
  } else if constexpr (ODT == QUANTO_HIP_F8_E4M3FN) {
    return __builtin_fminf(__builtin_fmaxf(q, -448.f), 448.f);
  } else {
    return __builtin_fminf(__builtin_fmaxf(q, -57344.f), 57344.f);
  }
}

template <int ODT>
__device__ __forceinline__ uint32_t pack4(const float* q) {
  if constexpr (ODT == QUANTO_HIP_I8) {
    return ((uint32_t)(int)q[0] & 0xFFu) | (((uint32_t)(int)q[1] & 0xFFu) << 8) | (((uint32_t)(int)q[2] & 0xFFu) << 16) |
           (((uint32_t)(int)q[3] & 0xFFu) << 24);
  } else if constexpr (ODT == QUANTO_HIP_F8_E4M3FN) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w, true);
  } else {
    int w = __builtin_amdgcn_cvt_pk_bf8_f32(q[0], q[1], 0, false);
    return (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(q[2], q[3], w, true);
  }
}

// 8 elements per thread and iteration: one 16-byte (16-bit T) or two 16-byte (fp32) loads, one 8-byte store.
template <int IDT, int ODT, int MODE>
__global__ void __launch_bounds__(256) quantize_symmetric_kernel(const typename Elem<IDT>::T* __restrict__ x,
                                                                 const typename Elem<IDT>::T* __restrict__ scale,
                                                                 uint8_t* __restrict__ out, int64_t numel, int64_t inner) {
  using E = Elem<IDT>;
  using T = typename E::T;
  const int64_t nvec = numel >> 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float s_tensor = 1.f;
  if constexpr (MODE == SCALE_TENSOR) s_tensor = E::to_f32(scale[0]);
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec; v += stride) {
    T e[8];
    if constexpr (sizeof(T) == 2) {
      *reinterpret_cast<u32x4*>(e) = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x) + v);
    } else {
      reinterpret_cast<u32x4*>(e)[0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x) + 2 * v);
      reinterpret_cast<u32x4*>(e)[1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x) + 2 * v + 1);
    }
    float q[8];
    const int64_t i0 = v << 3;
    if constexpr (MODE == SCALE_FIRST) {
      // `inner` elements share a scale; a vector of 8 may straddle two slices
      const int64_t r0 = i0 / inner;
      const int64_t left = (r0 + 1) * inner - i0;  // elements of this vector that belong to slice r0
      const float s0 = E::to_f32(scale[r0]);
      const float s1 = left < 8 ? E::to_f32(scale[r0 + 1 < (numel / inner) ? r0 + 1 : r0]) : s0;
      if (inner >= 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = E::to_f32(e[k]) / (k < left ? s0 : s1);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = E::to_f32(e[k]) / E::to_f32(scale[(i0 + k) / inner]);
      }
    } else if constexpr (MODE == SCALE_LAST) {
      int64_t c = i0 % inner;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        q[k] = E::to_f32(e[k]) / E::to_f32(scale[c]);
        c = c + 1 == inner ? 0 : c + 1;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) q[k] = E::to_f32(e[k]) / s_tensor;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = clamp_target<ODT>(E::to_f32(E::from_f32(q[k])));
    uint2 w;
    w.x = pack4<ODT>(q);
    w.y = pack4<ODT>(q + 4);
    reinterpret_cast<uint2*>(out)[v] = w;
  }
  // ragged tail (< 8 elements), one thread
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int64_t i = nvec << 3; i < numel; ++i) {
      float s;
      if constexpr (MODE == SCALE_FIRST)
        s = E::to_f32(scale[i / inner]);
      else if constexpr (MODE == SCALE_LAST)
        s = E::to_f32(scale[i % inner]);
      else
        s = s_tensor;
      float q[4] = {clamp_target<ODT>(E::to_f32(E::from_f32(E::to_f32(x[i]) / s))), 0.f, 0.f, 0.f};
      out[i] = (uint8_t)(pack4<ODT>(q) & 0xFFu);
    }
  }
}

template <int IDT, int ODT>
int launch_mode(const void* x, const void* s, void* out, int64_t numel, int64_t inner, int mode, hipStream_t stream) {
  using T = typename Elem<IDT>::T;
  const int64_t nvec = numel >> 3;
  int64_t blocks = (nvec + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 workgroups per CU
  const T* xp = reinterpret_cast<const T*>(x);
  const T* sp = reinterpret_cast<const T*>(s);
  uint8_t* op = reinterpret_cast<uint8_t*>(out);
  if (mode == SCALE_TENSOR)
    hipLaunchKernelGGL((quantize_symmetric_kernel<IDT, ODT, SCALE_TENSOR>), dim3(blocks), dim3(256), 0, stream, xp, sp, op, numel, inner);
  else if (mode == SCALE_FIRST)
    hipLaunchKernelGGL((quantize_symmetric_kernel<IDT, ODT, SCALE_FIRST>), dim3(blocks), dim3(256), 0, stream, xp, sp, op, numel, inner);
  else
    hipLaunchKernelGGL((quantize_symmetric_kernel<IDT, ODT, SCALE_LAST>), dim3(blocks), dim3(256), 0, stream, xp, sp, op, numel, inner);
  return launch_status();
}

// ---- QBytesTensor.dequantize() for a per-tensor scale (activations): out = T(scale * T(q)) in ONE pass -------------------------------------------------
// (tensor/qbytes.py:23-36: `scale * data.to(dtype)` = a cast kernel and a multiply kernel, 7 bytes of traffic per element in bf16 - here 3).  Every int8 / fp8
// value is exact in bf16 / fp16 / fp32, so T(q) is exact and the product has one rounding: fp32 multiply, rounded to T - what aten's mul does through its
// opmath type.  16 elements per thread and iteration: one 16-byte load, two 16-byte (16-bit T) or four (fp32) stores.
template <int QDT, int ODT>
__global__ void __launch_bounds__(256) dequantize_symmetric_kernel(const uint8_t* __restrict__ q, const typename Elem<ODT>::T* __restrict__ scale,
                                                                   typename Elem<ODT>::T* __restrict__ out, int64_t numel) {
  using E = Elem<ODT>;
  using T = typename E::T;
  const float s = E::to_f32(scale[0]);
  const int64_t nvec = numel >> 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec; v += stride) {
    uint8_t b[16];
    *reinterpret_cast<u32x4*>(b) = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q) + v);
    T o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) o[k] = E::from_f32(decode8<QDT>(b[k]) * s);
#pragma unroll
    for (int k = 0; k < (int)(16 * sizeof(T) / 16); ++k)
      __builtin_nontemporal_store(reinterpret_cast<const u32x4*>(o)[k], reinterpret_cast<u32x4*>(out + (v << 4)) + k);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = nvec << 4; i < numel; ++i) out[i] = E::from_f32(decode8<QDT>(q[i]) * s);
}

template <int QDT, int ODT>
int launch_dequantize(const void* q, const void* scale, void* out, int64_t numel, hipStream_t stream) {
  using T = typename Elem<ODT>::T;
  const int64_t nvec = numel >> 4;
  const int blocks = (int)(nvec < 256 ? 1 : (nvec + 255) / 256 > 4096 ? 4096 : (nvec + 255) / 256);
  hipLaunchKernelGGL((dequantize_symmetric_kernel<QDT, ODT>), dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const uint8_t*>(q),
                     reinterpret_cast<const T*>(scale), reinterpret_cast<T*>(out), numel);
  return launch_status();
}

template <int IDT>
int launch_out(const void* x, const void* s, void* out, int64_t numel, int64_t inner, int mode, int out_dtype, hipStream_t stream) {
  switch (out_dtype) {
    case QUANTO_HIP_I8: return launch_mode<IDT, QUANTO_HIP_I8>(x, s, out, numel, inner, mode, stream);
    case QUANTO_HIP_F8_E4M3FN: return launch_mode<IDT, QUANTO_HIP_F8_E4M3FN>(x, s, out, numel, inner, mode, stream);
    case QUANTO_HIP_F8_E5M2: return launch_mode<IDT, QUANTO_HIP_F8_E5M2>(x, s, out, numel, inner, mode, stream);
  }
  return QUANTO_HIP_ENOTSUP;
}

// ---- quanto::quantize_affine for axis-0 weights: uint8 in [0, 2^bits) per element of the grouped matrix ---------------------
// (library/quantize.py:66-78).  Grouping along axis 0 is a pure reshape ([N,K] -> [N*K/C, C], tensor/grouped.py:17-39), so
// element i uses scale/shift entry i / C.  float shift: round(round_T(round_T(base + shift) / scale)); integer zero-point:
// round(round_T(base / scale)) + zp; both clamped to [0, 2^bits - 1] - every intermediate rounded to T like the torch sequence.
template <int IDT, bool INT_SHIFT>
__device__ __forceinline__ uint32_t affine_code(float xv, float s, const void* __restrict__ shift, int64_t r, float qmax) {
  using E = Elem<IDT>;
  using T = typename E::T;
  float q;
  if constexpr (INT_SHIFT) {
    const float zp = (float)(int8_t) reinterpret_cast<const uint8_t*>(shift)[r];
    q = __builtin_rintf(E::to_f32(E::from_f32(xv / s))) + zp;
    q = E::to_f32(E::from_f32(q));
  } else {
    const float z = E::to_f32(reinterpret_cast<const T*>(shift)[r]);
    float t = E::to_f32(E::from_f32(xv + z));
    q = __builtin_rintf(E::to_f32(E::from_f32(t / s)));
  }
  q = __builtin_fminf(__builtin_fmaxf(q, 0.f), qmax);
  return (uint32_t)(int)q & 0xFFu;
}

template <int IDT, bool INT_SHIFT>
__global__ void __launch_bounds__(256) quantize_affine_kernel(const typename Elem<IDT>::T* __restrict__ x,
                                                              const typename Elem<IDT>::T* __restrict__ scale, const void* __restrict__ shift,
                                                              uint8_t* __restrict__ out, int64_t numel, int64_t C, float qmax) {
  using E = Elem<IDT>;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i0 < numel; i0 += stride) {
    uint32_t word = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = i0 + k;
      if (i < numel) {
        const int64_t r = i / C;
        word |= affine_code<IDT, INT_SHIFT>(E::to_f32(x[i]), E::to_f32(scale[r]), shift, r, qmax) << (8 * k);
      }
    }
    if (i0 + 3 < numel && (reinterpret_cast<uintptr_t>(out + i0) & 3) == 0) {
      *reinterpret_cast<uint32_t*>(out + i0) = word;
    } else {
      for (int k = 0; k < 4 && i0 + k < numel; ++k) out[i0 + k] = (uint8_t)(word >> (8 * k));
    }
  }
}

// ---- quantize_affine + pack_weights in one pass (what freeze() does to an int4 / int2 weight: library/quantize.py:66-78, then
// tensor/packed.py:24-69).  The grouped matrix is [rows, C] with one scale/shift per row, so packed byte (r, c) holds the codes
// of elements (r + k*row_dim, c), k = 0 .. 8/bits - 1: each thread quantizes those 2 or 4 elements and writes one byte; the
// one-code-per-byte intermediate (2 x the packed size written, then re-read) never exists.
template <int IDT, bool INT_SHIFT>
__global__ void __launch_bounds__(256) quantize_affine_pack_kernel(const typename Elem<IDT>::T* __restrict__ x,
                                                                   const typename Elem<IDT>::T* __restrict__ scale,
                                                                   const void* __restrict__ shift, uint8_t* __restrict__ out, int64_t rows,
                                                                   int64_t C, int64_t row_dim, int bits, float qmax) {
  using E = Elem<IDT>;
  const int64_t total = row_dim * C;
  const int vpi = 8 / bits;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C, c = i - r * C;
    uint32_t v = 0;
    for (int k = 0; k < vpi; ++k) {
      const int64_t rr = r + k * row_dim;
      if (rr < rows) v |= affine_code<IDT, INT_SHIFT>(E::to_f32(x[rr * C + c]), E::to_f32(scale[rr]), shift, rr, qmax) << (bits * k);
    }
    out[i] = (uint8_t)v;
  }
}

// ---- pack_weights (tensor/packed.py:24-69): packed[r, c] = OR_i unpacked[r + i*row_dim, c] << (bits*i) -------------------------
__global__ void __launch_bounds__(256) pack_kernel(const uint8_t* __restrict__ u, uint8_t* __restrict__ p, int64_t rows, int64_t cols,
                                                   int64_t row_dim, int bits) {
  const int64_t total = row_dim * cols;
  const int vpi = 8 / bits;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    uint32_t v = 0;
    for (int k = 0; k < vpi; ++k) {
      const int64_t rr = r + k * row_dim;
      if (rr < rows) v |= (uint32_t)u[rr * cols + (i - r * cols)] << (bits * k);
    }
    p[i] = (uint8_t)v;
  }
}

}  // namespace

int quantize_symmetric(const void* x, const void* s, void* out, int64_t numel, int64_t inner, int mode, int in_dtype, int out_dtype,
                       hipStream_t stream) {
  if ((reinterpret_cast<uintptr_t>(x) % 16) || (reinterpret_cast<uintptr_t>(out) % 8)) return QUANTO_HIP_EALIGN;
  switch (in_dtype) {
    case QUANTO_HIP_F32: return launch_out<QUANTO_HIP_F32>(x, s, out, numel, inner, mode, out_dtype, stream);
    case QUANTO_HIP_F16: return launch_out<QUANTO_HIP_F16>(x, s, out, numel, inner, mode, out_dtype, stream);
    case QUANTO_HIP_BF16: return launch_out<QUANTO_HIP_BF16>(x, s, out, numel, inner, mode, out_dtype, stream);
  }
  return QUANTO_HIP_ENOTSUP;
}

int dequantize_symmetric(const void* q, const void* scale, void* out, int64_t numel, int q_dtype, int out_dtype, hipStream_t stream) {
  if ((reinterpret_cast<uintptr_t>(q) % 16) || (reinterpret_cast<uintptr_t>(out) % 16)) return QUANTO_HIP_EALIGN;
#define QH_DEQ(ODT)                                                                                                  \
  if (q_dtype == QUANTO_HIP_I8) return launch_dequantize<QUANTO_HIP_I8, ODT>(q, scale, out, numel, stream);              \
  if (q_dtype == QUANTO_HIP_F8_E4M3FN) return launch_dequantize<QUANTO_HIP_F8_E4M3FN, ODT>(q, scale, out, numel, stream); \
  if (q_dtype == QUANTO_HIP_F8_E5M2) return launch_dequantize<QUANTO_HIP_F8_E5M2, ODT>(q, scale, out, numel, stream);     \
  return QUANTO_HIP_ENOTSUP
  if (out_dtype == QUANTO_HIP_BF16) { QH_DEQ(QUANTO_HIP_BF16); }
  if (out_dtype == QUANTO_HIP_F16) { QH_DEQ(QUANTO_HIP_F16); }
  if (out_dtype == QUANTO_HIP_F32) { QH_DEQ(QUANTO_HIP_F32); }
#undef QH_DEQ
  return QUANTO_HIP_ENOTSUP;
}

int quantize_affine(const void* x, const void* scale, const void* shift, void* out, int64_t numel, int64_t C, int bits, int dtype,
                    bool int_shift, hipStream_t stream) {
  int64_t blocks = (numel / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const float qmax = (float)((1 << bits) - 1);
  uint8_t* op = reinterpret_cast<uint8_t*>(out);
#define QH_QA(DT)                                                                                                               \
  if (int_shift)                                                                                                                \
    hipLaunchKernelGGL((quantize_affine_kernel<DT, true>), dim3(blocks), dim3(256), 0, stream,                                  \
                       reinterpret_cast<const Elem<DT>::T*>(x), reinterpret_cast<const Elem<DT>::T*>(scale), shift, op, numel, C, qmax); \
  else                                                                                                                          \
    hipLaunchKernelGGL((quantize_affine_kernel<DT, false>), dim3(blocks), dim3(256), 0, stream,                                 \
                       reinterpret_cast<const Elem<DT>::T*>(x), reinterpret_cast<const Elem<DT>::T*>(scale), shift, op, numel, C, qmax)
  switch (dtype) {
    case QUANTO_HIP_F32: QH_QA(QUANTO_HIP_F32); break;
    case QUANTO_HIP_F16: QH_QA(QUANTO_HIP_F16); break;
    case QUANTO_HIP_BF16: QH_QA(QUANTO_HIP_BF16); break;
    default: return QUANTO_HIP_ENOTSUP;
  }
#undef QH_QA
  return launch_status();
}

int quantize_affine_packed(const void* x, const void* scale, const void* shift, void* out, int64_t rows, int64_t C, int bits, int dtype,
                           bool int_shift, hipStream_t stream) {
  const int vpi = 8 / bits;
  const int64_t row_dim = (rows + vpi - 1) / vpi;
  int64_t blocks = (row_dim * C + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const float qmax = (float)((1 << bits) - 1);
  uint8_t* op = reinterpret_cast<uint8_t*>(out);
#define QH_QAP(DT)                                                                                                              \
  if (int_shift)                                                                                                                \
    hipLaunchKernelGGL((quantize_affine_pack_kernel<DT, true>), dim3(blocks), dim3(256), 0, stream,                             \
                       reinterpret_cast<const Elem<DT>::T*>(x), reinterpret_cast<const Elem<DT>::T*>(scale), shift, op, rows, C, row_dim, bits, qmax); \
  else                                                                                                                          \
    hipLaunchKernelGGL((quantize_affine_pack_kernel<DT, false>), dim3(blocks), dim3(256), 0, stream,                            \
                       reinterpret_cast<const Elem<DT>::T*>(x), reinterpret_cast<const Elem<DT>::T*>(scale), shift, op, rows, C, row_dim, bits, qmax)
  switch (dtype) {
    case QUANTO_HIP_F32: QH_QAP(QUANTO_HIP_F32); break;
    case QUANTO_HIP_F16: QH_QAP(QUANTO_HIP_F16); break;
    case QUANTO_HIP_BF16: QH_QAP(QUANTO_HIP_BF16); break;
    default: return QUANTO_HIP_ENOTSUP;
  }
#undef QH_QAP
  return launch_status();
}

int pack_weights(const uint8_t* unpacked, uint8_t* packed, int64_t rows, int64_t cols, int bits, hipStream_t stream) {
  const int vpi = 8 / bits;
  const int64_t row_dim = (rows + vpi - 1) / vpi;
  int64_t blocks = (row_dim * cols + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, stream, unpacked, packed, rows, cols, row_dim, bits);
  return launch_status();
}

}  // namespace qh
