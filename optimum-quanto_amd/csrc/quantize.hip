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

template <int IDT>
int launch_out(const void* x, const void* s, void* out, int64_t numel, int64_t inner, int mode, int out_dtype, hipStream_t stream) {
  switch (out_dtype) {
    case QUANTO_HIP_I8: return launch_mode<IDT, QUANTO_HIP_I8>(x, s, out, numel, inner, mode, stream);
    case QUANTO_HIP_F8_E4M3FN: return launch_mode<IDT, QUANTO_HIP_F8_E4M3FN>(x, s, out, numel, inner, mode, stream);
    case QUANTO_HIP_F8_E5M2: return launch_mode<IDT, QUANTO_HIP_F8_E5M2>(x, s, out, numel, inner, mode, stream);
  }
  return QUANTO_HIP_ENOTSUP;
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

}  // namespace qh
