// Shape-agnostic correctness kernels: one thread per output element, fp32 accumulation over K.
// Used for shapes the fast kernels reject (odd K, int2, fp32 activations, e4m3fnuz, ...), and as the
// on-device cross-check in the parity tests.  Not a performance path.
#include "qh_common.h"

namespace qh {

// ---- qbytes_mm: y[m,n] = scale[n] * sum_k a[m,k] * b[n,k] (+ bias[n]) ------------------------------
template <typename AT>
__device__ __forceinline__ float a_to_f32(AT v) {
  return (float)v;
}

template <int ADT>
struct ALoad;  // activation element loader
template <>
struct ALoad<QUANTO_HIP_F32> {
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return reinterpret_cast<const float*>(p)[i]; }
};
template <>
struct ALoad<QUANTO_HIP_F16> {
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return (float)reinterpret_cast<const _Float16*>(p)[i]; }
};
template <>
struct ALoad<QUANTO_HIP_BF16> {
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return (float)reinterpret_cast<const __bf16*>(p)[i]; }
};
template <>
struct ALoad<QUANTO_HIP_I8> {
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return (float)reinterpret_cast<const int8_t*>(p)[i]; }
};
template <>
struct ALoad<QUANTO_HIP_F8_E4M3FN> {
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return decode8<QUANTO_HIP_F8_E4M3FN>(reinterpret_cast<const uint8_t*>(p)[i]); }
};
template <>
struct ALoad<QUANTO_HIP_F8_E5M2> {
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return decode8<QUANTO_HIP_F8_E5M2>(reinterpret_cast<const uint8_t*>(p)[i]); }
};
template <>
struct ALoad<QUANTO_HIP_F8_E4M3FNUZ> {
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return decode8<QUANTO_HIP_F8_E4M3FNUZ>(reinterpret_cast<const uint8_t*>(p)[i]); }
};

template <int ADT, int BDT, int ODT>
__global__ void __launch_bounds__(256)
    qbytes_mm_naive_kernel(const void* __restrict__ a, const uint8_t* __restrict__ b, const typename Elem<ODT>::T* __restrict__ scales,
                           const typename Elem<ODT>::T* __restrict__ bias, typename Elem<ODT>::T* __restrict__ y, int64_t M,
                           int64_t N, int64_t K) {
  using E = Elem<ODT>;
  const int64_t total = M * N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N, n = i - m * N;
    float acc = 0.f;
    if constexpr (ADT == QUANTO_HIP_I8 && BDT == QUANTO_HIP_I8) {
      int32_t iacc = 0;  // library/qbytes_mm.py:36-50: exact int32 accumulation, fp32 rescale
      for (int64_t k = 0; k < K; ++k)
        iacc += (int32_t) reinterpret_cast<const int8_t*>(a)[m * K + k] * (int32_t)(int8_t)b[n * K + k];
      acc = (float)iacc;
    } else {
      for (int64_t k = 0; k < K; ++k) acc = __builtin_fmaf(ALoad<ADT>::ld(a, m * K + k), decode8<BDT>(b[n * K + k]), acc);
    }
    float r = acc * E::to_f32(scales[n]);
    asm volatile("" : "+v"(r));  // keep the fp32 rounding of the product (no fused v_fma_mixlo_f16): reference rounds twice
    if (bias) r = E::to_f32(E::from_f32(r)) + E::to_f32(bias[n]);  // output rounded, then bias added (tensor/weights/qbytes.py:79-81)
    y[i] = E::from_f32(r);
  }
}

// ---- qbits_mm: generic PackedTensor layout ----------------------------------------------------------
template <int DT, int BITS, bool INT_SHIFT>
__global__ void __launch_bounds__(256)
    qbits_mm_naive_kernel(const typename Elem<DT>::T* __restrict__ x, const uint8_t* __restrict__ packed,
                          const typename Elem<DT>::T* __restrict__ scale, const void* __restrict__ shift_,
                          const typename Elem<DT>::T* __restrict__ bias, typename Elem<DT>::T* __restrict__ y, int64_t M, int64_t N,
                          int64_t K, int64_t C, int64_t G, int64_t row_dim) {
  using E = Elem<DT>;
  using T = typename E::T;
  constexpr uint32_t MASK = (1u << BITS) - 1u;
  const int64_t total = M * N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N, n = i - m * N;
    float acc = 0.f;
    for (int64_t kg = 0; kg < G; ++kg) {
      const int64_t gr = n * G + kg;
      const int64_t plane = gr / row_dim, r = gr - plane * row_dim;
      const uint8_t* prow = packed + r * C;
      const T* xr = x + m * K + kg * C;
      float dot = 0.f, xs = 0.f;
      for (int64_t c = 0; c < C; ++c) {
        const float q = (float)((prow[c] >> (BITS * plane)) & MASK);
        const float xv = E::to_f32(xr[c]);
        dot = __builtin_fmaf(q, xv, dot);
        xs += xv;
      }
      const float s = E::to_f32(scale[gr]);
      if constexpr (INT_SHIFT) {
        const float zp = (float)(int8_t) reinterpret_cast<const uint8_t*>(shift_)[gr];
        acc += s * (dot - zp * xs);
      } else {
        const float z = E::to_f32(reinterpret_cast<const T*>(shift_)[gr]);
        acc += s * dot - z * xs;
      }
    }
    if (bias) acc = E::to_f32(E::from_f32(acc)) + E::to_f32(bias[n]);
    y[i] = E::from_f32(acc);
  }
}

static inline int grid_for(int64_t work_items) {
  int64_t g = (work_items + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

template <int ADT, int BDT, int ODT>
static int qbytes_naive_launch(const void* a, const void* b, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K,
                               hipStream_t stream) {
  using T = typename Elem<ODT>::T;
  hipLaunchKernelGGL((qbytes_mm_naive_kernel<ADT, BDT, ODT>), dim3(grid_for(M * N)), dim3(256), 0, stream, a,
                     reinterpret_cast<const uint8_t*>(b), reinterpret_cast<const T*>(s), reinterpret_cast<const T*>(bias),
                     reinterpret_cast<T*>(y), M, N, K);
  return launch_status();
}

template <int ADT, int ODT>
static int qbytes_naive_b(const void* a, const void* b, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K,
                          int b_dtype, hipStream_t stream) {
  switch (b_dtype) {
    case QUANTO_HIP_I8: return qbytes_naive_launch<ADT, QUANTO_HIP_I8, ODT>(a, b, s, bias, y, M, N, K, stream);
    case QUANTO_HIP_F8_E4M3FN: return qbytes_naive_launch<ADT, QUANTO_HIP_F8_E4M3FN, ODT>(a, b, s, bias, y, M, N, K, stream);
    case QUANTO_HIP_F8_E5M2: return qbytes_naive_launch<ADT, QUANTO_HIP_F8_E5M2, ODT>(a, b, s, bias, y, M, N, K, stream);
    case QUANTO_HIP_F8_E4M3FNUZ: return qbytes_naive_launch<ADT, QUANTO_HIP_F8_E4M3FNUZ, ODT>(a, b, s, bias, y, M, N, K, stream);
  }
  return QUANTO_HIP_ENOTSUP;
}

template <int ODT>
static int qbytes_naive_a(const void* a, const void* b, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K,
                          int a_dtype, int b_dtype, hipStream_t stream) {
  switch (a_dtype) {
    case QUANTO_HIP_F32: return qbytes_naive_b<QUANTO_HIP_F32, ODT>(a, b, s, bias, y, M, N, K, b_dtype, stream);
    case QUANTO_HIP_F16: return qbytes_naive_b<QUANTO_HIP_F16, ODT>(a, b, s, bias, y, M, N, K, b_dtype, stream);
    case QUANTO_HIP_BF16: return qbytes_naive_b<QUANTO_HIP_BF16, ODT>(a, b, s, bias, y, M, N, K, b_dtype, stream);
    case QUANTO_HIP_I8: return qbytes_naive_b<QUANTO_HIP_I8, ODT>(a, b, s, bias, y, M, N, K, b_dtype, stream);
    case QUANTO_HIP_F8_E4M3FN: return qbytes_naive_b<QUANTO_HIP_F8_E4M3FN, ODT>(a, b, s, bias, y, M, N, K, b_dtype, stream);
    case QUANTO_HIP_F8_E5M2: return qbytes_naive_b<QUANTO_HIP_F8_E5M2, ODT>(a, b, s, bias, y, M, N, K, b_dtype, stream);
    case QUANTO_HIP_F8_E4M3FNUZ: return qbytes_naive_b<QUANTO_HIP_F8_E4M3FNUZ, ODT>(a, b, s, bias, y, M, N, K, b_dtype, stream);
  }
  return QUANTO_HIP_ENOTSUP;
}

int qbytes_mm_naive(const void* a, const void* b, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int a_dtype,
                    int b_dtype, int out_dtype, hipStream_t stream) {
  switch (out_dtype) {
    case QUANTO_HIP_F32: return qbytes_naive_a<QUANTO_HIP_F32>(a, b, s, bias, y, M, N, K, a_dtype, b_dtype, stream);
    case QUANTO_HIP_F16: return qbytes_naive_a<QUANTO_HIP_F16>(a, b, s, bias, y, M, N, K, a_dtype, b_dtype, stream);
    case QUANTO_HIP_BF16: return qbytes_naive_a<QUANTO_HIP_BF16>(a, b, s, bias, y, M, N, K, a_dtype, b_dtype, stream);
  }
  return QUANTO_HIP_ENOTSUP;
}

template <int DT, int BITS, bool INT_SHIFT>
static int qbits_naive_launch(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y,
                              int64_t M, const PackedGeom& g, hipStream_t stream) {
  using T = typename Elem<DT>::T;
  hipLaunchKernelGGL((qbits_mm_naive_kernel<DT, BITS, INT_SHIFT>), dim3(grid_for(M * g.N)), dim3(256), 0, stream,
                     reinterpret_cast<const T*>(x), packed, reinterpret_cast<const T*>(scale), shift, reinterpret_cast<const T*>(bias),
                     reinterpret_cast<T*>(y), M, g.N, g.K, g.C, g.G, g.row_dim);
  return launch_status();
}

template <int DT>
static int qbits_naive_dt(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y,
                          int64_t M, const PackedGeom& g, bool int_shift, hipStream_t stream) {
  if (g.bits == 4)
    return int_shift ? qbits_naive_launch<DT, 4, true>(x, packed, scale, shift, bias, y, M, g, stream)
                     : qbits_naive_launch<DT, 4, false>(x, packed, scale, shift, bias, y, M, g, stream);
  return int_shift ? qbits_naive_launch<DT, 2, true>(x, packed, scale, shift, bias, y, M, g, stream)
                   : qbits_naive_launch<DT, 2, false>(x, packed, scale, shift, bias, y, M, g, stream);
}

int qbits_mm_naive(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t M,
                   const PackedGeom& g, int dtype, bool int_shift, hipStream_t stream) {
  switch (dtype) {
    case QUANTO_HIP_F32: return qbits_naive_dt<QUANTO_HIP_F32>(x, packed, scale, shift, bias, y, M, g, int_shift, stream);
    case QUANTO_HIP_F16: return qbits_naive_dt<QUANTO_HIP_F16>(x, packed, scale, shift, bias, y, M, g, int_shift, stream);
    case QUANTO_HIP_BF16: return qbits_naive_dt<QUANTO_HIP_BF16>(x, packed, scale, shift, bias, y, M, g, int_shift, stream);
  }
  return QUANTO_HIP_ENOTSUP;
}

}  // namespace qh
