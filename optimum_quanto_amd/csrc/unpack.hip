// quanto::unpack and the fused unpack+dequantize for axis-0 int4/int2 weights.
//
// HBM-bound byte work: each thread moves 16 packed bytes with one dwordx4 load and writes one
// dwordx4 per bit-plane (unpack) or 16 dequantized elements per plane (dequantize).  The reference
// HIP kernel (library/extensions/hip/unpack.cu:33-97) moves 1 byte per thread on the legacy
// stream; this one is vectorised and runs on the caller's stream.
#include "qh_common.h"

namespace qh {

template <int BITS>
__global__ void __launch_bounds__(256) unpack_vec16_kernel(const uint4* __restrict__ in, uint8_t* __restrict__ out,
                                                           int64_t n16, int64_t plane_stride) {
  constexpr int VPI = 8 / BITS;
  constexpr uint32_t MASK = BITS == 4 ? 0x0F0F0F0Fu : 0x03030303u;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 v = in[i];
#pragma unroll
    for (int p = 0; p < VPI; ++p) {
      uint4 o;
      o.x = (v.x >> (BITS * p)) & MASK;
      o.y = (v.y >> (BITS * p)) & MASK;
      o.z = (v.z >> (BITS * p)) & MASK;
      o.w = (v.w >> (BITS * p)) & MASK;
      *reinterpret_cast<uint4*>(out + p * plane_stride + i * 16) = o;
    }
  }
}

template <int BITS>
__global__ void __launch_bounds__(256) unpack_scalar_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                            int64_t n) {
  constexpr int VPI = 8 / BITS;
  constexpr uint32_t MASK = (1u << BITS) - 1u;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t v = in[i];
#pragma unroll
    for (int p = 0; p < VPI; ++p) out[p * n + i] = (uint8_t)((v >> (BITS * p)) & MASK);
  }
}

// ---- fused dequantize (axis 0) -----------------------------------------------------------------
// One thread handles VEC consecutive packed bytes of one packed row; for every bit-plane it owns
// the VEC weights W[n, kg*C + c .. c+VEC) of grouped row gr = r + plane*row_dim.
#pragma clang fp contract(off)  // the reference rounds scale*q before subtracting the shift
template <int DT, int BITS, int VEC, bool INT_SHIFT>
__global__ void __launch_bounds__(256)
    dequantize_qbits_kernel(const uint8_t* __restrict__ packed, const typename Elem<DT>::T* __restrict__ scale,
                            const void* __restrict__ shift_, typename Elem<DT>::T* __restrict__ out, int64_t C,
                            int64_t G, int64_t R, int64_t row_dim, int64_t K) {
  using E = Elem<DT>;
  using T = typename E::T;
  constexpr int VPI = 8 / BITS;
  constexpr uint32_t MASK = (1u << BITS) - 1u;
  const int64_t chunks_per_row = C / VEC;
  const int64_t total = row_dim * chunks_per_row;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / chunks_per_row;
    const int64_t c = (i - r * chunks_per_row) * VEC;
    uint8_t bytes[VEC];
    if constexpr (VEC == 16) {
      *reinterpret_cast<uint4*>(bytes) = *reinterpret_cast<const uint4*>(packed + r * C + c);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) bytes[j] = packed[r * C + c + j];
    }
#pragma unroll
    for (int p = 0; p < VPI; ++p) {
      const int64_t gr = r + p * row_dim;
      if (gr >= R) break;
      const int64_t n = gr / G, kg = gr - n * G;
      const float s = E::to_f32(scale[gr]);
      T* dst = out + n * K + kg * C + c;
      T vals[VEC];
      if constexpr (INT_SHIFT) {
        // data.to(int8) - shift.to(int8), then scale * data : one rounding (tensor/qbits.py:35-42)
        const int zp = (int)(int8_t) reinterpret_cast<const uint8_t*>(shift_)[gr];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const int q = (int)((bytes[j] >> (BITS * p)) & MASK);
          vals[j] = E::from_f32(s * (float)(int8_t)(q - zp));
        }
      } else {
        const float z = E::to_f32(reinterpret_cast<const T*>(shift_)[gr]);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float q = (float)((bytes[j] >> (BITS * p)) & MASK);
          const float t = E::to_f32(E::from_f32(s * q));  // first rounding
          vals[j] = E::from_f32(t - z);                   // second rounding (tensor/qbits.py:43-45)
        }
      }
      if constexpr (VEC == 16 && sizeof(T) == 2) {
        reinterpret_cast<uint4*>(dst)[0] = reinterpret_cast<const uint4*>(vals)[0];
        reinterpret_cast<uint4*>(dst)[1] = reinterpret_cast<const uint4*>(vals)[1];
      } else if constexpr (VEC == 16 && sizeof(T) == 4) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) reinterpret_cast<uint4*>(dst)[q4] = reinterpret_cast<const uint4*>(vals)[q4];
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) dst[j] = vals[j];
      }
    }
  }
}

static inline int grid_for(int64_t work_items, int block = 256) {
  int64_t g = (work_items + block - 1) / block;
  const int64_t cap = 256 * 8;  // 256 CUs x 8 blocks, grid-stride beyond
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

int unpack_dispatch(const uint8_t* packed, uint8_t* unpacked, int64_t n, int bits, hipStream_t stream) {
  if (n == 0) return QUANTO_HIP_OK;
  const bool vec = (n % 16 == 0) && ((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(unpacked)) % 16 == 0);
  if (vec) {
    const int64_t n16 = n / 16;
    const int grid = grid_for(n16);
    if (bits == 4)
      hipLaunchKernelGGL(unpack_vec16_kernel<4>, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const uint4*>(packed), unpacked, n16, n);
    else
      hipLaunchKernelGGL(unpack_vec16_kernel<2>, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const uint4*>(packed), unpacked, n16, n);
  } else {
    const int grid = grid_for(n);
    if (bits == 4)
      hipLaunchKernelGGL(unpack_scalar_kernel<4>, dim3(grid), dim3(256), 0, stream, packed, unpacked, n);
    else
      hipLaunchKernelGGL(unpack_scalar_kernel<2>, dim3(grid), dim3(256), 0, stream, packed, unpacked, n);
  }
  return launch_status();
}

template <int DT, int BITS, bool INT_SHIFT>
static int dequantize_launch(const uint8_t* packed, const void* scale, const void* shift, void* out, const PackedGeom& g,
                             hipStream_t stream) {
  using T = typename Elem<DT>::T;
  const bool vec = (g.C % 16 == 0) && (reinterpret_cast<uintptr_t>(packed) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(out) % 16 == 0);
  if (vec) {
    const int64_t total = g.row_dim * (g.C / 16);
    hipLaunchKernelGGL((dequantize_qbits_kernel<DT, BITS, 16, INT_SHIFT>), dim3(grid_for(total)), dim3(256), 0, stream, packed,
                       reinterpret_cast<const T*>(scale), shift, reinterpret_cast<T*>(out), g.C, g.G, g.R, g.row_dim, g.K);
  } else {
    const int64_t total = g.row_dim * g.C;
    hipLaunchKernelGGL((dequantize_qbits_kernel<DT, BITS, 1, INT_SHIFT>), dim3(grid_for(total)), dim3(256), 0, stream, packed,
                       reinterpret_cast<const T*>(scale), shift, reinterpret_cast<T*>(out), g.C, g.G, g.R, g.row_dim, g.K);
  }
  return launch_status();
}

template <int DT>
static int dequantize_dt(const uint8_t* packed, const void* scale, const void* shift, void* out, const PackedGeom& g,
                         bool int_shift, hipStream_t stream) {
  if (g.bits == 4)
    return int_shift ? dequantize_launch<DT, 4, true>(packed, scale, shift, out, g, stream)
                     : dequantize_launch<DT, 4, false>(packed, scale, shift, out, g, stream);
  return int_shift ? dequantize_launch<DT, 2, true>(packed, scale, shift, out, g, stream)
                   : dequantize_launch<DT, 2, false>(packed, scale, shift, out, g, stream);
}

int dequantize_qbits_dispatch(const uint8_t* packed, const void* scale, const void* shift, void* out, const PackedGeom& g,
                              int dtype, bool int_shift, hipStream_t stream) {
  switch (dtype) {
    case QUANTO_HIP_F32: return dequantize_dt<QUANTO_HIP_F32>(packed, scale, shift, out, g, int_shift, stream);
    case QUANTO_HIP_F16: return dequantize_dt<QUANTO_HIP_F16>(packed, scale, shift, out, g, int_shift, stream);
    case QUANTO_HIP_BF16: return dequantize_dt<QUANTO_HIP_BF16>(packed, scale, shift, out, g, int_shift, stream);
  }
  return QUANTO_HIP_ENOTSUP;
}

}  // namespace qh
