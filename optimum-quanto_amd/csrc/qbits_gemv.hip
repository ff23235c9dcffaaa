// qbits_mm for decode shapes (M <= 8): weight-streaming GEMV over the generic PackedTensor layout.
//
// HBM-bound.  With axis-0 grouping, group size 128 and N even, the packed tensor is simply
// P[N/2][K] bytes: byte (p, k) holds W[p, k] in its low nibble and W[p + N/2, k] in its high
// nibble (tensor/packed.py:24-69 + tensor/grouped.py:17-39), so every packed row is K contiguous
// bytes and one wave streams it with fully coalesced 16-byte loads.  The scale/shift of byte (p,k),
// plane h, is entry (p + h*N/2)*G + k/128.
//
// Work split: one wave owns a K-slab of ITERS*1024 bytes (its x values stay in registers for the
// whole kernel) and loops over packed rows; WPR waves of a block cover one row when K > 4096.
//
// Arithmetic per 32-bit word of packed data (8 weights): 3 shifts + 4 v_and_or_b32 build four
// bf16x2 operands (128+q_a, 128+q_b) - 0x4300|q is exactly 128+q in bf16 - and 4 v_dot2c_f32_bf16
// accumulate them against pre-permuted x pairs; the +128 bias is cancelled by initialising each
// accumulator with -128*sum(x).  Per group: y += scale * dot - shift * sum_group(x), fp32 throughout,
// so the result is the exact-math value of the reference's integers/scales (no bf16 rounding of W).
#include "qh_common.h"

namespace qh {

template <int DT>
struct Dot2;
template <>
struct Dot2<QUANTO_HIP_BF16> {
  static constexpr uint32_t MAGIC = 0x43004300u;  // bf16 128.0 | q
  static constexpr float OFFSET = 128.f;
  static __device__ __forceinline__ float dot(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
  }
};
template <>
struct Dot2<QUANTO_HIP_F16> {
  static constexpr uint32_t MAGIC = 0x64006400u;  // fp16 1024.0 | q
  static constexpr float OFFSET = 1024.f;
  static __device__ __forceinline__ float dot(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
  }
};

template <int DT>
__device__ __forceinline__ float pair_lo(uint32_t p) {
  return Elem<DT>::to_f32(__builtin_bit_cast(typename Elem<DT>::T, (uint16_t)(p & 0xFFFFu)));
}
template <int DT>
__device__ __forceinline__ float pair_hi(uint32_t p) {
  return Elem<DT>::to_f32(__builtin_bit_cast(typename Elem<DT>::T, (uint16_t)(p >> 16)));
}

template <int DT, int MT, int ITERS, bool INT_SHIFT>
__global__ void __launch_bounds__(256)
    qbits_gemv_g128_kernel(const uint16_t* __restrict__ x, const uint8_t* __restrict__ packed, const uint16_t* __restrict__ scale,
                           const void* __restrict__ shift_, const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int N,
                           int K, int wpr /* waves per packed row: 1, 2 or 4 */) {
  using E = Elem<DT>;
  using T = typename E::T;
  using D2 = Dot2<DT>;
  __shared__ float red[2][4][2][MT];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slab = wave % wpr;          // which K-slab this wave owns
  const int row_in_block = wave / wpr;  // which of the block's rows
  const int rpb = 4 / wpr;
  const int P = N >> 1;
  const int G = K >> 7;
  const int half = lane >> 5, j32 = lane & 31;

  // ---- per-wave constants: x pairs, their sums, group sums -----------------------------------
  uint32_t X02[ITERS][MT][4], X13[ITERS][MT][4];
  float dinit[ITERS][MT];
  bool valid[ITERS];
  float xsg[ITERS][MT];  // sum of x over this lane's group (all 8 lanes of the group hold it)
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int k0 = ((slab * ITERS + it) * 64 + lane) * 16;
    valid[it] = k0 < K;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
      if (valid[it]) {
        const uint4* px = reinterpret_cast<const uint4*>(x + (size_t)m * K + k0);
        a = px[0];
        b = px[1];
      }
      const uint32_t pr[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        X02[it][m][d] = __builtin_amdgcn_perm(pr[2 * d + 1], pr[2 * d], 0x05040100u);
        X13[it][m][d] = __builtin_amdgcn_perm(pr[2 * d + 1], pr[2 * d], 0x07060302u);
        s += pair_lo<DT>(pr[2 * d]) + pair_hi<DT>(pr[2 * d]) + pair_lo<DT>(pr[2 * d + 1]) + pair_hi<DT>(pr[2 * d + 1]);
      }
      dinit[it][m] = -D2::OFFSET * s;
      float g = s;
      g += __shfl_xor(g, 1, 64);
      g += __shfl_xor(g, 2, 64);
      g += __shfl_xor(g, 4, 64);
      xsg[it][m] = g;
    }
  }
  // lane j keeps the group sum of local group j32 (= 8*it' + l'/8): fetch it from lane 8*(j32%8), register it' = j32/8
  float XSj[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float v = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const float t = __shfl(xsg[it][m], 8 * (j32 & 7), 64);
      if ((j32 >> 3) == it) v = t;
    }
    XSj[m] = v;
  }
  const int gl = slab * 8 * ITERS + j32;       // global group index handled by this lane's scale/shift slot
  const bool slot_ok = (j32 < 8 * ITERS) && (gl < G);

  const uint8_t* wbase = packed + (size_t)(slab * ITERS) * 1024 + lane * 16;

  auto load_row = [&](int p, uint4 (&W)[ITERS]) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      W[it] = make_uint4(0, 0, 0, 0);
      if (valid[it]) W[it] = *reinterpret_cast<const uint4*>(wbase + (size_t)p * K + it * 1024);
    }
  };

  uint4 Wcur[ITERS], Wnxt[ITERS];
  const int stride = gridDim.x * rpb;
  int p = blockIdx.x * rpb + row_in_block;
  if (p < P) load_row(p, Wcur);
  int parity = 0;
  for (int pbase = blockIdx.x * rpb; pbase < P; pbase += stride, p += stride, parity ^= 1) {
    const bool active = p < P;
    const int pn = p + stride;
    if (pn < P) load_row(pn, Wnxt);

    float acc[2][MT];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[h][m] = 0.f;

    if (active) {
      // scale / shift slots: lanes 0-31 plane 0 (row p), lanes 32-63 plane 1 (row p + N/2)
      const size_t sidx = (size_t)(p + half * P) * G + gl;
      float s_reg = 0.f, z_reg = 0.f;
      if (slot_ok) {
        s_reg = E::to_f32(__builtin_bit_cast(T, scale[sidx]));
        if constexpr (INT_SHIFT)
          z_reg = s_reg * (float)(int8_t) reinterpret_cast<const uint8_t*>(shift_)[sidx];
        else
          z_reg = E::to_f32(__builtin_bit_cast(T, reinterpret_cast<const uint16_t*>(shift_)[sidx]));
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        float dot[2][MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) dot[0][m] = dot[1][m] = dinit[it][m];
        const uint32_t w4[4] = {Wcur[it].x, Wcur[it].y, Wcur[it].z, Wcur[it].w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const uint32_t w = w4[d];
          const uint32_t lo02 = (w & 0x000F000Fu) | D2::MAGIC;
          const uint32_t hi02 = ((w >> 4) & 0x000F000Fu) | D2::MAGIC;
          const uint32_t lo13 = ((w >> 8) & 0x000F000Fu) | D2::MAGIC;
          const uint32_t hi13 = ((w >> 12) & 0x000F000Fu) | D2::MAGIC;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            dot[0][m] = D2::dot(lo02, X02[it][m][d], dot[0][m]);
            dot[1][m] = D2::dot(hi02, X02[it][m][d], dot[1][m]);
            dot[0][m] = D2::dot(lo13, X13[it][m][d], dot[0][m]);
            dot[1][m] = D2::dot(hi13, X13[it][m][d], dot[1][m]);
          }
        }
        const float s0 = __shfl(s_reg, 8 * it + (lane >> 3), 64);
        const float s1 = __shfl(s_reg, 32 + 8 * it + (lane >> 3), 64);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc[0][m] = __builtin_fmaf(s0, dot[0][m], acc[0][m]);
          acc[1][m] = __builtin_fmaf(s1, dot[1][m], acc[1][m]);
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float zt = z_reg * XSj[m];
        acc[0][m] -= half == 0 ? zt : 0.f;
        acc[1][m] -= half == 1 ? zt : 0.f;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[h][m] = wave_sum(acc[h][m]);
    }

    if (wpr == 1) {
      if (active && lane == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int n = p + h * P;
            float r = acc[h][m];
            if (bias) r = E::to_f32(E::from_f32(r)) + E::to_f32(__builtin_bit_cast(T, bias[n]));
            y[(size_t)m * N + n] = __builtin_bit_cast(uint16_t, E::from_f32(r));
          }
      }
    } else {
      if (lane == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int m = 0; m < MT; ++m) red[parity][wave][h][m] = acc[h][m];
      }
      __syncthreads();
      if (active && slab == 0 && lane < 2 * MT) {
        const int h = lane / MT, m = lane % MT;
        float r = 0.f;
        for (int s = 0; s < wpr; ++s) r += red[parity][wave + s][h][m];
        const int n = p + h * P;
        if (bias) r = E::to_f32(E::from_f32(r)) + E::to_f32(__builtin_bit_cast(T, bias[n]));
        y[(size_t)m * N + n] = __builtin_bit_cast(uint16_t, E::from_f32(r));
      }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) Wcur[it] = Wnxt[it];
  }
}

// ------------------------------------------------------------------------------------------------
template <int DT, int MT, bool INT_SHIFT>
static int gemv_launch_iters(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y,
                             int N, int K, hipStream_t stream) {
  const int its_total = (K + 1023) / 1024;
  int wpr = 1, iters = its_total;
  if (its_total > 4) {
    wpr = its_total > 8 ? 4 : 2;
    iters = (its_total + wpr - 1) / wpr;
  }
  if (iters > 4) return QUANTO_HIP_ENOTSUP;
  const int rpb = 4 / wpr;
  const int P = N / 2;
  int grid = (P + rpb - 1) / rpb;
  if (grid > 2048) grid = 2048;
  auto xs = reinterpret_cast<const uint16_t*>(x);
  auto ss = reinterpret_cast<const uint16_t*>(scale);
  auto bs = reinterpret_cast<const uint16_t*>(bias);
  auto ys = reinterpret_cast<uint16_t*>(y);
#define QH_LAUNCH(IT)                                                                                                        \
  hipLaunchKernelGGL((qbits_gemv_g128_kernel<DT, MT, IT, INT_SHIFT>), dim3(grid), dim3(256), 0, stream, xs, packed, ss, shift, bs, \
                     ys, N, K, wpr)
  switch (iters) {
    case 1: QH_LAUNCH(1); break;
    case 2: QH_LAUNCH(2); break;
    case 3: QH_LAUNCH(3); break;
    case 4: QH_LAUNCH(4); break;
  }
#undef QH_LAUNCH
  return launch_status();
}

template <int DT, bool INT_SHIFT>
static int gemv_launch_m(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int M,
                         int N, int K, hipStream_t stream) {
  // rows of x are processed in passes of at most 4 (register-resident x); weights of later passes come from L2/MALL
  int m0 = 0;
  while (m0 < M) {
    const int mt = (M - m0) >= 4 ? 4 : ((M - m0) >= 2 ? 2 : 1);
    const void* xp = reinterpret_cast<const uint16_t*>(x) + (size_t)m0 * K;
    void* yp = reinterpret_cast<uint16_t*>(y) + (size_t)m0 * N;
    int st;
    if (mt == 4)
      st = gemv_launch_iters<DT, 4, INT_SHIFT>(xp, packed, scale, shift, bias, yp, N, K, stream);
    else if (mt == 2)
      st = gemv_launch_iters<DT, 2, INT_SHIFT>(xp, packed, scale, shift, bias, yp, N, K, stream);
    else
      st = gemv_launch_iters<DT, 1, INT_SHIFT>(xp, packed, scale, shift, bias, yp, N, K, stream);
    if (st != QUANTO_HIP_OK) return st;
    m0 += mt;
  }
  return QUANTO_HIP_OK;
}

bool qbits_gemv_supported(int64_t M, const PackedGeom& g, int dtype) {
  return g.bits == 4 && g.C == 128 && (g.N % 2 == 0) && (g.K % 128 == 0) && g.K <= 16384 && M >= 1 &&
         M <= QUANTO_HIP_GEMV_MAX_M && (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) && g.N < (1 << 30);
}

int qbits_mm_gemv(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t M,
                  const PackedGeom& g, int dtype, bool int_shift, hipStream_t stream) {
  if (!qbits_gemv_supported(M, g, dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed)) % 16) return QUANTO_HIP_EALIGN;
  const int Mi = (int)M, N = (int)g.N, K = (int)g.K;
  if (dtype == QUANTO_HIP_BF16)
    return int_shift ? gemv_launch_m<QUANTO_HIP_BF16, true>(x, packed, scale, shift, bias, y, Mi, N, K, stream)
                     : gemv_launch_m<QUANTO_HIP_BF16, false>(x, packed, scale, shift, bias, y, Mi, N, K, stream);
  return int_shift ? gemv_launch_m<QUANTO_HIP_F16, true>(x, packed, scale, shift, bias, y, Mi, N, K, stream)
                   : gemv_launch_m<QUANTO_HIP_F16, false>(x, packed, scale, shift, bias, y, Mi, N, K, stream);
}

}  // namespace qh
