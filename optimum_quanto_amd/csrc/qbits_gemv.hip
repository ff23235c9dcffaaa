// qbits_mm for decode shapes (M <= 64, in passes of up to 8 rows of x): weight-streaming GEMV over the generic PackedTensor layout.
//
// HBM-bound.  With axis-0 grouping, group size 128 and N even, the packed tensor is simply
// P[N/2][K] bytes: byte (p, k) holds W[p, k] in its low nibble and W[p + N/2, k] in its high
// nibble (tensor/packed.py:24-69 + tensor/grouped.py:17-39), so every packed row is K contiguous
// bytes streamed with fully coalesced 16-byte loads.  The scale/shift of byte (p,k), plane h, is
// entry (p + h*N/2)*G + k/128.
//
// Work split (v2, latency-first: a decode call lasts a few microseconds, so everything a wave needs
// is requested in its first instructions):
//   * a wave owns K-slabs of 1024 byte-columns (slab = wave, wave+4, ...; at most ITERS of them) and
//     keeps only that slice of x in registers (8 VGPRs per slab and row of x);
//   * a block of 4 waves covers RR=4 packed rows per iteration: RR*ITERS 16-byte weight loads per lane
//     are in flight at once; per-slab partial results are combined through 64 bytes of LDS;
//   * scale/shift: lane l loads the 2-byte entries of row (l&3), group (l>>3) once per slab - four
//     tiny coalesced loads - and the four rows are fanned out inside each quad with DPP quad_perm.
//
// Arithmetic per 32-bit word of packed data (8 weights): 3 shifts + 4 v_and_or_b32 build four
// bf16x2 operands (128+q_a, 128+q_b) - 0x4300|q is exactly 128+q in bf16 - and 4 v_dot2c_f32_bf16
// accumulate them against pre-permuted x pairs; the +128 bias is cancelled by initialising each
// accumulator with -128*sum(x).  Per lane and group: y += scale * dot - shift * sum(x), fp32
// throughout, so the result is the exact-math value of the reference's integers/scales (no bf16
// rounding of W).  Wave reductions use DPP adds (no LDS traffic).
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

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// sum over the 64 lanes; only lane 63 holds the total
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v += dpp_f<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);       // row_half_mirror
  v += dpp_f<0x140>(v);       // row_mirror  -> every lane of a 16-lane row holds the row sum
  v += dpp_f<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v += dpp_f<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3 -> lane 63 holds the total
  return v;
}
template <int R>
__device__ __forceinline__ float quad_bcast(float v) {
  constexpr int ctrl = R | (R << 2) | (R << 4) | (R << 6);
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true));
}

constexpr int RR = 4;  // packed rows per wave pass

template <int DT, int MT, int ITERS, bool INT_SHIFT>
__global__ void __launch_bounds__(256)
    qbits_gemv_g128_kernel(const uint16_t* __restrict__ x, const uint8_t* __restrict__ packed, const uint16_t* __restrict__ scale,
                           const void* __restrict__ shift_, const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int N,
                           int K, int wpr /* waves cooperating on one row group: 1, 2 or 4 */) {
  using E = Elem<DT>;
  using T = typename E::T;
  using D2 = Dot2<DT>;
  __shared__ float red[4][RR][2][MT];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slab0 = wave % wpr;   // first K-slab of this wave; further slabs at stride wpr
  const int rgroup = wave / wpr;  // which row group of the block
  const int groups_per_block = 4 / wpr;
  const int P = N >> 1;
  const int G = K >> 7;
  const int p0 = (blockIdx.x * groups_per_block + rgroup) * RR;

  // ---- 1. request everything this wave will touch: weights, scales/shifts, x slice ---------------
  int k0[ITERS];
  bool valid[ITERS];
  uint4 W[RR][ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    k0[it] = ((slab0 + it * wpr) * 64 + lane) * 16;
    valid[it] = k0[it] < K;
    k0[it] = valid[it] ? k0[it] : 0;  // out-of-range slabs read (and then ignore) the start of the row
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      // unconditional, clamped loads: rows beyond P are computed on duplicate data and never stored
      const int pr = p0 + r < P ? p0 + r : P - 1;
      W[r][it] = *reinterpret_cast<const uint4*>(packed + (size_t)pr * K + k0[it]);
    }
  }
  // lane l fetches the entries of row (l & 3), group of its 16 bytes, both planes
  float sq[ITERS][2], zq[ITERS][2];
  {
    const int rq = p0 + (lane & 3) < P ? p0 + (lane & 3) : P - 1;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const size_t idx = (size_t)(rq + h * P) * G + (k0[it] >> 7);
        sq[it][h] = E::to_f32(__builtin_bit_cast(T, scale[idx]));
        if constexpr (INT_SHIFT)
          zq[it][h] = sq[it][h] * (float)(int8_t) reinterpret_cast<const uint8_t*>(shift_)[idx];
        else
          zq[it][h] = E::to_f32(__builtin_bit_cast(T, reinterpret_cast<const uint16_t*>(shift_)[idx]));
      }
    }
  }
  uint32_t X02[ITERS][MT][4], X13[ITERS][MT][4];
  float xs[ITERS][MT];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const uint4* px = reinterpret_cast<const uint4*>(x + (size_t)m * K + k0[it]);
      const uint4 a = px[0], b = px[1];
      const uint32_t keep = valid[it] ? 0xFFFFFFFFu : 0u;  // x = 0 beyond K: such a slab contributes exactly 0
      const uint32_t pr[8] = {a.x & keep, a.y & keep, a.z & keep, a.w & keep, b.x & keep, b.y & keep, b.z & keep, b.w & keep};
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        X02[it][m][d] = __builtin_amdgcn_perm(pr[2 * d + 1], pr[2 * d], 0x05040100u);
        X13[it][m][d] = __builtin_amdgcn_perm(pr[2 * d + 1], pr[2 * d], 0x07060302u);
        s += pair_lo<DT>(pr[2 * d]) + pair_hi<DT>(pr[2 * d]) + pair_lo<DT>(pr[2 * d + 1]) + pair_hi<DT>(pr[2 * d + 1]);
      }
      xs[it][m] = s;
    }
  }

  // ---- 2. per row: dot products, scale, shift ------------------------------------------------------
  // v_and_or_b32 is VOP3: no literal operands on gfx9, at most one SGPR.  Keep the mask in an SGPR and the magic
  // exponent in a VGPR, opaque to the constant folder, so that (w & mask) | magic is ONE instruction.
  uint32_t kmask = 0x000F000Fu, kmagic = D2::MAGIC;
  asm volatile("" : "+s"(kmask));
  asm volatile("" : "+v"(kmagic));
  float acc[RR][2][MT];
#pragma unroll
  for (int r = 0; r < RR; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[r][h][m] = 0.f;

#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    float s_r[RR][2], z_r[RR][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      s_r[0][h] = quad_bcast<0>(sq[it][h]);
      s_r[1][h] = quad_bcast<1>(sq[it][h]);
      s_r[2][h] = quad_bcast<2>(sq[it][h]);
      s_r[3][h] = quad_bcast<3>(sq[it][h]);
      z_r[0][h] = quad_bcast<0>(zq[it][h]);
      z_r[1][h] = quad_bcast<1>(zq[it][h]);
      z_r[2][h] = quad_bcast<2>(zq[it][h]);
      z_r[3][h] = quad_bcast<3>(zq[it][h]);
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      float dot[2][MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) dot[0][m] = dot[1][m] = -D2::OFFSET * xs[it][m];
      const uint32_t w4[4] = {W[r][it].x, W[r][it].y, W[r][it].z, W[r][it].w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const uint32_t w = w4[d];
        const uint32_t lo02 = (w & kmask) | kmagic;
        const uint32_t hi02 = ((w >> 4) & kmask) | kmagic;
        const uint32_t lo13 = ((w >> 8) & kmask) | kmagic;
        const uint32_t hi13 = ((w >> 12) & kmask) | kmagic;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          dot[0][m] = D2::dot(lo02, X02[it][m][d], dot[0][m]);
          dot[1][m] = D2::dot(hi02, X02[it][m][d], dot[1][m]);
          dot[0][m] = D2::dot(lo13, X13[it][m][d], dot[0][m]);
          dot[1][m] = D2::dot(hi13, X13[it][m][d], dot[1][m]);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[r][h][m] += s_r[r][h] * dot[h][m] - z_r[r][h] * xs[it][m];
    }
  }

  // ---- 3. reduce over lanes (DPP), over the wpr waves (LDS), store -----------------------------------
#pragma unroll
  for (int r = 0; r < RR; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[r][h][m] = wave_sum_lane63(acc[r][h][m]);
  if (lane == 63) {
#pragma unroll
    for (int r = 0; r < RR; ++r)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = 0; m < MT; ++m) red[wave][r][h][m] = acc[r][h][m];
  }
  __syncthreads();
  if (slab0 == 0 && lane < RR * 2 * MT) {
    const int r = lane / (2 * MT), h = (lane / MT) & 1, m = lane % MT;
    const int p = p0 + r;
    if (p < P) {
      float v = 0.f;
      for (int s = 0; s < wpr; ++s) v += red[wave + s][r][h][m];
      const int n = p + h * P;
      if (bias) v = E::to_f32(E::from_f32(v)) + E::to_f32(__builtin_bit_cast(T, bias[n]));
      y[(size_t)m * N + n] = __builtin_bit_cast(uint16_t, E::from_f32(v));
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <int DT, int MT, bool INT_SHIFT>
static int gemv_launch_iters(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y,
                             int N, int K, hipStream_t stream) {
  const int nslab = (K + 1023) / 1024;
  const int wpr = nslab >= 3 ? 4 : nslab;  // 1, 2 or 4 waves per row group
  const int iters = (nslab + wpr - 1) / wpr;
  if (iters > 4) return QUANTO_HIP_ENOTSUP;
  const int rows_per_block = RR * (4 / wpr);
  const int P = N / 2;
  const int grid = (P + rows_per_block - 1) / rows_per_block;
  auto xs = reinterpret_cast<const uint16_t*>(x);
  auto ss = reinterpret_cast<const uint16_t*>(scale);
  auto bs = reinterpret_cast<const uint16_t*>(bias);
  auto ys = reinterpret_cast<uint16_t*>(y);
#define QH_LAUNCH(IT)                                                                                                        \
  hipLaunchKernelGGL((qbits_gemv_g128_kernel<DT, MT, IT, INT_SHIFT>), dim3(grid), dim3(256), 0, stream, xs, packed, ss, shift, bs, \
                     ys, N, K, wpr)
  if constexpr (MT == 8) {
    switch (iters) {
      case 1: QH_LAUNCH(1); break;
      case 2: QH_LAUNCH(2); break;
      default: return QUANTO_HIP_ENOTSUP;
    }
  } else {
    switch (iters) {
      case 1: QH_LAUNCH(1); break;
      case 2: QH_LAUNCH(2); break;
      case 3: QH_LAUNCH(3); break;
      case 4: QH_LAUNCH(4); break;
    }
  }
#undef QH_LAUNCH
  return launch_status();
}

template <int DT, bool INT_SHIFT>
static int gemv_launch_m(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int M,
                         int N, int K, hipStream_t stream) {
  // Rows of x are processed in passes of at most 8 (x lives in registers: 8 VGPRs per slab and row); the weights of
  // later passes come from the Infinity Cache (a Linear's packed weight is 8-30 MB).  8 rows per pass need
  // iters <= 2 slabs per wave (K <= 8192) to stay inside the register file.
  const int nslab = (K + 1023) / 1024;
  const int iters = (nslab + (nslab >= 3 ? 4 : nslab) - 1) / (nslab >= 3 ? 4 : nslab);
  const int mt_max = iters <= 2 ? 8 : 4;
  int m0 = 0;
  while (m0 < M) {
    const int left = M - m0;
    const int mt = (left >= 8 && mt_max >= 8) ? 8 : (left >= 4 ? 4 : (left >= 2 ? 2 : 1));
    const void* xp = reinterpret_cast<const uint16_t*>(x) + (size_t)m0 * K;
    void* yp = reinterpret_cast<uint16_t*>(y) + (size_t)m0 * N;
    int st;
    if (mt == 8)
      st = gemv_launch_iters<DT, 8, INT_SHIFT>(xp, packed, scale, shift, bias, yp, N, K, stream);
    else if (mt == 4)
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
         M <= QUANTO_HIP_GEMV_MAX_M_QBITS && (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) && g.N < (1 << 30);
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
