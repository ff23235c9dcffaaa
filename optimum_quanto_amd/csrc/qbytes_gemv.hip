// qbytes_mm for decode shapes (M <= 8): weight-streaming GEMV for int8 / fp8 weights [N, K].
//
// HBM-bound: coalesced 16-byte loads of the weight rows, x stays in registers as fp32, products are
// accumulated in fp32 (int8 -> fp32 by v_cvt_f32_i32 with SDWA byte select, fp8 -> fp32 by
// v_cvt_pk_f32_fp8) and the per-channel scale is applied once in the epilogue:
// y[m,n] = scale[n] * sum_k x[m,k] * q[n,k]   (library/qbytes_mm.py:25-33 without materialising scale*W).
#include "qh_common.h"

namespace qh {

template <int BDT>
__device__ __forceinline__ void decode_word(uint32_t w, float (&f)[4]);
template <>
__device__ __forceinline__ void decode_word<QUANTO_HIP_I8>(uint32_t w, float (&f)[4]) {
  f[0] = (float)(int8_t)(w & 0xFFu);
  f[1] = (float)(int8_t)((w >> 8) & 0xFFu);
  f[2] = (float)(int8_t)((w >> 16) & 0xFFu);
  f[3] = (float)(int8_t)(w >> 24);
}
template <>
__device__ __forceinline__ void decode_word<QUANTO_HIP_F8_E4M3FN>(uint32_t w, float (&f)[4]) {
  const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false);
  const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
  f[0] = lo.x; f[1] = lo.y; f[2] = hi.x; f[3] = hi.y;
}
template <>
__device__ __forceinline__ void decode_word<QUANTO_HIP_F8_E5M2>(uint32_t w, float (&f)[4]) {
  const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_bf8((int)w, false);
  const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_bf8((int)w, true);
  f[0] = lo.x; f[1] = lo.y; f[2] = hi.x; f[3] = hi.y;
}

template <>
__device__ __forceinline__ void decode_word<QUANTO_HIP_F8_E4M3FNUZ>(uint32_t w, float (&f)[4]) {  // qh_common.h: fn value / 2 + three patched patterns
  const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false);
  const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
  f[0] = fnuz_fix_f32(lo.x * 0.5f, w & 0xFFu);
  f[1] = fnuz_fix_f32(lo.y * 0.5f, (w >> 8) & 0xFFu);
  f[2] = fnuz_fix_f32(hi.x * 0.5f, (w >> 16) & 0xFFu);
  f[3] = fnuz_fix_f32(hi.y * 0.5f, w >> 24);
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
  v += dpp_f<0x140>(v);       // row_mirror
  v += dpp_f<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v += dpp_f<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3 -> lane 63 holds the total
  return v;
}

constexpr int RR8 = 4;  // weight rows per wave pass

// Latency-first layout (same as qbits_gemv.hip): a decode call lasts a few microseconds, so every byte a wave needs is
// requested in its first instructions.  A wave owns K-slabs of 1024 k (slab = wave % wpr, stride wpr, at most ITERS of
// them), keeps that slice of x in registers as fp32 and streams RR8 weight rows at a time: RR8 * ITERS 16-byte loads per
// lane in flight.  Per-slab partial sums are reduced with DPP adds and combined across the wpr waves through LDS.
// Several Linears that share x (q/k/v, gate/up) in one launch: slot 0 is the plain op (MULTI = false: compile-time slot, one
// round of scalar loads); with MULTI a block finds its segment by its first workgroup (qbits_gemv.hip).
constexpr int MAX_SEGS8 = QUANTO_HIP_MAX_MULTI;
struct GemvSegs8 {
  const uint8_t* w[MAX_SEGS8];
  const uint16_t* scales[MAX_SEGS8];
  const uint16_t* bias[MAX_SEGS8];
  uint16_t* y[MAX_SEGS8];
  int N[MAX_SEGS8];
  int first_block[MAX_SEGS8];  // INT_MAX for unused slots
};

template <int DT, int BDT, int MT, int ITERS, bool MULTI = false>
__global__ void __launch_bounds__(256) qbytes_gemv_kernel(const uint16_t* __restrict__ x, const GemvSegs8 segs, int K, int wpr) {
  using E = Elem<DT>;
  using T = typename E::T;
  __shared__ float red[4][RR8][MT];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slab0 = wave % wpr;
  const int rgroup = wave / wpr;
  int seg = 0, block = blockIdx.x;
  if constexpr (MULTI) {
#pragma unroll
    for (int i = 1; i < MAX_SEGS8; ++i) seg += (int)blockIdx.x >= segs.first_block[i];
    block -= segs.first_block[seg];
  }
  const uint8_t* __restrict__ w = segs.w[seg];
  const uint16_t* __restrict__ scales = segs.scales[seg];
  const uint16_t* __restrict__ bias = segs.bias[seg];
  uint16_t* __restrict__ y = segs.y[seg];
  const int N = segs.N[seg];
  const int n0 = (block * (4 / wpr) + rgroup) * RR8;

  // ---- 1. request everything: weights first, then the x slice ----------------------------------------------------------
  int k0[ITERS];
  bool valid[ITERS];
  uint4 W[RR8][ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    k0[it] = ((slab0 + it * wpr) * 64 + lane) * 16;
    valid[it] = k0[it] < K;
    k0[it] = valid[it] ? k0[it] : 0;  // out-of-range slabs read (and then ignore) the start of the row
#pragma unroll
    for (int r = 0; r < RR8; ++r) {
      const int n = n0 + r < N ? n0 + r : N - 1;  // clamped, unconditional: rows beyond N are computed on duplicates, never stored
      W[r][it] = *reinterpret_cast<const uint4*>(w + (size_t)n * K + k0[it]);
    }
  }
  float X[ITERS][MT][16];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const uint4* px = reinterpret_cast<const uint4*>(x + (size_t)m * K + k0[it]);
      const uint4 a = px[0], b = px[1];
      const uint32_t keep = valid[it] ? 0xFFFFFFFFu : 0u;  // x = 0 beyond K: such a slab contributes exactly 0
      const uint32_t pr[8] = {a.x & keep, a.y & keep, a.z & keep, a.w & keep, b.x & keep, b.y & keep, b.z & keep, b.w & keep};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        X[it][m][2 * q] = E::to_f32(__builtin_bit_cast(T, (uint16_t)(pr[q] & 0xFFFFu)));
        X[it][m][2 * q + 1] = E::to_f32(__builtin_bit_cast(T, (uint16_t)(pr[q] >> 16)));
      }
    }
  }

  // ---- 2. products: exact value of every int8 / fp8 weight in fp32, fp32 fma --------------------------------------------------
  float acc[RR8][MT];
#pragma unroll
  for (int r = 0; r < RR8; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[r][m] = 0.f;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < RR8; ++r) {
      const uint32_t w4[4] = {W[r][it].x, W[r][it].y, W[r][it].z, W[r][it].w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float f[4];
        decode_word<BDT>(w4[d], f);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[r][m] = __builtin_fmaf(f[e], X[it][m][4 * d + e], acc[r][m]);
      }
    }
  }

  // ---- 3. reduce over lanes (DPP), over the wpr waves (LDS), scale, store --------------------------------------------------------
#pragma unroll
  for (int r = 0; r < RR8; ++r)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[r][m] = wave_sum_lane63(acc[r][m]);
  if (lane == 63) {
#pragma unroll
    for (int r = 0; r < RR8; ++r)
#pragma unroll
      for (int m = 0; m < MT; ++m) red[wave][r][m] = acc[r][m];
  }
  __syncthreads();
  if (slab0 == 0 && lane < RR8 * MT) {
    const int r = lane / MT, m = lane % MT;
    const int n = n0 + r;
    if (n < N) {
      float v = 0.f;
      for (int s = 0; s < wpr; ++s) v += red[wave + s][r][m];
      v *= E::to_f32(__builtin_bit_cast(T, scales[n]));
      asm volatile("" : "+v"(v));  // product rounded to fp32 first, with and without bias (no single-rounding v_fma_mixlo_f16)
      if (bias) v = E::to_f32(E::from_f32(v)) + E::to_f32(__builtin_bit_cast(T, bias[n]));
      y[(size_t)m * N + n] = __builtin_bit_cast(uint16_t, E::from_f32(v));
    }
  }
}

struct GemvProblem8 {
  int nseg;
  const void* w[MAX_SEGS8];
  const void* s[MAX_SEGS8];
  const void* bias[MAX_SEGS8];
  void* y[MAX_SEGS8];
  int N[MAX_SEGS8];
};

template <int DT, int BDT, int MT>
static int launch_iters(const void* x, const GemvProblem8& pb, int m0, int K, hipStream_t stream) {
  const int nslab = (K + 1023) / 1024;
  const int wpr = nslab >= 3 ? 4 : nslab;  // 1, 2 or 4 waves per row group
  const int iters = (nslab + wpr - 1) / wpr;
  if (iters > 4) return QUANTO_HIP_ENOTSUP;
  const int rows_per_block = RR8 * (4 / wpr);
  GemvSegs8 segs;
  int grid = 0;
  for (int i = 0; i < MAX_SEGS8; ++i) {
    const int j = i < pb.nseg ? i : 0;  // unused slots repeat segment 0 and are never selected
    segs.w[i] = reinterpret_cast<const uint8_t*>(pb.w[j]);
    segs.scales[i] = reinterpret_cast<const uint16_t*>(pb.s[j]);
    segs.bias[i] = reinterpret_cast<const uint16_t*>(pb.bias[j]);
    segs.y[i] = reinterpret_cast<uint16_t*>(pb.y[j]) + (size_t)m0 * pb.N[j];
    segs.N[i] = pb.N[j];
    segs.first_block[i] = i < pb.nseg ? grid : 0x7FFFFFFF;
    if (i < pb.nseg) grid += (pb.N[i] + rows_per_block - 1) / rows_per_block;
  }
  auto xs = reinterpret_cast<const uint16_t*>(x) + (size_t)m0 * K;
#define QH_LAUNCH(IT)                                                                                                           \
  if (pb.nseg > 1)                                                                                                              \
    hipLaunchKernelGGL((qbytes_gemv_kernel<DT, BDT, MT, IT, true>), dim3(grid), dim3(256), 0, stream, xs, segs, K, wpr);         \
  else                                                                                                                          \
    hipLaunchKernelGGL((qbytes_gemv_kernel<DT, BDT, MT, IT, false>), dim3(grid), dim3(256), 0, stream, xs, segs, K, wpr)
  switch (iters) {
    case 1: QH_LAUNCH(1); break;
    case 2: QH_LAUNCH(2); break;
    case 3: QH_LAUNCH(3); break;
    case 4: QH_LAUNCH(4); break;
  }
#undef QH_LAUNCH
  return launch_status();
}

template <int DT, int BDT>
static int launch_m(const void* x, const GemvProblem8& pb, int M, int K, hipStream_t stream) {
  int m0 = 0;
  while (m0 < M) {  // x lives in registers as fp32 (16 VGPRs per slab and row): two rows per pass, later passes hit the MALL
    const int mt = (M - m0) >= 2 ? 2 : 1;
    const int st = mt == 2 ? launch_iters<DT, BDT, 2>(x, pb, m0, K, stream) : launch_iters<DT, BDT, 1>(x, pb, m0, K, stream);
    if (st != QUANTO_HIP_OK) return st;
    m0 += mt;
  }
  return QUANTO_HIP_OK;
}

static int gemv8_dispatch(const void* a, const GemvProblem8& pb, int M, int K, int b_dtype, int out_dtype, hipStream_t stream) {
#define QH_CASE(DT, BDT) return launch_m<DT, BDT>(a, pb, M, K, stream)
  if (out_dtype == QUANTO_HIP_BF16) {
    if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_BF16, QUANTO_HIP_I8);
    if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_BF16, QUANTO_HIP_F8_E4M3FN);
    if (b_dtype == QUANTO_HIP_F8_E4M3FNUZ) QH_CASE(QUANTO_HIP_BF16, QUANTO_HIP_F8_E4M3FNUZ);
    QH_CASE(QUANTO_HIP_BF16, QUANTO_HIP_F8_E5M2);
  }
  if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_F16, QUANTO_HIP_I8);
  if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_F16, QUANTO_HIP_F8_E4M3FN);
  if (b_dtype == QUANTO_HIP_F8_E4M3FNUZ) QH_CASE(QUANTO_HIP_F16, QUANTO_HIP_F8_E4M3FNUZ);
  QH_CASE(QUANTO_HIP_F16, QUANTO_HIP_F8_E5M2);
#undef QH_CASE
}

bool qbytes_gemv_supported(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  const bool bd = b_dtype == QUANTO_HIP_I8 || b_dtype == QUANTO_HIP_F8_E4M3FN || b_dtype == QUANTO_HIP_F8_E5M2 || b_dtype == QUANTO_HIP_F8_E4M3FNUZ;
  return bd && a_dtype == out_dtype && (out_dtype == QUANTO_HIP_BF16 || out_dtype == QUANTO_HIP_F16) && M >= 1 &&
         M <= QUANTO_HIP_GEMV_MAX_M && K % 16 == 0 && K <= 16384 && N < (1 << 30);
}

int qbytes_mm_gemv(const void* a, const void* b, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int a_dtype,
                   int b_dtype, int out_dtype, hipStream_t stream) {
  if (!qbytes_gemv_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16) return QUANTO_HIP_EALIGN;
  GemvProblem8 pb{};
  pb.nseg = 1;
  pb.w[0] = b;
  pb.s[0] = s;
  pb.bias[0] = bias;
  pb.y[0] = y;
  pb.N[0] = (int)N;
  return gemv8_dispatch(a, pb, (int)M, (int)K, b_dtype, out_dtype, stream);
}

// up to QUANTO_HIP_MAX_MULTI weights that share the activation, ONE launch (every member must pass qbytes_gemv_supported)
int qbytes_mm_gemv_multi(const void* a, int nseg, const void* const* b, const void* const* s, const void* const* bias, void* const* y,
                         const int64_t* N, int64_t M, int64_t K, int b_dtype, int out_dtype, hipStream_t stream) {
  if (nseg < 1 || nseg > MAX_SEGS8) return QUANTO_HIP_EINVAL;
  GemvProblem8 pb{};
  pb.nseg = nseg;
  uintptr_t align = reinterpret_cast<uintptr_t>(a);
  for (int i = 0; i < nseg; ++i) {
    pb.w[i] = b[i];
    pb.s[i] = s[i];
    pb.bias[i] = bias ? bias[i] : nullptr;
    pb.y[i] = y[i];
    pb.N[i] = (int)N[i];
    align |= reinterpret_cast<uintptr_t>(b[i]);
  }
  if (align % 16) return QUANTO_HIP_EALIGN;
  return gemv8_dispatch(a, pb, (int)M, (int)K, b_dtype, out_dtype, stream);
}

}  // namespace qh
