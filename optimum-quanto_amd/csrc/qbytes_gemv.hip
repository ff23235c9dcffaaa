// qbytes_mm for decode shapes (M <= 8): weight-streaming GEMV for int8 / fp8 weights [N, K].
//
// HBM-bound: one wave streams one weight row (K bytes, coalesced 16-byte loads), x stays in
// registers as fp32, products are accumulated in fp32 (int8 -> fp32 by v_cvt_f32_i32 with SDWA byte
// select, fp8 -> fp32 by v_cvt_pk_f32_fp8) and the per-channel scale is applied once in the epilogue:
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

template <int DT, int BDT, int MT, int ITERS>
__global__ void __launch_bounds__(256)
    qbytes_gemv_kernel(const uint16_t* __restrict__ x, const uint8_t* __restrict__ w, const uint16_t* __restrict__ scales,
                       const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int N, int K, int wpr) {
  using E = Elem<DT>;
  using T = typename E::T;
  __shared__ float red[2][4][MT];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slab = wave % wpr;
  const int row_in_block = wave / wpr;
  const int rpb = 4 / wpr;

  float X[ITERS][MT][16];
  bool valid[ITERS];
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
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        X[it][m][2 * q] = E::to_f32(__builtin_bit_cast(T, (uint16_t)(pr[q] & 0xFFFFu)));
        X[it][m][2 * q + 1] = E::to_f32(__builtin_bit_cast(T, (uint16_t)(pr[q] >> 16)));
      }
    }
  }
  const uint8_t* wbase = w + (size_t)(slab * ITERS) * 1024 + lane * 16;
  auto load_row = [&](int n, uint4 (&W)[ITERS]) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      W[it] = make_uint4(0, 0, 0, 0);
      if (valid[it]) W[it] = *reinterpret_cast<const uint4*>(wbase + (size_t)n * K + it * 1024);
    }
  };

  uint4 Wcur[ITERS], Wnxt[ITERS];
  const int stride = gridDim.x * rpb;
  int n = blockIdx.x * rpb + row_in_block;
  if (n < N) load_row(n, Wcur);
  int parity = 0;
  for (int nbase = blockIdx.x * rpb; nbase < N; nbase += stride, n += stride, parity ^= 1) {
    const bool active = n < N;
    if (n + stride < N) load_row(n + stride, Wnxt);
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.f;
    if (active) {
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const uint32_t w4[4] = {Wcur[it].x, Wcur[it].y, Wcur[it].z, Wcur[it].w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          float f[4];
          decode_word<BDT>(w4[d], f);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[m] = __builtin_fmaf(f[e], X[it][m][4 * d + e], acc[m]);
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = wave_sum(acc[m]);
    }
    if (wpr == 1) {
      if (active && lane < MT) {
        float r = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) r = lane == m ? acc[m] : r;
        r *= E::to_f32(__builtin_bit_cast(T, scales[n]));
        if (bias) r = E::to_f32(E::from_f32(r)) + E::to_f32(__builtin_bit_cast(T, bias[n]));
        y[(size_t)lane * N + n] = __builtin_bit_cast(uint16_t, E::from_f32(r));
      }
    } else {
      if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) red[parity][wave][m] = acc[m];
      }
      __syncthreads();
      if (active && slab == 0 && lane < MT) {
        float r = 0.f;
        for (int s = 0; s < wpr; ++s) r += red[parity][wave + s][lane];
        r *= E::to_f32(__builtin_bit_cast(T, scales[n]));
        if (bias) r = E::to_f32(E::from_f32(r)) + E::to_f32(__builtin_bit_cast(T, bias[n]));
        y[(size_t)lane * N + n] = __builtin_bit_cast(uint16_t, E::from_f32(r));
      }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) Wcur[it] = Wnxt[it];
  }
}

template <int DT, int BDT, int MT>
static int launch_iters(const void* x, const void* w, const void* s, const void* bias, void* y, int N, int K, hipStream_t stream) {
  const int its_total = (K + 1023) / 1024;
  int wpr = 1, iters = its_total;
  if (its_total > 4) {
    wpr = its_total > 8 ? 4 : 2;
    iters = (its_total + wpr - 1) / wpr;
  }
  if (iters > 4) return QUANTO_HIP_ENOTSUP;
  const int rpb = 4 / wpr;
  int grid = (N + rpb - 1) / rpb;
  if (grid > 2048) grid = 2048;
  auto xs = reinterpret_cast<const uint16_t*>(x);
  auto ws = reinterpret_cast<const uint8_t*>(w);
  auto ss = reinterpret_cast<const uint16_t*>(s);
  auto bs = reinterpret_cast<const uint16_t*>(bias);
  auto ys = reinterpret_cast<uint16_t*>(y);
#define QH_LAUNCH(IT) \
  hipLaunchKernelGGL((qbytes_gemv_kernel<DT, BDT, MT, IT>), dim3(grid), dim3(256), 0, stream, xs, ws, ss, bs, ys, N, K, wpr)
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
static int launch_m(const void* x, const void* w, const void* s, const void* bias, void* y, int M, int N, int K, hipStream_t stream) {
  int m0 = 0;
  while (m0 < M) {
    const int mt = (M - m0) >= 2 ? 2 : 1;
    const void* xp = reinterpret_cast<const uint16_t*>(x) + (size_t)m0 * K;
    void* yp = reinterpret_cast<uint16_t*>(y) + (size_t)m0 * N;
    const int st = mt == 2 ? launch_iters<DT, BDT, 2>(xp, w, s, bias, yp, N, K, stream)
                           : launch_iters<DT, BDT, 1>(xp, w, s, bias, yp, N, K, stream);
    if (st != QUANTO_HIP_OK) return st;
    m0 += mt;
  }
  return QUANTO_HIP_OK;
}

bool qbytes_gemv_supported(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  const bool bd = b_dtype == QUANTO_HIP_I8 || b_dtype == QUANTO_HIP_F8_E4M3FN || b_dtype == QUANTO_HIP_F8_E5M2;
  return bd && a_dtype == out_dtype && (out_dtype == QUANTO_HIP_BF16 || out_dtype == QUANTO_HIP_F16) && M >= 1 &&
         M <= QUANTO_HIP_GEMV_MAX_M && K % 16 == 0 && K <= 16384 && N < (1 << 30);
}

int qbytes_mm_gemv(const void* a, const void* b, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int a_dtype,
                   int b_dtype, int out_dtype, hipStream_t stream) {
  if (!qbytes_gemv_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16) return QUANTO_HIP_EALIGN;
#define QH_CASE(DT, BDT) return launch_m<DT, BDT>(a, b, s, bias, y, (int)M, (int)N, (int)K, stream)
  if (out_dtype == QUANTO_HIP_BF16) {
    if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_BF16, QUANTO_HIP_I8);
    if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_BF16, QUANTO_HIP_F8_E4M3FN);
    QH_CASE(QUANTO_HIP_BF16, QUANTO_HIP_F8_E5M2);
  }
  if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_F16, QUANTO_HIP_I8);
  if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_F16, QUANTO_HIP_F8_E4M3FN);
  QH_CASE(QUANTO_HIP_F16, QUANTO_HIP_F8_E5M2);
#undef QH_CASE
}

}  // namespace qh
