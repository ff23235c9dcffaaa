// LDS-tiled MFMA GEMM for quantized weights (M > 8): y[M,N] = x[M,K] @ dequant(W)[N,K]^T.
//
// One kernel template serves the three weight formats of the hot path:
//   W8  int8 / fp8 weights [N,K] with a per-channel scale (quanto::qbytes_mm),
//   W4  int4 weights in the generic PackedTensor layout with per-group scale+shift (quanto::qbits_mm).
// Design (v1: 128x128x64 tile, 4 waves as 2x2, v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate):
//   * activations: global -> registers -> LDS (row-major [128][64], 16-byte chunks XOR-swizzled by row&7 so the
//     ds_read_b128 fragment reads are bank-conflict free);
//   * weights: global (1 byte per weight, or 1 byte per 2 weights) -> registers -> converted IN REGISTERS to the
//     activation dtype -> the same swizzled LDS image.  The conversion is exact (int8, fp8 and 128+q are all
//     representable in bf16/fp16), so the MFMA sums exact products;
//   * scales never touch the operands: W8 applies scale[n] to the fp32 accumulator in the epilogue; W4 keeps a
//     per-group accumulator and folds  acc += s[n,g]*acc_g - (z[n,g] + 128*s[n,g]) * XS[m,g]  every group,
//     where XS[m,g] = sum_{k in g} x[m,k] comes from a small pre-kernel (workspace).  The result is the
//     exact-math product of the reference's integers and scales up to fp32 accumulation order.
// Weight tiles are read from HBM exactly once per (m-tile, n-tile); the dense dequantized weight the reference
// materialises on every call (library/qbytes_mm.py:25-33, tensor/qbits.py:27-49) never exists.
#include "qh_common.h"

namespace qh {

enum WFmt { W_I8 = 0, W_F8E4M3 = 1, W_F8E5M2 = 2, W_I4 = 3 };

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // one operand tile in LDS (16 KiB)

__device__ __forceinline__ int lds_off(int row, int kc) { return row * (BK * 2) + ((kc ^ (row & 7)) << 4); }

template <int DT>
struct Mma;
template <>
struct Mma<QUANTO_HIP_BF16> {
  using V8 = bf16x8;
  static __device__ __forceinline__ f32x4 run(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    bf16x2 r;
    r.x = (__bf16)a;
    r.y = (__bf16)b;
    return __builtin_bit_cast(uint32_t, r);
  }
  static constexpr uint32_t MAGIC = 0x43004300u;
  static constexpr float OFFSET = 128.f;
};
template <>
struct Mma<QUANTO_HIP_F16> {
  using V8 = f16x8;
  static __device__ __forceinline__ f32x4 run(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));  // exact for every int8 / fp8 value
  }
  static constexpr uint32_t MAGIC = 0x64006400u;
  static constexpr float OFFSET = 1024.f;
};

// 16 one-byte weights -> 16 elements of the activation dtype (two 16-byte LDS chunks)
template <int DT, int FMT>
__device__ __forceinline__ void convert16(const uint4& w, uint4& c0, uint4& c1) {
  const uint32_t in[4] = {w.x, w.y, w.z, w.w};
  uint32_t out[8];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    float f0, f1, f2, f3;
    if constexpr (FMT == W_I8) {
      f0 = (float)(int8_t)(in[d] & 0xFFu);
      f1 = (float)(int8_t)((in[d] >> 8) & 0xFFu);
      f2 = (float)(int8_t)((in[d] >> 16) & 0xFFu);
      f3 = (float)(int8_t)(in[d] >> 24);
    } else if constexpr (FMT == W_F8E4M3) {
      const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)in[d], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)in[d], true);
      f0 = lo.x; f1 = lo.y; f2 = hi.x; f3 = hi.y;
    } else {
      const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_bf8((int)in[d], false), hi = __builtin_amdgcn_cvt_pk_f32_bf8((int)in[d], true);
      f0 = lo.x; f1 = lo.y; f2 = hi.x; f3 = hi.y;
    }
    out[2 * d] = Mma<DT>::pack(f0, f1);
    out[2 * d + 1] = Mma<DT>::pack(f2, f3);
  }
  c0 = make_uint4(out[0], out[1], out[2], out[3]);
  c1 = make_uint4(out[4], out[5], out[6], out[7]);
}

// 16 packed bytes -> 16 low-nibble and 16 high-nibble weights as (OFFSET + q) in the activation dtype
template <int DT>
__device__ __forceinline__ void convert16_i4(const uint4& w, uint4 (&lo)[2], uint4 (&hi)[2]) {
  const uint32_t in[4] = {w.x, w.y, w.z, w.w};
  uint32_t l[8], h[8];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t b01 = __builtin_amdgcn_perm(0u, in[d], 0x0C010C00u);  // byte0 | byte1 << 16
    const uint32_t b23 = __builtin_amdgcn_perm(0u, in[d], 0x0C030C02u);  // byte2 | byte3 << 16
    l[2 * d] = (b01 & 0x000F000Fu) | Mma<DT>::MAGIC;
    h[2 * d] = ((b01 >> 4) & 0x000F000Fu) | Mma<DT>::MAGIC;
    l[2 * d + 1] = (b23 & 0x000F000Fu) | Mma<DT>::MAGIC;
    h[2 * d + 1] = ((b23 >> 4) & 0x000F000Fu) | Mma<DT>::MAGIC;
  }
  lo[0] = make_uint4(l[0], l[1], l[2], l[3]);
  lo[1] = make_uint4(l[4], l[5], l[6], l[7]);
  hi[0] = make_uint4(h[0], h[1], h[2], h[3]);
  hi[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

struct MmaArgs {
  const void* x;        // [M, K] activation dtype
  const uint8_t* w;     // W8: [N, K] bytes; W4: packed [N/2, K] bytes
  const void* scale;    // W8: [N]; W4: [N*G]
  const void* shift;    // W4 only: [N*G] (activation dtype, or uint8/int8 zero-point)
  const float* xs;      // W4 only: workspace [G][Mpad] group sums of x
  const void* bias;     // [N] or null
  void* y;              // [M, N]
  int M, N, K, C, G, Mpad;
};

template <int DT, int FMT, bool INT_SHIFT>
__global__ void __launch_bounds__(256, 2) qmm_mfma_kernel(const MmaArgs a) {
  using E = Elem<DT>;
  using T = typename E::T;
  using V8 = typename Mma<DT>::V8;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // [2 buffers][A tile | B tile]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, nt = blockIdx.x;
  const int M = a.M, N = a.N, K = a.K;
  const int nk = K / BK;
  const int P = N >> 1;  // packed rows (W4)

  const T* xg = reinterpret_cast<const T*>(a.x);

  // ---- staging assignment ----------------------------------------------------------------------
  // A: 4 chunks/thread: chunk c = tid + 256*j -> row c>>3, kc c&7
  // W8: 2 chunks/thread of 16 bytes: c = tid + 256*j -> row c>>2, part c&3 (16 weights -> kc 2*part, 2*part+1)
  // W4: 1 chunk/thread: packed row tid>>2, part tid&3 -> tile rows (tid>>2) [low plane] and 64+(tid>>2) [high plane]
  uint4 ra[4], rw[2];
  auto issue_loads = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + 256 * j, row = c >> 3, kc = c & 7;
      int m = m0 + row;
      m = m < M ? m : M - 1;
      ra[j] = *reinterpret_cast<const uint4*>(xg + (size_t)m * K + k0 + kc * 8);
    }
    if constexpr (FMT == W_I4) {
      int p = nt * 64 + (tid >> 2);
      p = p < P ? p : P - 1;
      rw[0] = *reinterpret_cast<const uint4*>(a.w + (size_t)p * K + k0 + (tid & 3) * 16);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = tid + 256 * j, row = c >> 2, part = c & 3;
        int n = nt * BN + row;
        n = n < N ? n : N - 1;
        rw[j] = *reinterpret_cast<const uint4*>(a.w + (size_t)n * K + k0 + part * 16);
      }
    }
  };
  auto write_lds = [&](int buf) {
    uint8_t* sa = smem + buf * 2 * TILE_BYTES;
    uint8_t* sb = sa + TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + 256 * j, row = c >> 3, kc = c & 7;
      *reinterpret_cast<uint4*>(sa + lds_off(row, kc)) = ra[j];
    }
    if constexpr (FMT == W_I4) {
      uint4 lo[2], hi[2];
      convert16_i4<DT>(rw[0], lo, hi);
      const int row = tid >> 2, part = tid & 3;
      *reinterpret_cast<uint4*>(sb + lds_off(row, 2 * part)) = lo[0];
      *reinterpret_cast<uint4*>(sb + lds_off(row, 2 * part + 1)) = lo[1];
      *reinterpret_cast<uint4*>(sb + lds_off(64 + row, 2 * part)) = hi[0];
      *reinterpret_cast<uint4*>(sb + lds_off(64 + row, 2 * part + 1)) = hi[1];
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = tid + 256 * j, row = c >> 2, part = c & 3;
        uint4 c0, c1;
        convert16<DT, FMT>(rw[j], c0, c1);
        *reinterpret_cast<uint4*>(sb + lds_off(row, 2 * part)) = c0;
        *reinterpret_cast<uint4*>(sb + lds_off(row, 2 * part + 1)) = c1;
      }
    }
  };

  f32x4 acc[4][4], accg[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      accg[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  // column owned by this lane in fragment j: tile column wn*64 + j*16 + (lane&15)
  int ncol[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tc = wn * 64 + j * 16 + (lane & 15);
    if constexpr (FMT == W_I4) {
      const int p = nt * 64 + (tc & 63);
      ncol[j] = (p < P) ? p + (tc >> 6) * P : -1;
    } else {
      const int n = nt * BN + tc;
      ncol[j] = n < N ? n : -1;
    }
  }
  const int steps_per_group = (FMT == W_I4) ? a.C / BK : 1;

  issue_loads(0);
  write_lds(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) issue_loads(kt + 1);
    const uint8_t* sa = smem + cur * 2 * TILE_BYTES;
    const uint8_t* sb = sa + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      V8 fa[4], fb[4];
      const int kc = kk * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const V8*>(sa + lds_off(wm * 64 + i * 16 + (lane & 15), kc));
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const V8*>(sb + lds_off(wn * 64 + j * 16 + (lane & 15), kc));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (FMT == W_I4)
            accg[i][j] = Mma<DT>::run(fa[i], fb[j], accg[i][j]);
          else
            acc[i][j] = Mma<DT>::run(fa[i], fb[j], acc[i][j]);
        }
    }
    if constexpr (FMT == W_I4) {
      if ((kt + 1) % steps_per_group == 0) {
        const int g = kt / steps_per_group;
        float s[4], zz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[j] = 0.f;
          zz[j] = 0.f;
          if (ncol[j] >= 0) {
            const size_t idx = (size_t)ncol[j] * a.G + g;
            s[j] = E::to_f32(reinterpret_cast<const T*>(a.scale)[idx]);
            if constexpr (INT_SHIFT)
              zz[j] = s[j] * ((float)(int8_t) reinterpret_cast<const uint8_t*>(a.shift)[idx] + Mma<DT>::OFFSET);
            else
              zz[j] = E::to_f32(reinterpret_cast<const T*>(a.shift)[idx]) + Mma<DT>::OFFSET * s[j];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 xs = *reinterpret_cast<const f32x4*>(a.xs + (size_t)g * a.Mpad + m0 + wm * 64 + i * 16 + (lane >> 4) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              acc[i][j][r] += s[j] * accg[i][j][r] - zz[j] * xs[r];
              accg[i][j][r] = 0.f;
            }
          }
        }
      }
    }
    if (kt + 1 < nk) write_lds(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue ----------------------------------------------------------------------------------
  T* yg = reinterpret_cast<T*>(a.y);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = ncol[j];
    if (n < 0) continue;
    float sc = 1.f;
    if constexpr (FMT != W_I4) sc = E::to_f32(reinterpret_cast<const T*>(a.scale)[n]);
    const bool has_bias = a.bias != nullptr;
    const float bv = has_bias ? E::to_f32(reinterpret_cast<const T*>(a.bias)[n]) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (m < M) {
          float v = acc[i][j][r] * sc;
          asm volatile("" : "+v"(v));  // product rounded to fp32 first, with and without bias (no single-rounding v_fma_mixlo_f16)
          if (has_bias) v = E::to_f32(E::from_f32(v)) + bv;
          yg[(size_t)m * N + n] = E::from_f32(v);
        }
      }
    }
  }
}

// ---- XS[g][m] = sum_{k in group g} x[m,k]  (one wave per row, 16 lanes per 128-wide... C-wide group) ----
template <int DT>
__global__ void __launch_bounds__(256) group_sums_kernel(const typename Elem<DT>::T* __restrict__ x, float* __restrict__ xs, int M,
                                                        int K, int C, int Mpad) {
  using E = Elem<DT>;
  using T = typename E::T;
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int lanes_per_group = C / 8;  // 8 elements (16 bytes) per lane; C in {64, 128} -> 8 or 16 lanes
  for (int k0 = lane * 8; k0 < K; k0 += 512) {
    const uint4 v = *reinterpret_cast<const uint4*>(x + (size_t)m * K + k0);
    const uint32_t pr[4] = {v.x, v.y, v.z, v.w};
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s += E::to_f32(__builtin_bit_cast(T, (uint16_t)(pr[q] & 0xFFFFu)));
      s += E::to_f32(__builtin_bit_cast(T, (uint16_t)(pr[q] >> 16)));
    }
    for (int off = 1; off < lanes_per_group; off <<= 1) s += __shfl_xor(s, off, 64);
    if ((lane % lanes_per_group) == 0) xs[(size_t)(k0 / C) * Mpad + m] = s;
  }
}

int qbits_group_sums(const void* x, float* xs, int M, int K, int C, int Mpad, int dtype, hipStream_t stream) {
  const dim3 grid((unsigned)((M + 3) / 4));
  if (dtype == QUANTO_HIP_BF16)
    hipLaunchKernelGGL(group_sums_kernel<QUANTO_HIP_BF16>, grid, dim3(256), 0, stream, reinterpret_cast<const __bf16*>(x), xs, M, K, C, Mpad);
  else if (dtype == QUANTO_HIP_F16)
    hipLaunchKernelGGL(group_sums_kernel<QUANTO_HIP_F16>, grid, dim3(256), 0, stream, reinterpret_cast<const _Float16*>(x), xs, M, K, C, Mpad);
  else
    return QUANTO_HIP_ENOTSUP;
  return launch_status();
}

// ------------------------------------------------------------------------------------------------
template <int DT, int FMT, bool INT_SHIFT>
static int mma_launch(const MmaArgs& a, hipStream_t stream) {
  static bool attr_done = false;
  constexpr int lds = 2 * 2 * TILE_BYTES;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qmm_mfma_kernel<DT, FMT, INT_SHIFT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const int ntiles = FMT == W_I4 ? (a.N / 2 + 63) / 64 : (a.N + BN - 1) / BN;
  dim3 grid(ntiles, (a.M + BM - 1) / BM);
  hipLaunchKernelGGL((qmm_mfma_kernel<DT, FMT, INT_SHIFT>), grid, dim3(256), lds, stream, a);
  return launch_status();
}

bool qbytes_mfma_supported(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  const bool bd = b_dtype == QUANTO_HIP_I8 || b_dtype == QUANTO_HIP_F8_E4M3FN || b_dtype == QUANTO_HIP_F8_E5M2;
  return bd && a_dtype == out_dtype && (out_dtype == QUANTO_HIP_BF16 || out_dtype == QUANTO_HIP_F16) && M >= 1 && K % BK == 0 &&
         K >= BK && M < (1 << 30) && N < (1 << 30) && K < (1 << 30);
}

int qbytes_mm_mfma(const void* x, const void* w, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int a_dtype,
                   int b_dtype, int out_dtype, hipStream_t stream) {
  if (!qbytes_mfma_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16) return QUANTO_HIP_EALIGN;
  MmaArgs a{x, reinterpret_cast<const uint8_t*>(w), s, nullptr, nullptr, bias, y, (int)M, (int)N, (int)K, 0, 0, 0};
#define QH_CASE(DT, FMT) return mma_launch<DT, FMT, false>(a, stream)
  if (out_dtype == QUANTO_HIP_BF16) {
    if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_BF16, W_I8);
    if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_BF16, W_F8E4M3);
    QH_CASE(QUANTO_HIP_BF16, W_F8E5M2);
  }
  if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_F16, W_I8);
  if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_F16, W_F8E4M3);
  QH_CASE(QUANTO_HIP_F16, W_F8E5M2);
#undef QH_CASE
}

bool qbits_mfma_supported(int64_t M, const PackedGeom& g, int dtype) {
  return g.bits == 4 && (g.C == 64 || g.C == 128) && (g.N % 2 == 0) && (g.K % g.C == 0) && M >= 1 &&
         (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) && M < (1 << 30) && g.N < (1 << 30) && g.K < (1 << 30);
}

size_t qbits_mfma_workspace(int64_t M, const PackedGeom& g) {
  const int64_t Mpad = (M + BM - 1) / BM * BM;
  return (size_t)g.G * Mpad * sizeof(float);
}

int qbits_mm_mfma(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t M,
                  const PackedGeom& g, int dtype, bool int_shift, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!qbits_mfma_supported(M, g, dtype)) return QUANTO_HIP_ENOTSUP;
  if (workspace == nullptr || workspace_bytes < qbits_mfma_workspace(M, g)) return QUANTO_HIP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(workspace)) % 16)
    return QUANTO_HIP_EALIGN;
  const int Mpad = (int)((M + BM - 1) / BM * BM);
  MmaArgs a{x, packed, scale, shift, reinterpret_cast<const float*>(workspace), bias, y, (int)M, (int)g.N, (int)g.K, (int)g.C, (int)g.G, Mpad};
  const dim3 sgrid((unsigned)((M + 3) / 4));
  if (dtype == QUANTO_HIP_BF16) {
    hipLaunchKernelGGL(group_sums_kernel<QUANTO_HIP_BF16>, sgrid, dim3(256), 0, stream, reinterpret_cast<const __bf16*>(x),
                       reinterpret_cast<float*>(workspace), a.M, a.K, a.C, Mpad);
    return int_shift ? mma_launch<QUANTO_HIP_BF16, W_I4, true>(a, stream) : mma_launch<QUANTO_HIP_BF16, W_I4, false>(a, stream);
  }
  hipLaunchKernelGGL(group_sums_kernel<QUANTO_HIP_F16>, sgrid, dim3(256), 0, stream, reinterpret_cast<const _Float16*>(x),
                     reinterpret_cast<float*>(workspace), a.M, a.K, a.C, Mpad);
  return int_shift ? mma_launch<QUANTO_HIP_F16, W_I4, true>(a, stream) : mma_launch<QUANTO_HIP_F16, W_I4, false>(a, stream);
}

}  // namespace qh
