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
//
// CONV (r4): the same kernel as an IMPLICIT GEMM for a dense convolution with an 8-bit weight [OC, C, kh, kw] (nn/qconv2d.py:54-55): the
// activation operand A[m][k], m = (image, oh, ow), k = (c, i, j) - the order the weight is flattened in - is gathered from the NCHW input
// inside the staging loads (zero where the window hangs over the padding), no im2col tensor is ever written; the epilogue stores NCHW.
#include "qh_common.h"

namespace qh {

// W_I4: generic packed int4, operand OFFSET + q, per-group fold in fp32 (exact math).  W_I4R (r4, convolution): the same bytes dequantized at
// staging time with the reference's roundings (tensor/qbits.py:27-49: T(T(s q) - z) for float shifts, T(s (q - zp)) for zero-points) - the LDS
// operand IS the reference's dense weight, no fold, no workspace
enum WFmt { W_I8 = 0, W_F8E4M3 = 1, W_F8E5M2 = 2, W_I4 = 3, W_I4R = 4 };

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

// 16 packed bytes -> 16 low-nibble and 16 high-nibble weights dequantized as the reference does (two roundings for float shifts, one for
// zero-points), round-to-nearest-even into the activation dtype
template <int DT, bool INT_SHIFT>
__device__ __forceinline__ void convert16_i4r(const uint4& w, float s_lo, float z_lo, float s_hi, float z_hi, uint4 (&lo)[2], uint4 (&hi)[2]) {
  using E = Elem<DT>;
  auto deq = [](uint32_t q, float sc, float z) -> uint32_t {
    float v;
    if constexpr (INT_SHIFT)
      v = sc * ((float)q - z);
    else
      v = E::to_f32(E::from_f32(sc * (float)q)) - z;
    return (uint32_t)__builtin_bit_cast(uint16_t, E::from_f32(v));
  };
  const uint32_t in[4] = {w.x, w.y, w.z, w.w};
  uint32_t l[8], h[8];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const uint32_t two = in[d] >> (16 * b);  // bytes 2b, 2b + 1
      l[2 * d + b] = deq(two & 0xFu, s_lo, z_lo) | (deq((two >> 8) & 0xFu, s_lo, z_lo) << 16);
      h[2 * d + b] = deq((two >> 4) & 0xFu, s_hi, z_hi) | (deq((two >> 12) & 0xFu, s_hi, z_hi) << 16);
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
  void* y;              // [M, N]; CONV: [B, N, OH, OW]
  int M, N, K, C, G, Mpad;
  // CONV only: x is [B, cin, H, W]; M = B * OH * OW, K = cin * KH * KW
  int cin, H, W, KH, KW, OH, OW, sh, sw, ph, pw, dh, dw;
};

template <int DT, int FMT, bool INT_SHIFT, bool CONV = false>
__global__ void __launch_bounds__(256, 2) qmm_mfma_kernel(const MmaArgs a) {
  static_assert(!CONV || FMT != W_I4, "implicit-GEMM convolution: 8-bit weights, or int4 dequantized at staging (W_I4R)");
  constexpr bool PACKED4 = FMT == W_I4 || FMT == W_I4R;  // generic packed int4 rows: a tile's 128 columns are 64 packed rows x both nibble planes
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
  float rs[2] = {0.f, 0.f}, rz[2] = {0.f, 0.f};  // W_I4R: scale / shift of the thread's packed row, low and high plane, group of the K-tile chunk
  // CONV (r4, table-driven gather): a thread stages ONE output pixel - tile row tid & 127, chunks kc = (tid >> 7) + 2 j - so the 64 lanes of a
  // load are 64 neighbouring pixels and k is uniform across a wave.  What depends on k only - the byte offset of tap (ci, ki, kj) relative to the
  // window's top-left tap, and the tap's number ki KW + kj - is computed ONCE per K-tile by 64 threads into a 64-entry LDS table (two buffers);
  // what depends on the pixel only - its base offset and one validity bit per tap - lives in three registers.  An element then costs
  // add + bit-extract + two selects instead of a division-free but ~30-instruction walk per lane (a lone wave per SIMD issues a VALU op every
  // ~8 cycles: the first form spent 3.7 us per K-tile whatever M was, profiles/r04_qconv2d_paths_grid_before_table_gather.jsonl).
  uint32_t cv_voff = 0;  // byte offset of input element (b, 0, oh sh, ow sw): the window's top-left tap shifted right/down by the padding
  uint64_t cv_mask = 0;  // bit ki KW + kj: tap (ki, kj) of this pixel's window lies inside the image (KH KW <= 64)
  uint32_t cv_raw[CONV ? 4 : 1][8], cv_keep = 0;  // the gathered elements of the K-tile in flight (invalid taps hold x[0]) and their validity bits
  int2* ktab = reinterpret_cast<int2*>(smem + 2 * 2 * TILE_BYTES);  // [2][64] {byte offset (signed), tap number}
  auto fill_ktab = [&](int kt) {
    if (tid < 64) {
      const int khw = a.KH * a.KW, k = kt * BK + tid;
      const int ci = k / khw, rem = k - ci * khw, ki = rem / a.KW, kj = rem - ki * a.KW;
      ktab[(kt & 1) * 64 + tid] = make_int2(2 * ((ci * a.H + ki * a.dh) * a.W + kj * a.dw - (a.ph * a.W + a.pw)), rem);
    }
  };
  if constexpr (CONV) {
    const int L = a.OH * a.OW;
    int m = m0 + (tid & 127);
    m = m < M ? m : M - 1;
    const int b = m / L, l = m - b * L, oh = l / a.OW, ow = l - oh * a.OW;
    const int ih0 = oh * a.sh - a.ph, iw0 = ow * a.sw - a.pw;
    cv_voff = 2u * (uint32_t)(b * a.cin * a.H * a.W + oh * a.sh * a.W + ow * a.sw);
    for (int ki = 0; ki < a.KH; ++ki)
      for (int kj = 0; kj < a.KW; ++kj) {
        const int ih = ih0 + ki * a.dh, iw = iw0 + kj * a.dw;
        if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) cv_mask |= 1ull << (ki * a.KW + kj);
      }
  }
  auto issue_loads = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + 256 * j, row = c >> 3, kc = c & 7;
      if constexpr (CONV) {
        // eight consecutive k of the thread's pixel: table entries by broadcast LDS reads, invalid taps read element 0 and are zeroed
        const int kcw = __builtin_amdgcn_readfirstlane(tid >> 7) + 2 * j;
        const int4* tp = reinterpret_cast<const int4*>(ktab + (kt & 1) * 64 + kcw * 8);
        const int4 t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3];
        const int off[8] = {t0.x, t0.z, t1.x, t1.z, t2.x, t2.z, t3.x, t3.z}, tap[8] = {t0.y, t0.w, t1.y, t1.w, t2.y, t2.w, t3.y, t3.w};
        // the 32 loads of a K-tile are issued back to back and stay in flight during the MFMAs; masking and packing happen in write_lds
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const uint32_t ok = (uint32_t)((cv_mask >> tap[q]) & 1ull);
          cv_keep = j == 0 && q == 0 ? ok : cv_keep | (ok << (8 * j + q));
          const uint32_t voff = (cv_voff + (uint32_t)off[q]) & (0u - ok);
          cv_raw[j][q] = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(xg) + voff);
        }
      } else {
        int m = m0 + row;
        m = m < M ? m : M - 1;
        ra[j] = *reinterpret_cast<const uint4*>(xg + (size_t)m * K + k0 + kc * 8);
      }
    }
    if constexpr (PACKED4) {
      int p = nt * 64 + (tid >> 2);
      p = p < P ? p : P - 1;
      rw[0] = *reinterpret_cast<const uint4*>(a.w + (size_t)p * K + k0 + (tid & 3) * 16);
      if constexpr (FMT == W_I4R) {
        const int g = (k0 + (tid & 3) * 16) / a.C;  // the 16 k of a chunk lie in one group (C % 16 == 0)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const size_t idx = (size_t)(pl * P + p) * a.G + g;
          rs[pl] = E::to_f32(reinterpret_cast<const T*>(a.scale)[idx]);
          if constexpr (INT_SHIFT)
            rz[pl] = (float)(int8_t) reinterpret_cast<const uint8_t*>(a.shift)[idx];
          else
            rz[pl] = E::to_f32(reinterpret_cast<const T*>(a.shift)[idx]);
        }
      }
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
      const int c = tid + 256 * j;
      const int row = CONV ? (tid & 127) : c >> 3, kc = CONV ? (tid >> 7) + 2 * j : c & 7;
      if constexpr (CONV) {
        uint32_t e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) e[q] = cv_raw[j][q] & (uint32_t)__builtin_amdgcn_sbfe(cv_keep, 8 * j + q, 1);  // zero where the tap hangs over the padding
        ra[j] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
      }
      *reinterpret_cast<uint4*>(sa + lds_off(row, kc)) = ra[j];
    }
    if constexpr (PACKED4) {
      uint4 lo[2], hi[2];
      if constexpr (FMT == W_I4R)
        convert16_i4r<DT, INT_SHIFT>(rw[0], rs[0], rz[0], rs[1], rz[1], lo, hi);
      else
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
    if constexpr (PACKED4) {
      const int p = nt * 64 + (tc & 63);
      ncol[j] = (p < P) ? p + (tc >> 6) * P : -1;
    } else {
      const int n = nt * BN + tc;
      ncol[j] = n < N ? n : -1;
    }
  }
  const int steps_per_group = (FMT == W_I4) ? a.C / BK : 1;

  if constexpr (CONV) {
    fill_ktab(0);
    if (nk > 1) fill_ktab(1);
    __syncthreads();
  }
  issue_loads(0);
  write_lds(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) issue_loads(kt + 1);
    if constexpr (CONV) {  // table of tile kt + 2 into the buffer tile kt's gather (an iteration ago) was the last to read; visible after this iteration's barrier
      if (kt + 2 < nk) fill_ktab(kt + 2);
    }
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
    if constexpr (!PACKED4) sc = E::to_f32(reinterpret_cast<const T*>(a.scale)[n]);
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
          if constexpr (CONV) {
            const int L = a.OH * a.OW, b = m / L;
            yg[((size_t)b * N + n) * L + (m - b * L)] = E::from_f32(v);  // NCHW: the lane's four rows are four neighbouring pixels
          } else {
            yg[(size_t)m * N + n] = E::from_f32(v);
          }
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
template <int DT, int FMT, bool INT_SHIFT, bool CONV = false>
static int mma_launch(const MmaArgs& a, hipStream_t stream) {
  static bool attr_done = false;
  constexpr int lds = 2 * 2 * TILE_BYTES + (CONV ? 2 * 64 * 8 : 0);  // + the convolution's two k-tables
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qmm_mfma_kernel<DT, FMT, INT_SHIFT, CONV>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const int ntiles = (FMT == W_I4 || FMT == W_I4R) ? (a.N / 2 + 63) / 64 : (a.N + BN - 1) / BN;
  dim3 grid(ntiles, (a.M + BM - 1) / BM);
  hipLaunchKernelGGL((qmm_mfma_kernel<DT, FMT, INT_SHIFT, CONV>), grid, dim3(256), lds, stream, a);
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

// Dense convolution with an 8-bit weight as an implicit GEMM (CONV above).  K = cin * KH * KW must be a multiple of 64 (the K-tile); every
// element offset must fit 31 bits (input: byte offsets), windows of up to 64 taps (one validity bit per tap and pixel).
bool qbytes_conv2d_supported(int64_t B, int64_t cin, int64_t H, int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int a_dtype,
                             int b_dtype, int out_dtype) {
  const bool bd = b_dtype == QUANTO_HIP_I8 || b_dtype == QUANTO_HIP_F8_E4M3FN || b_dtype == QUANTO_HIP_F8_E5M2;
  const int64_t K = cin * KH * KW;
  return bd && a_dtype == out_dtype && (out_dtype == QUANTO_HIP_BF16 || out_dtype == QUANTO_HIP_F16) && B >= 1 && OH >= 1 && OW >= 1 && K % BK == 0 &&
         KH * KW <= 64 && B * cin * H * W < (1ll << 30) && B * OC * OH * OW < (1ll << 31) && OC * K < (1ll << 31) && B * OH * OW < (1ll << 30);
}

int qbytes_conv2d_mfma(const void* x, const void* w, const void* s, const void* bias, void* y, int64_t B, int64_t cin, int64_t H, int64_t W, int64_t OC,
                       int64_t KH, int64_t KW, int64_t OH, int64_t OW, int sh, int sw, int ph, int pw, int dh, int dw, int a_dtype, int b_dtype,
                       int out_dtype, hipStream_t stream) {
  if (!qbytes_conv2d_supported(B, cin, H, W, OC, KH, KW, OH, OW, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  if (reinterpret_cast<uintptr_t>(w) % 16) return QUANTO_HIP_EALIGN;
  MmaArgs a{x, reinterpret_cast<const uint8_t*>(w), s, nullptr, nullptr, bias, y, (int)(B * OH * OW), (int)OC, (int)(cin * KH * KW), 0, 0, 0,
            (int)cin, (int)H, (int)W, (int)KH, (int)KW, (int)OH, (int)OW, sh, sw, ph, pw, dh, dw};
#define QH_CASE(DT, FMT) return mma_launch<DT, FMT, false, true>(a, stream)
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

// Dense convolution with a generic packed int4 weight (r4): the CONV gather above + W_I4R staging.  The weight [OC, cin, KH, KW] quantized along
// axis 0 is the [OC, K = cin * KH * KW] operand of qbits_mm (groups run along the flattened K); group sizes that are multiples of 16 (a
// staging chunk must not straddle groups) and per-channel scales.
bool qbits_conv2d_supported(int64_t B, int64_t cin, int64_t H, int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, const PackedGeom& g,
                            int dtype) {
  const int64_t K = cin * KH * KW;
  return g.bits == 4 && g.N == OC && g.K == K && OC % 2 == 0 && g.C % 16 == 0 && (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) && B >= 1 &&
         OH >= 1 && OW >= 1 && K % BK == 0 && KH * KW <= 64 && B * cin * H * W < (1ll << 30) && B * OC * OH * OW < (1ll << 31) && OC * K < (1ll << 31) &&
         OC * g.G < (1ll << 31) && B * OH * OW < (1ll << 30);
}

int qbits_conv2d_mfma(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t B, int64_t cin,
                      int64_t H, int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int sh, int sw, int ph, int pw, int dh, int dw,
                      const PackedGeom& g, int dtype, bool int_shift, hipStream_t stream) {
  if (!qbits_conv2d_supported(B, cin, H, W, OC, KH, KW, OH, OW, g, dtype)) return QUANTO_HIP_ENOTSUP;
  if (reinterpret_cast<uintptr_t>(packed) % 16) return QUANTO_HIP_EALIGN;
  MmaArgs a{x, packed, scale, shift, nullptr, bias, y, (int)(B * OH * OW), (int)OC, (int)(cin * KH * KW), (int)g.C, (int)g.G, 0,
            (int)cin, (int)H, (int)W, (int)KH, (int)KW, (int)OH, (int)OW, sh, sw, ph, pw, dh, dw};
  if (dtype == QUANTO_HIP_BF16)
    return int_shift ? mma_launch<QUANTO_HIP_BF16, W_I4R, true, true>(a, stream) : mma_launch<QUANTO_HIP_BF16, W_I4R, false, true>(a, stream);
  return int_shift ? mma_launch<QUANTO_HIP_F16, W_I4R, true, true>(a, stream) : mma_launch<QUANTO_HIP_F16, W_I4R, false, true>(a, stream);
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
