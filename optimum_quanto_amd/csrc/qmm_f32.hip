// fp32 activations on the fast path (r6).  A model quantized without a dtype cast - PyTorch's default, and what the reference's own
// tests/nn/test_qlinear.py:116-135 runs - computes in float32: scales, shifts, activations and outputs are fp32.  Until r6 such calls took the
// one-thread-per-output kernels (naive_mm.hip).  Two kernels, every weight format of the ABI (int4 / int2 packed planes with any group size that is a
// multiple of 16 or per-channel scales, float shift or integer zero-point; int8 / fp8-e4m3fn / e5m2 / e4m3fnuz with per-channel scales):
//
//   * gemv_f32 (M <= 8, in passes of up to 4 rows): the weight stream of qbits_gemv.hip / qbytes_gemv.hip - a wave owns 1 KiB slabs of 4 (packed)
//     rows, all of its loads requested up front, 16-byte coalesced weight loads - with fp32 arithmetic: the stored integers (v_cvt_f32_ubyteN) or
//     fp8 values times x on v_fma_f32, per-lane group fold  y += s * sum(q x) - z * sum(x)  (exact products of the stored values, fp32 accumulation:
//     the numerics contract of DESIGN.md section 3), DPP wave reduction, one LDS exchange between the four waves of a block.  HBM-bound.
//   * mm_f32 (M > 8): 128 x 128 x 32 tiles on gfx950's fp32-input matrix instruction v_mfma_f32_16x16x4_f32 - exact fp32 products and a k-ordered
//     fp32 fma chain, at the fp32 vector rate (157 TF chip peak, 1/16 of bf16).  The weight tile is dequantized to fp32 WHILE IT IS STAGED into LDS with
//     the reference's own expression and roundings for fp32 tensors (tensor/qbits.py:27-49: fp32(fp32(scale * q) - shift), scale * (q - zero_point)),
//     so the matrix cores multiply by exactly the dense fp32 weight the reference materialises; 8-bit weights are staged as their exact fp32 values and
//     the per-channel scale multiplies the fp32 accumulator (library/qbytes_mm.py:25-33 up to the order of one rounding).
//
// Reference call sites: tensor/function.py:41-47 (QuantizedLinearFunction), tensor/weights/qbytes.py:68-82, library/qbytes_mm.py:22-33.
#include "qh_common.h"

namespace qh {
namespace f32k {

enum { W_I4 = 0, W_I2 = 1, W_I8 = 2, W_F8E4M3 = 3, W_F8E5M2 = 4, W_F8FNUZ = 5 };

template <int FMT>
struct Fmt {
  static constexpr int PL = FMT == W_I4 ? 2 : (FMT == W_I2 ? 4 : 1);  // values per stored byte
  static constexpr bool QBITS = FMT == W_I4 || FMT == W_I2;
};

struct Args {
  const float* x;        // [M, K]
  const uint8_t* w;      // qbits: packed [N / PL, K]; 8-bit: [N, K]
  const void* scale;     // fp32: qbits [N * G], 8-bit [N]
  const void* shift;     // qbits: fp32 [N * G], or uint8 / int8 zero-points; 8-bit: unused
  const float* bias;     // [N] or null
  float* y;              // [M, N]
  int M, N, K;
  int C, G;              // group size along K and groups per row (per-channel: C = K, G = 1)
};

// byte i (0..3) of a dword as the fp32 value of the stored element
template <int FMT>
__device__ __forceinline__ float byte_value(uint32_t word, int i) {
  const uint32_t b = (word >> (8 * i)) & 0xFFu;
  if constexpr (FMT == W_I8) return (float)(int8_t)b;
  else if constexpr (FMT == W_F8E4M3) {  // the byte select of the converter is an immediate
    return i == 0 ? __builtin_amdgcn_cvt_f32_fp8((int)word, 0) : i == 1 ? __builtin_amdgcn_cvt_f32_fp8((int)word, 1)
         : i == 2 ? __builtin_amdgcn_cvt_f32_fp8((int)word, 2) : __builtin_amdgcn_cvt_f32_fp8((int)word, 3);
  } else if constexpr (FMT == W_F8E5M2) {
    return i == 0 ? __builtin_amdgcn_cvt_f32_bf8((int)word, 0) : i == 1 ? __builtin_amdgcn_cvt_f32_bf8((int)word, 1)
         : i == 2 ? __builtin_amdgcn_cvt_f32_bf8((int)word, 2) : __builtin_amdgcn_cvt_f32_bf8((int)word, 3);
  }
  else if constexpr (FMT == W_F8FNUZ) return decode_e4m3fnuz(b);
  else return (float)b;  // already masked plane bytes (v_cvt_f32_ubyteN)
}
// the four elements of plane h held by a dword of packed bytes (qbits), or the dword's four 8-bit elements
template <int FMT>
__device__ __forceinline__ void unpack4(uint32_t word, int h, float (&out)[4]) {
  uint32_t v = word;
  if constexpr (FMT == W_I4) v = (word >> (4 * h)) & 0x0F0F0F0Fu;
  if constexpr (FMT == W_I2) v = (word >> (2 * h)) & 0x03030303u;
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = byte_value<FMT>(v, i);
}

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// sum over the 64 lanes; lane 63 holds the total (same ladder as qbits_gemv.hip)
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  v = dpp_add<0x142, 0xA>(v);
  v = dpp_add<0x143, 0xC>(v);
  return v;
}

// ================================================================================================================================
// decode: M <= 4 rows per launch
// ================================================================================================================================
constexpr int RR = 4;  // (packed) rows per block

template <int FMT, int MT, bool INT_SHIFT>
__global__ void __launch_bounds__(256) gemv_f32_kernel(const Args a, int m0) {
  constexpr int PL = Fmt<FMT>::PL;
  constexpr bool QBITS = Fmt<FMT>::QBITS;
  constexpr int NV = RR * PL * MT;
  __shared__ float red[4][NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = a.K, N = a.N, P = N / PL;
  const int p0 = blockIdx.x * RR;
  const int nslab = (K + 1023) >> 10;
  const float* __restrict__ scale = reinterpret_cast<const float*>(a.scale);

  float vals[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) vals[v] = 0.f;

  for (int slab = wave; slab < nslab; slab += 4) {
    const int k0 = (slab * 64 + lane) * 16;
    const bool valid = k0 < K;  // K % 16 == 0: a lane's 16 columns are all inside or all outside
    const int kk = valid ? k0 : 0;
    uint4 W[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int pr = p0 + r < P ? p0 + r : P - 1;  // rows beyond the matrix: duplicate data, never stored
      typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.w + (size_t)pr * K + kk));  // every weight byte is read once
      W[r] = make_uint4(v.x, v.y, v.z, v.w);
    }
    float xv[MT][16], xs[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4* px = reinterpret_cast<const float4*>(a.x + (size_t)(m0 + m) * K + kk);
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = px[q];
        if (!valid) v = make_float4(0.f, 0.f, 0.f, 0.f);  // x = 0 beyond K: such a slab contributes exactly 0
        xv[m][4 * q + 0] = v.x;
        xv[m][4 * q + 1] = v.y;
        xv[m][4 * q + 2] = v.z;
        xv[m][4 * q + 3] = v.w;
        s += (v.x + v.y) + (v.z + v.w);
      }
      xs[m] = s;
    }
    const int g = a.G == 1 ? 0 : kk / a.C;  // the lane's 16 columns lie inside one group (C % 16 == 0)
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const int pr = p0 + r < P ? p0 + r : P - 1;
      const uint32_t w4[4] = {W[r].x, W[r].y, W[r].z, W[r].w};
#pragma unroll
      for (int h = 0; h < PL; ++h) {
        float dot[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) dot[m] = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          float q[4];
          unpack4<FMT>(w4[d], h, q);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int m = 0; m < MT; ++m) dot[m] = __builtin_fmaf(q[i], xv[m][4 * d + i], dot[m]);
        }
        if constexpr (QBITS) {
          const size_t idx = (size_t)(pr + h * P) * a.G + g;
          const float s = scale[idx];
          float z;
          if constexpr (INT_SHIFT)
            z = s * (float)(int8_t) reinterpret_cast<const uint8_t*>(a.shift)[idx];  // scale * (q - zp) = scale * q - scale * zp
          else
            z = reinterpret_cast<const float*>(a.shift)[idx];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            float& v = vals[(r * PL + h) * MT + m];
            v = __builtin_fmaf(s, dot[m], v);
            v = __builtin_fmaf(-z, xs[m], v);
          }
        } else {
#pragma unroll
          for (int m = 0; m < MT; ++m) vals[(r * PL + h) * MT + m] += dot[m];
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const float t = wave_sum_lane63(vals[v]);
    if (lane == 63) red[wave][v] = t;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int v = threadIdx.x, r = v / (PL * MT), h = (v / MT) % PL, m = v % MT;
    const int p = p0 + r;
    if (p < P) {
      const int n = p + h * P;
      float t = (red[0][v] + red[1][v]) + (red[2][v] + red[3][v]);
      if constexpr (!QBITS) t *= scale[n];
      if (a.bias) t += a.bias[n];
      a.y[(size_t)(m0 + m) * N + n] = t;
    }
  }
}

template <int FMT, bool INT_SHIFT>
static int launch_gemv(const Args& a, hipStream_t stream) {
  constexpr int PL = Fmt<FMT>::PL;
  const int grid = (a.N / PL + RR - 1) / RR;
  int m0 = 0;
  while (m0 < a.M) {
    const int left = a.M - m0;
    // int2 keeps 16 values per row of x: two rows per pass keep the kernel inside 128 registers
    const int mt = (left >= 4 && PL < 4) ? 4 : (left >= 2 ? 2 : 1);
    if (mt == 4) {
      if constexpr (PL < 4) hipLaunchKernelGGL((gemv_f32_kernel<FMT, 4, INT_SHIFT>), dim3(grid), dim3(256), 0, stream, a, m0);
    } else if (mt == 2) {
      hipLaunchKernelGGL((gemv_f32_kernel<FMT, 2, INT_SHIFT>), dim3(grid), dim3(256), 0, stream, a, m0);
    } else {
      hipLaunchKernelGGL((gemv_f32_kernel<FMT, 1, INT_SHIFT>), dim3(grid), dim3(256), 0, stream, a, m0);
    }
    m0 += mt;
  }
  return launch_status();
}

// ================================================================================================================================
// M > 8: 128 x 128 x 32 tiles on v_mfma_f32_16x16x4_f32
// ================================================================================================================================
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDR = BK + 4;  // floats per LDS row: 144-byte pitch keeps the 16-byte fragment reads of 16 rows off each other's banks

template <int FMT, bool INT_SHIFT>
__global__ void __launch_bounds__(256) mm_f32_kernel(const Args a) {
  constexpr int PL = Fmt<FMT>::PL;
  constexpr bool QBITS = Fmt<FMT>::QBITS;
  constexpr int ROWS = BN / PL;  // stored rows per tile: feature slot f = plane * ROWS + row  ->  n = plane * P + p0 + row
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_f32[];  // 72 KiB: beyond the static limit
  typedef float (*tile_t)[BM][LDR];
  tile_t xs = reinterpret_cast<tile_t>(smem_f32);                                          // xs[2][BM][LDR]
  tile_t ws = reinterpret_cast<tile_t>(smem_f32 + 2 * BM * LDR * sizeof(float));           // ws[2][BN][LDR]
  static_assert(BM == BN, "one tile type for both operands");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = a.M, N = a.N, K = a.K, P = N / PL;
  const int m0 = blockIdx.y * BM, p0 = blockIdx.x * ROWS;
  const int nk = K / BK;
  const float* __restrict__ scale = reinterpret_cast<const float*>(a.scale);

  // ---- staging registers: 4 float4 of x and the thread's share of the weight tile (16 / 8 / 4 stored bytes = 16 fp32 values) ----
  float4 gx[4];
  uint4 gw;            // 8-bit: 16 bytes; int4: .xy = 8 bytes; int2: .x = 4 bytes
  float gs[PL], gz[PL];
  // x: thread t -> rows (t >> 3) + 32 i, columns 4 (t & 7)
  const int xr = tid >> 3, xc = (tid & 7) * 4;
  // w: stored row and byte column of the thread
  constexpr int WB = 16 / PL;                     // stored bytes per thread and K-tile
  constexpr int TPR = BK / WB;                    // threads per stored row: 2 / 4 / 8
  const int wr = tid / TPR, wc = (tid % TPR) * WB;
  const int prow = p0 + wr < P ? p0 + wr : P - 1;
  auto load_tile = [&](int kt) {
    const int kb = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = m0 + xr + 32 * i;
      m = m < M ? m : M - 1;
      gx[i] = *reinterpret_cast<const float4*>(a.x + (size_t)m * K + kb + xc);
    }
    const uint8_t* src = a.w + (size_t)prow * K + kb + wc;
    if constexpr (WB == 16) gw = *reinterpret_cast<const uint4*>(src);
    else if constexpr (WB == 8) {
      const uint2 v = *reinterpret_cast<const uint2*>(src);
      gw = make_uint4(v.x, v.y, 0, 0);
    } else {
      gw = make_uint4(*reinterpret_cast<const uint32_t*>(src), 0, 0, 0);
    }
    if constexpr (QBITS) {
      const int g = a.G == 1 ? 0 : (kb + wc) / a.C;  // the thread's WB columns lie inside one group (C % 16 == 0)
#pragma unroll
      for (int h = 0; h < PL; ++h) {
        const size_t idx = (size_t)(prow + h * P) * a.G + g;
        gs[h] = scale[idx];
        if constexpr (INT_SHIFT)
          gz[h] = (float)(int8_t) reinterpret_cast<const uint8_t*>(a.shift)[idx];
        else
          gz[h] = reinterpret_cast<const float*>(a.shift)[idx];
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&xs[buf][xr + 32 * i][xc]) = gx[i];
    const uint32_t w4[4] = {gw.x, gw.y, gw.z, gw.w};
#pragma unroll
    for (int h = 0; h < PL; ++h) {
#pragma unroll
      for (int d = 0; d < WB / 4; ++d) {
        float q[4];
        unpack4<FMT>(w4[d], h, q);
        if constexpr (QBITS) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (INT_SHIFT) {
              q[i] = gs[h] * (q[i] - gz[h]);  // tensor/qbits.py:38-41: scale * (data - zeropoint), the difference is exact
            } else {
              float t = gs[h] * q[i];          // tensor/qbits.py:43-45: scale * data, rounded to fp32 ...
              asm volatile("" : "+v"(t));      // ... and only then minus shift (no fused multiply-add: the reference rounds twice)
              q[i] = t - gz[h];
            }
          }
        }
        *reinterpret_cast<float4*>(&ws[buf][h * ROWS + wr][wc + 4 * d]) = make_float4(q[0], q[1], q[2], q[3]);
      }
    }
  };

  f32x4 acc[4][4];  // acc[j][i]: feature fragment j, token fragment i of the wave's 64 x 64 block
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_tile(0);
  store_tile(0);
  __syncthreads();
  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // lane (fr, fg) holds columns 16 half + 4 fg .. + 3 of its row: element e feeds matrix step e, which therefore sums the columns
      // {16 half + 4 g + e : g = 0..3} - the same four columns on both operands, every column exactly once per tile
      float4 wf[4], xf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const float4*>(&ws[buf][wn * 64 + j * 16 + fr][half * 16 + fg * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const float4*>(&xs[buf][wm * 64 + i * 16 + fr][half * 16 + fg * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float wv = e == 0 ? wf[j].x : (e == 1 ? wf[j].y : (e == 2 ? wf[j].z : wf[j].w));
            const float xv = e == 0 ? xf[i].x : (e == 1 ? xf[i].y : (e == 2 ? xf[i].z : xf[i].w));
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, acc[j][i], 0, 0, 0);
          }
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);  // the other buffer: its readers finished before the barrier that closed tile kt-1
    __syncthreads();
  }

  // ---- epilogue: lane (fr, fg) holds features (slots) 4 fg .. 4 fg + 3 of fragment j for token fr of fragment i ----
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int slot = wn * 64 + j * 16 + fg * 4;
    const int h = slot / ROWS, row = slot % ROWS;  // four consecutive slots stay inside one plane (16 divides ROWS)
    const int p = p0 + row;
    if (p >= P) continue;
    const int n = h * P + p;
    float sc[4], bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nn = p + r < P ? n + r : n;
      sc[r] = QBITS ? 1.f : scale[nn];
      bv[r] = a.bias ? a.bias[nn] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      if (m >= M) continue;
      float out[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[j][i][r];
        if constexpr (!QBITS) v *= sc[r];
        out[r] = v + bv[r];
      }
      float* dst = a.y + (size_t)m * N + n;
      if (p + 3 < P && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        *reinterpret_cast<float4*>(dst) = make_float4(out[0], out[1], out[2], out[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (p + r < P) dst[r] = out[r];
      }
    }
  }
}

template <int FMT, bool INT_SHIFT>
static int launch_mm(const Args& a, hipStream_t stream) {
  constexpr int PL = Fmt<FMT>::PL;
  const dim3 grid((a.N / PL + BN / PL - 1) / (BN / PL), (a.M + BM - 1) / BM);
  constexpr int lds = 2 * (BM + BN) * LDR * (int)sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mm_f32_kernel<FMT, INT_SHIFT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((mm_f32_kernel<FMT, INT_SHIFT>), grid, dim3(256), lds, stream, a);
  return launch_status();
}

}  // namespace f32k

// ---- qbits_mm with fp32 activations ----------------------------------------------------------------------------------------------
static bool f32_qbits_common(int64_t M, const PackedGeom& g, int dtype) {
  const bool groups = g.C == g.K || g.C % 16 == 0;  // a lane's / thread's 16 columns never straddle a group
  return dtype == QUANTO_HIP_F32 && (g.bits == 4 || g.bits == 2) && g.N % g.vpi == 0 && g.K % g.C == 0 && groups && M >= 1 && M < (1 << 30) &&
         g.N < (1 << 30) && g.K < (1 << 30) && g.N * g.G < (1ll << 31);
}
bool qbits_gemv_f32_supported(int64_t M, const PackedGeom& g, int dtype) {
  return f32_qbits_common(M, g, dtype) && g.K % 16 == 0 && M <= QUANTO_HIP_GEMV_F32_MAX_M;
}
bool qbits_mm_f32_supported(int64_t M, const PackedGeom& g, int dtype) {
  return f32_qbits_common(M, g, dtype) && g.K % f32k::BK == 0;
}

int qbits_mm_gemv_f32(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t M,
                      const PackedGeom& g, int dtype, bool int_shift, hipStream_t stream) {
  if (!qbits_gemv_f32_supported(M, g, dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed)) % 16) return QUANTO_HIP_EALIGN;
  const f32k::Args a{reinterpret_cast<const float*>(x), packed, scale, shift, reinterpret_cast<const float*>(bias), reinterpret_cast<float*>(y),
                     (int)M, (int)g.N, (int)g.K, (int)g.C, (int)g.G};
  if (g.bits == 4) return int_shift ? f32k::launch_gemv<f32k::W_I4, true>(a, stream) : f32k::launch_gemv<f32k::W_I4, false>(a, stream);
  return int_shift ? f32k::launch_gemv<f32k::W_I2, true>(a, stream) : f32k::launch_gemv<f32k::W_I2, false>(a, stream);
}

int qbits_mm_f32(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t M, const PackedGeom& g,
                 int dtype, bool int_shift, hipStream_t stream) {
  if (!qbits_mm_f32_supported(M, g, dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed)) % 16) return QUANTO_HIP_EALIGN;
  const f32k::Args a{reinterpret_cast<const float*>(x), packed, scale, shift, reinterpret_cast<const float*>(bias), reinterpret_cast<float*>(y),
                     (int)M, (int)g.N, (int)g.K, (int)g.C, (int)g.G};
  if (g.bits == 4) return int_shift ? f32k::launch_mm<f32k::W_I4, true>(a, stream) : f32k::launch_mm<f32k::W_I4, false>(a, stream);
  return int_shift ? f32k::launch_mm<f32k::W_I2, true>(a, stream) : f32k::launch_mm<f32k::W_I2, false>(a, stream);
}

// ---- qbytes_mm with fp32 activations ---------------------------------------------------------------------------------------------
static bool f32_qbytes_common(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  const bool bd = b_dtype == QUANTO_HIP_I8 || b_dtype == QUANTO_HIP_F8_E4M3FN || b_dtype == QUANTO_HIP_F8_E5M2 || b_dtype == QUANTO_HIP_F8_E4M3FNUZ;
  return bd && a_dtype == QUANTO_HIP_F32 && out_dtype == QUANTO_HIP_F32 && M >= 1 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30);
}
bool qbytes_gemv_f32_supported(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  return f32_qbytes_common(M, N, K, a_dtype, b_dtype, out_dtype) && K % 16 == 0 && M <= QUANTO_HIP_GEMV_F32_MAX_M;
}
bool qbytes_mm_f32_supported(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  return f32_qbytes_common(M, N, K, a_dtype, b_dtype, out_dtype) && K % f32k::BK == 0;
}

template <bool GEMV>
static int qbytes_f32_dispatch(const f32k::Args& a, int b_dtype, hipStream_t stream) {
#define QH_F32_CASE(FMT) return GEMV ? f32k::launch_gemv<FMT, false>(a, stream) : f32k::launch_mm<FMT, false>(a, stream)
  if (b_dtype == QUANTO_HIP_I8) QH_F32_CASE(f32k::W_I8);
  if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_F32_CASE(f32k::W_F8E4M3);
  if (b_dtype == QUANTO_HIP_F8_E5M2) QH_F32_CASE(f32k::W_F8E5M2);
  QH_F32_CASE(f32k::W_F8FNUZ);
#undef QH_F32_CASE
}

int qbytes_mm_gemv_f32(const void* x, const void* w, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype,
                       int out_dtype, hipStream_t stream) {
  if (!qbytes_gemv_f32_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16) return QUANTO_HIP_EALIGN;
  const f32k::Args a{reinterpret_cast<const float*>(x), reinterpret_cast<const uint8_t*>(w), s, nullptr, reinterpret_cast<const float*>(bias),
                     reinterpret_cast<float*>(y), (int)M, (int)N, (int)K, (int)K, 1};
  return qbytes_f32_dispatch<true>(a, b_dtype, stream);
}

int qbytes_mm_f32(const void* x, const void* w, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype,
                  int out_dtype, hipStream_t stream) {
  if (!qbytes_mm_f32_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16) return QUANTO_HIP_EALIGN;
  const f32k::Args a{reinterpret_cast<const float*>(x), reinterpret_cast<const uint8_t*>(w), s, nullptr, reinterpret_cast<const float*>(bias),
                     reinterpret_cast<float*>(y), (int)M, (int)N, (int)K, (int)K, 1};
  return qbytes_f32_dispatch<false>(a, b_dtype, stream);
}

}  // namespace qh
