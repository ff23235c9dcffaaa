// F.conv2d with a quantized weight as an IMPLICIT GEMM (r4): QConv2d.forward (nn/qconv2d.py:54-55) -> F.conv2d on a WeightQBytesTensor /
// WeightQBitsTensor, where the reference dequantizes the whole weight per call (qfallback) and runs a float convolution.
//
//   y[b, n, oh, ow] = scale[n] * sum_{c,i,j} x[b, c, oh*sh - ph + i*dh, ow*sw - pw + j*dw] * w[n, c, i, j]  (+ bias[n])       (8-bit weights)
//   y = conv(x, dequantize(w)) (+ bias), w dequantized with the reference's roundings (tensor/qbits.py:27-49)                  (int4 weights)
//
// GEMM view: M = B*OH*OW output pixels, N = OC, K = cin*KH*KW in the weight's own (c, i, j) order - the axis-0 quantized weight [OC, K] is the
// operand byte for byte (int4: the generic packed layout, byte (p, k) = q[p, k] | q[p + OC/2, k] << 4, groups along K).  Nothing is
// materialised: no im2col tensor, no dequantized weight.
//
// One workgroup = 128 pixels x 128 channels, EIGHT waves (2 x 4, 64 pixels x 32 channels each), K-tiles of 64, two LDS buffers.  The cost of
// this kernel is the gather, not the MFMAs, and the gather is instruction issue (a lone wave per SIMD issues a VALU op every ~8 cycles, two
// waves one every ~4.75 between them: qmm_mfma_large.hip's issue probe), so it is organised around instructions per gathered element:
//   * a thread stages ONE pixel (tile row tid & 127) and two 8-element chunks of it per K-tile: kc = (tid >> 7) + 4 j.  k is therefore uniform
//     across a wave and the 64 lanes of a load are 64 neighbouring pixels (one or two cache lines);
//   * what depends on k only - byte offset of tap (c, i, j) relative to the window's top-left tap, and the tap's number i*KW + j - is computed
//     once per K-tile by 64 threads into an LDS table (two buffers, rides on the K loop's barrier) and read back by broadcast ds_reads;
//   * what depends on the pixel only - its base offset and ONE validity bit per tap (set: the tap lies inside the image; a 32-bit word for windows
//     of up to 31 taps, two 64-bit words up to 127) - lives in two registers.  The one-pixel gather uses BUFFER loads over x with its true size as
//     the range (r5): an element costs add + v_bfe_i32 + or (+ its two-byte load) - the sign-extended bit turns the offset of a padding tap into
//     0xFFFFFFFF, the range check returns 0 for it, and nothing is masked afterwards (r4: global loads of element 0 for those taps, a keep mask
//     accumulated per element and an and + sbfe per element at staging: ~9 VALU per element); the pixel-PAIR gather is described at the kernel;
//   * the loads of a K-tile are issued back to back before the MFMAs of the current tile and packed in pairs (one v_perm_b32 each) after them;
//   * two K-tiles of gather in flight (a second staging register set, r5) were measured and dropped: no gain with one workgroup per CU, a loss
//     with two (profiles/r05_qconv2d_depth2_negative.jsonl) - the kernel pays the instruction stream of its waves, not a load latency
//     (compile-time ablations: QH_CONV_ABLATE below, profiles/r05_qconv2d_ablations*.jsonl).
// History (profiles/r04_qconv2d_*.jsonl, r05_qconv2d_*.jsonl): first form - four pixels per thread, a counted (c, i, j) walk and four compares per
// element, four waves (inside qmm_mfma.hip) - 3.7 us per K-tile whatever M was; the table on four waves 2.0; eight waves 1.85 (r4); buffer-load
// gather, pixel pairs, magic-number table, 8-byte epilogue stores (r5): ~1.2 us per K-tile; DESIGN.md 4.8.
#include <type_traits>

#include "qh_common.h"

#ifndef QH_CONV_ABLATE
#define QH_CONV_ABLATE 0  // timing experiments only (scripts/probes/conv_ablate.hip; WRONG results): 1 no gather loads, 2 no MFMAs / fragment reads, 4 no weight
#endif                      // loads, 8 no gather address arithmetic either, 16 no staging writes, 32 no tap-table fill, 64 (row form) no weight conversion, 128 (row form) no epilogue, 256 / 512 (row form) gather windows 4-byte aligned / one dword per lane


namespace qh {
namespace conv {

constexpr int BM = 128, BN = 128, BK = 64, NT = 512;
constexpr int TILE_BYTES = BM * BK * 2;  // one operand tile in LDS (16 KiB)
constexpr int LDS_BYTES = 2 * 2 * TILE_BYTES + 2 * BK * 8;

enum WFmt { W_I8 = 0, W_F8E4M3 = 1, W_F8E5M2 = 2, W_I4R = 3, W_I2R = 4, W_DENSE16 = 5 };  // W_DENSE16 (row form only): a weight already in the activation dtype
constexpr int planes_of(int fmt) { return fmt == W_I4R ? 2 : (fmt == W_I2R ? 4 : 1); }  // values per packed byte

// byte-aligned 8- and 16-byte loads (K = cin KH KW need not be a multiple of anything: an RGB stem has K = 27 or 147): hipcc lowers them to
// global_load_dwordx2 / x4, which the gfx950 memory pipeline serves at any alignment (unaligned access mode, the HSA default)
struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U2u { uint32_t x, y; };
struct __attribute__((packed, aligned(1))) U1u { uint32_t x; };

// 128-byte rows of eight 16-byte chunks; chunk kc of row r sits at position kc ^ (r & 7): the fragment reads (16 rows x 4 chunks) and the staging
// writes are conflict-free
__device__ __forceinline__ int lds_off(int row, int kc) { return row * (BK * 2) + ((kc ^ (row & 7)) << 4); }

template <int DT>
struct Mma;
template <>
struct Mma<QUANTO_HIP_BF16> {
  using V8 = bf16x8;
  static __device__ __forceinline__ f32x4 run(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <>
struct Mma<QUANTO_HIP_F16> {
  using V8 = f16x8;
  static __device__ __forceinline__ f32x4 run(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

template <int DT>
__device__ __forceinline__ uint32_t pack_rne(float a, float b) {
  using E = Elem<DT>;
  return (uint32_t)__builtin_bit_cast(uint16_t, E::from_f32(a)) | ((uint32_t)__builtin_bit_cast(uint16_t, E::from_f32(b)) << 16);
}

// two floats that are EXACT in the 16-bit type (int8 / fp8 weights) -> one dword with one instruction (v_cvt_pk_bf16_f32 / v_cvt_pkrtz_f16_f32);
// pack_rne converts element by element and ors the halves (two conversions + shift + or)
template <int DT>
__device__ __forceinline__ uint32_t pack_exact(float a, float b) {
  if constexpr (DT == QUANTO_HIP_BF16) {
    bf16x2 r;
    r.x = (__bf16)a;
    r.y = (__bf16)b;
    return __builtin_bit_cast(uint32_t, r);
  } else {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
  }
}

// two 16-bit elements -> one dword.  As a two-element vector of 16-bit integers: hipcc emits ONE v_perm_b32 that takes the low halves of both
// registers; written as lo | hi << 16 on unsigned variables it zero-extends first (a v_and per element)
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack16(short lo, short hi) {
  const s16x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, v);
}

// 16 one-byte weights -> 16 elements of the activation dtype (every int8 / fp8 value is exact in bf16 and fp16)
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
    out[2 * d] = pack_exact<DT>(f0, f1);
    out[2 * d + 1] = pack_exact<DT>(f2, f3);
  }
  c0 = make_uint4(out[0], out[1], out[2], out[3]);
  c1 = make_uint4(out[4], out[5], out[6], out[7]);
}

// 8 packed bytes -> 8 low-nibble and 8 high-nibble weights dequantized as the reference does: T(T(s q) - z) for float shifts, T(s (q - zp)) for
// integer zero-points, round-to-nearest-even - the LDS operand is the dense weight the reference would have materialised
template <int DT, bool INT_SHIFT>
__device__ __forceinline__ void convert8_i4r(const uint2& w, float s_lo, float z_lo, float s_hi, float z_hi, uint4& lo, uint4& hi) {
  using E = Elem<DT>;
  auto deq = [](uint32_t q, float sc, float z) -> float {
    if constexpr (INT_SHIFT)
      return sc * ((float)q - z);
    else
      return E::to_f32(E::from_f32(sc * (float)q)) - z;
  };
  const uint32_t in[2] = {w.x, w.y};
  uint32_t l[4], h[4];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const uint32_t two = in[d] >> (16 * b);  // bytes 2b, 2b + 1
      l[2 * d + b] = pack_rne<DT>(deq(two & 0xFu, s_lo, z_lo), deq((two >> 8) & 0xFu, s_lo, z_lo));
      h[2 * d + b] = pack_rne<DT>(deq((two >> 4) & 0xFu, s_hi, z_hi), deq((two >> 12) & 0xFu, s_hi, z_hi));
    }
  lo = make_uint4(l[0], l[1], l[2], l[3]);
  hi = make_uint4(h[0], h[1], h[2], h[3]);
}

// 8 packed bytes of an int2 weight -> 8 weights of each of the four planes (bits 2 pl .. 2 pl + 1), dequantized like convert8_i4r
template <int DT, bool INT_SHIFT>
__device__ __forceinline__ void convert8_i2r(const uint2& w, const float (&sc)[4], const float (&z)[4], uint4 (&out)[4]) {
  using E = Elem<DT>;
  auto deq = [](uint32_t q, float s, float zz) -> float {
    if constexpr (INT_SHIFT)
      return s * ((float)q - zz);
    else
      return E::to_f32(E::from_f32(s * (float)q)) - zz;
  };
  const uint32_t in[2] = {w.x, w.y};
#pragma unroll
  for (int pl = 0; pl < 4; ++pl) {
    uint32_t o[4];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const uint32_t two = in[d] >> (16 * b);  // bytes 2b, 2b + 1
        o[2 * d + b] = pack_rne<DT>(deq((two >> (2 * pl)) & 3u, sc[pl], z[pl]), deq((two >> (8 + 2 * pl)) & 3u, sc[pl], z[pl]));
      }
    out[pl] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

struct Args {
  const void* x;        // [B, cin, H, W] activation dtype
  const uint8_t* w;     // 8-bit: [OC, K] bytes; int4: packed [OC/2, K] bytes
  const void* scale;    // 8-bit: [OC]; int4: [OC * G]
  const void* shift;    // int4 only: [OC * G] (activation dtype, or uint8 / int8 zero-points)
  const void* bias;     // [OC] or null
  void* y;              // [B, OC, OH, OW]
  int M, N, K, C, G;    // M = B OH OW, N = OC, K = cin KH KW; int4: group size C, G = K / C groups per channel
  int cin, H, W, KH, KW, OH, OW, sh, sw, ph, pw, dh, dw;
  // K split over blockIdx.z (S > 1): split z multiplies K-tiles [z nk / S, (z + 1) nk / S) and parks its fp32 sums in `partials`
  // ([S][tiles][8 waves][8 fragments][64 lanes] float4: whole lines per store); qconv2d_reduce_kernel adds them in split order and runs the epilogue
  int S;
  float* partials;
  uint32_t khw_magic, kw_magic;  // ceil(2^32 / (KH KW)), ceil(2^32 / KW); 0 when the divisor is 1 (fill_ktab)
  uint32_t kh_magic;             // ceil(2^32 / KH) (row form)
};
static uint32_t div_magic(int d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d); }

// the lane's 4 x 2 accumulator fragments -> output: D row = pixel (lane >> 4) * 4 + r of fragment i, D column = channel lane & 15 of fragment j; NCHW:
// the lane's four rows are four neighbouring pixels of one channel plane - ONE 8-byte store when they lie in one image and the plane size is a
// multiple of 4 (r5; before: four 2-byte stores, each with its own integer division by the plane size - the epilogue and the prologue were a
// third of a (8,128,56,56) -> 128 call, profiles/r05_qconv2d_ablations.jsonl).  m < 2^24 (geometry_ok): m / L through the fp32 reciprocal, corrected.
template <int DT, int PL>
__device__ __forceinline__ void store_tile(const Args& a, const f32x4 (&acc)[4][2], int m0, int nt, int wm, int wn, int lane) {
  using E = Elem<DT>;
  using T = typename E::T;
  T* yg = reinterpret_cast<T*>(a.y);
  const int M = a.M, N = a.N, P = N / (PL > 1 ? PL : 2), L = a.OH * a.OW;
  const float r_l = 1.0f / (float)L;
  const bool vec = (L & 3) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 7) == 0;
  int bq[4], lq[4];  // image and offset inside the plane of the first of the lane's four pixels of fragment i
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4;
    int b = (int)((float)m * r_l), l = m - b * L;
    if (l < 0) {
      --b;
      l += L;
    } else if (l >= L) {
      ++b;
      l -= L;
    }
    bq[i] = b;
    lq[i] = l;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int tc = wn * 32 + j * 16 + (lane & 15);
    int n;
    if constexpr (PL > 1) {  // the tile's 128 columns are (128 / PL) packed rows x PL planes; plane pl holds channels pl * P + p
      constexpr int RPT = BN / PL;
      const int p = nt * RPT + (tc % RPT);
      n = p < P ? p + (tc / RPT) * P : -1;
    } else {
      n = nt * BN + tc;
      n = n < N ? n : -1;
    }
    if (n < 0) continue;
    float sc = 1.f;
    if (PL == 1 && a.scale != nullptr) sc = E::to_f32(reinterpret_cast<const T*>(a.scale)[n]);  // (no scale: a dense weight)
    const bool has_bias = a.bias != nullptr;
    const float bv = has_bias ? E::to_f32(reinterpret_cast<const T*>(a.bias)[n]) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4;
      if (m >= M) continue;
      T out[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] * sc;
        asm volatile("" : "+v"(v));  // product rounded to fp32 first, with and without bias (no single-rounding v_fma_mixlo_f16)
        if (has_bias) v = E::to_f32(E::from_f32(v)) + bv;  // the reference's order: rounded convolution output + bias, rounded again
        out[r] = E::from_f32(v);
      }
      T* dst = yg + ((size_t)bq[i] * N + n) * L + lq[i];
      if (vec && m + 3 < M) {  // (L % 4 == 0 and m % 4 == 0: the four pixels are in one image, 8-byte aligned)
        *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(out);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (m + r < M) {
            int bb = bq[i], ll = lq[i] + r;
            while (ll >= L) {  // an image ends inside the lane's four pixels (planes of fewer than 4 pixels: more than once)
              ll -= L;
              ++bb;
            }
            yg[((size_t)bb * N + n) * L + ll] = out[r];
          }
      }
    }
  }
}

// The same epilogue through LDS (r5): store_tile's lanes hold four pixels of ONE channel each - a wave's store touches 16 channel planes with 32
// bytes apiece.  Here the tile is first laid out [channel][pixel] in LDS (rows of 272 bytes), then 16 neighbouring lanes store the 256 contiguous
// bytes a channel plane gets from this tile: four full lines per wave-store.  Needs planes of a multiple of 8 pixels (a thread's 8 pixels lie in one
// image, 16-byte aligned) and a 16-byte aligned y; values identical to store_tile's.  All 512 threads; `stage` >= 34816 bytes, free to overwrite.
constexpr int EPI_ROW = 272;
template <int DT, int PL>
__device__ __forceinline__ void store_tile_lds(const Args& a, const f32x4 (&acc)[4][2], int m0, int nt, int wm, int wn, int lane, int tid, uint8_t* stage) {
  using E = Elem<DT>;
  using T = typename E::T;
  const int M = a.M, N = a.N, P = N / (PL > 1 ? PL : 2), L = a.OH * a.OW;
  auto channel_of = [&](int tc) -> int {  // tile column -> output channel (-1: none)
    if constexpr (PL > 1) {
      constexpr int RPT = BN / PL;
      const int p = nt * RPT + (tc % RPT);
      return p < P ? p + (tc / RPT) * P : -1;
    } else {
      const int n = nt * BN + tc;
      return n < N ? n : -1;
    }
  };
  __syncthreads();  // the K loop's last fragment reads
  const bool has_bias = a.bias != nullptr;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int tc = wn * 32 + j * 16 + (lane & 15);
    const int n = channel_of(tc);
    float sc = 1.f, bv = 0.f;
    if (n >= 0) {
      if (PL == 1 && a.scale != nullptr) sc = E::to_f32(reinterpret_cast<const T*>(a.scale)[n]);
      if (has_bias) bv = E::to_f32(reinterpret_cast<const T*>(a.bias)[n]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      T out[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] * sc;
        asm volatile("" : "+v"(v));  // (store_tile: the product rounded to fp32 first)
        if (has_bias) v = E::to_f32(E::from_f32(v)) + bv;
        out[r] = E::from_f32(v);
      }
      *reinterpret_cast<uint2*>(stage + tc * EPI_ROW + (wm * 64 + i * 16 + (lane >> 4) * 4) * 2) = *reinterpret_cast<const uint2*>(out);
    }
  }
  __syncthreads();
  // thread -> 8 pixels (piece tid & 15) of channel pass * 32 + (tid >> 4)
  const int m = m0 + (tid & 15) * 8;
  if (m >= M) return;
  const int b = m / L, l = m - b * L;
  T* yg = reinterpret_cast<T*>(a.y);
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int tc = pass * 32 + (tid >> 4);
    const int n = channel_of(tc);
    if (n < 0) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(stage + tc * EPI_ROW + (tid & 15) * 16);
    T* dst = yg + ((size_t)b * N + n) * L + l;
    if (m + 7 < M) {
      *reinterpret_cast<uint4*>(dst) = v;
    } else {  // (M is a multiple of L, L of 8: never taken; kept for safety)
      const T* e = reinterpret_cast<const T*>(&v);
      for (int r = 0; r < 8; ++r)
        if (m + r < M) dst[r] = e[r];
    }
  }
}
__device__ __forceinline__ bool epilogue_through_lds(const Args& a) { return ((a.OH * a.OW) & 7) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0; }

// WIDE: windows of 32 .. 127 taps (two 64-bit mask words); the narrow form keeps bit 31 (WIDE: bit 127) of the mask free as the "no such k" tap of
// a ragged last K-tile.
// PAIR (r5): a thread gathers TWO neighbouring output pixels of one row (ow even, ow + 1) with ONE 4-byte load per tap - half the load
// instructions per tile, and the load instructions a workgroup pushes through its CU's address unit are what bounds this kernel (two K-tiles of
// gather in flight instead of one changed nothing, profiles/r05_qconv2d_depth2_negative.jsonl).  Needs stride 1 along the width (the two pixels'
// taps are neighbours in memory) and an even OW (a pair never straddles an output row).  Thread (pair tid & 63, chunk kc = tid >> 6 - k is still
// uniform across a wave, whose 64 loads now cover 128 neighbouring pixels).  Borders: per tap a validity bit for each of the two pixels; left
// border (iw = -1 for the first pixel only): load one element further right and shift the dword up by 16; right border (iw = W for the second
// pixel only): load one element further left and shift down; neither valid: element 0 and a zero selector.  The shifts are one v_perm_b32 with
// a per-lane selector; the loads stay inside x whatever the tap (global loads at 2-byte alignment: unaligned access mode, r4 probe).
template <int DT, int FMT, bool INT_SHIFT, bool WIDE, bool PAIR>
__global__ void __launch_bounds__(NT, 2) qconv2d_mfma_kernel(const Args a) {
  constexpr int PL = planes_of(FMT);      // > 1: a tile's 128 columns are 128 / PL packed rows x PL planes
  constexpr int RPT = BN / PL;            // packed rows per tile
  constexpr int NO_TAP = WIDE ? 127 : 31;
  constexpr int NCH = PAIR ? 1 : 2;       // 8-element chunks of a K-tile per thread
  using E = Elem<DT>;
  using T = typename E::T;
  using V8 = typename Mma<DT>::V8;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // [2 buffers][A tile | B tile] [2 tap tables]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // 64 pixels x 32 channels per wave
  const int m0 = blockIdx.y * BM, nt = blockIdx.x;
  const int M = a.M, N = a.N, K = a.K;
  const int S = a.S, sp = blockIdx.z;
  const int nk_all = (K + BK - 1) / BK;  // the last K-tile may be ragged: k >= K gathers nothing and multiplies zero weights
  const int kt_lo = (int)((long)sp * nk_all / S), nk = (int)((long)(sp + 1) * nk_all / S) - kt_lo;  // this split's K-tiles: kt_lo + t, t = 0 .. nk - 1
  const int P = N / (PL > 1 ? PL : 2);  // packed rows (int4 / int2)
  const uint8_t* xb = reinterpret_cast<const uint8_t*>(a.x);
  // x as a raw buffer whose range is its true size (geometry_ok: < 2^31 bytes): an offset of 0xFFFFFFFF reads as 0
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)(((long)M / (a.OH * a.OW)) * a.cin * a.H * a.W * 2), 0x00020000);

  // ---- the thread's pixel (PAIR: its two pixels m, m + 1) ------------------------------------------------------------------------------------
  // px_off: byte offset of input element (b, 0, oh sh, ow sw) - the window's top-left tap shifted right / down by the padding.
  // Bit i KW + j of a validity word (64 taps per word; narrow: 32 bits) is SET when tap (i, j) of the pixel's window lies inside the image.
  uint32_t px_off;
  uint64_t ok_a0 = 0, ok_a1 = 0, ok_b0 = 0, ok_b1 = 0;  // first / second pixel of the pair, taps 0..63 / 64..127 (scalars: a dynamically indexed array would live in scratch)
  {
    const int L = a.OH * a.OW;
    int m = PAIR ? m0 + 2 * (tid & 63) : m0 + (tid & 127);
    m = m < M ? m : (PAIR ? M - 2 : M - 1);
    const int b = m / L, l = m - b * L, oh = l / a.OW, ow = l - oh * a.OW;
    const int ih0 = oh * a.sh - a.ph, iw0 = ow * a.sw - a.pw;
    px_off = 2u * (uint32_t)(b * a.cin * a.H * a.W + oh * a.sh * a.W + ow * a.sw);
    for (int ki = 0; ki < a.KH; ++ki)
      for (int kj = 0; kj < a.KW; ++kj) {
        const int ih = ih0 + ki * a.dh, iw = iw0 + kj * a.dw;
        const int t = ki * a.KW + kj;
        const uint64_t bit = 1ull << (t & 63);
        const bool row_ok = ih >= 0 && ih < a.H;
        const bool hi = WIDE && t >= 64;
        if (row_ok && iw >= 0 && iw < a.W) {
          ok_a0 |= hi ? 0ull : bit;
          ok_a1 |= hi ? bit : 0ull;
        }
        if (PAIR && row_ok && iw + 1 >= 0 && iw + 1 < a.W) {  // (stride 1 along the width)
          ok_b0 |= hi ? 0ull : bit;
          ok_b1 |= hi ? bit : 0ull;
        }
      }
  }
  // -1 when tap `tp` of pixel `px` of the pair lies inside the image, else 0
  auto tap_ok = [&](int px, int tp) -> int {
    const uint64_t w0 = px ? ok_b0 : ok_a0, w1 = px ? ok_b1 : ok_a1;  // (px is a literal at every call)
    if constexpr (WIDE)
      return -(int)((((tp & 64) ? w1 : w0) >> (tp & 63)) & 1ull);
    else
      return __builtin_amdgcn_sbfe((uint32_t)w0, tp, 1);
  };
  int2* ktab = reinterpret_cast<int2*>(smem + 2 * 2 * TILE_BYTES);  // [2][64] {byte offset relative to px_off (signed), tap number}
  // k -> (channel, tap row, tap column): two divisions by small run-time constants per table entry, done by ONE wave per K-tile while the other
  // seven wait for it at the barrier.  As integer divisions they were ~60 of that wave's instructions per tile; here q = mulhi(n, ceil(2^32 / d)),
  // exact for n < 2^24 and d <= 127 (n (M d - 2^32) < 2^24 * 127 < 2^32); the magic numbers come from the host (0: d = 1)
  auto div_small = [](int n, int d, uint32_t magic, int& rem) {
    const int q = magic ? (int)__umulhi((uint32_t)n, magic) : n;
    rem = n - q * d;
    return q;
  };
  auto fill_ktab = [&](int t) {
    if (tid < BK && !((QH_CONV_ABLATE & 32) && t > 1)) {
      const int k = (kt_lo + t) * BK + tid;
      int rem, kj;
      const int ci = div_small(k, a.KH * a.KW, a.khw_magic, rem);
      const int ki = div_small(rem, a.KW, a.kw_magic, kj);
      ktab[(t & 1) * BK + tid] = k < K ? make_int2(2 * ((ci * a.H + ki * a.dh) * a.W + kj * a.dw - (a.ph * a.W + a.pw)), rem) : make_int2(0, NO_TAP);
    }
  };

  // ---- staging registers --------------------------------------------------------------------------------------------------------------------
  // gathered elements of the K-tile in flight (taps over the padding read 0).  16-bit variables on purpose: as uint32_t the zero-extension (a v_and
  // per element) sits next to the LOAD, hipcc schedules it early and waits for the newest loads right after issuing them
  short g_raw[2][8];
  uint32_t g_pair[8], g_sel[8];      // PAIR: the dword of both pixels per tap and the v_perm selector that realigns / zeroes it
  uint4 rw;                          // 8-bit: 16 weights of row tid >> 2, part tid & 3; int4: 8 packed bytes (rw.x, rw.y) of packed row tid >> 3, part tid & 7
  float rs[4] = {0.f, 0.f, 0.f, 0.f}, rz[4] = {0.f, 0.f, 0.f, 0.f};  // int4 / int2: scale / shift of the thread's packed row (per plane) in the group of its 8 k
  auto issue_loads = [&](int t) {
    const int k0 = (kt_lo + t) * BK;
#pragma unroll
    for (int j = 0; j < ((QH_CONV_ABLATE & 8) ? 0 : NCH); ++j) {
      const int kc = PAIR ? wave : __builtin_amdgcn_readfirstlane(tid >> 7) + 4 * j;
      const int4* tp = reinterpret_cast<const int4*>(ktab + (t & 1) * BK + kc * 8);
      const int4 t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3];
      const int off[8] = {t0.x, t0.z, t1.x, t1.z, t2.x, t2.z, t3.x, t3.z}, tap[8] = {t0.y, t0.w, t1.y, t1.w, t2.y, t2.w, t3.y, t3.w};
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if constexpr (PAIR) {
          const int va = tap_ok(0, tap[q]), vb = tap_ok(1, tap[q]);  // -1: valid
          // first pixel over the left border (va = 0, vb = -1): + 2 bytes; second pixel over the right border (va = -1, vb = 0): - 2 bytes
          const uint32_t addr = (px_off + (uint32_t)off[q] + (uint32_t)(2 * (va - vb))) & (uint32_t)(va | vb);
          constexpr uint32_t IDENT = 0x07060504u, UP16 = 0x05040c0cu, DOWN16 = 0x0c0c0706u, ZERO = 0x0c0c0c0cu;
          uint32_t sel = ((uint32_t)va & IDENT) | (~(uint32_t)va & UP16);     // first pixel invalid: its half becomes 0, the second pixel's element moves up
          sel = ((uint32_t)vb & sel) | (~(uint32_t)vb & DOWN16);             // second pixel invalid: the first pixel's element moves down
          sel = ((uint32_t)(va | vb) & sel) | (~(uint32_t)(va | vb) & ZERO);   // neither
          g_sel[q] = sel;
          g_pair[q] = (QH_CONV_ABLATE & 1) ? addr : reinterpret_cast<const U1u*>(xb + addr)->x;
        } else {
          // -1: over the padding -> offset 0xFFFFFFFF -> out of range -> 0
          g_raw[j][q] = (short)__builtin_amdgcn_raw_buffer_load_b16(xrsrc, (px_off + (uint32_t)off[q]) | ~(uint32_t)tap_ok(0, tap[q]), 0, 0);
        }
      }
    }
    if constexpr (PL > 1) {
      // packed sub-byte weight: thread (row tid >> 3 of the tile's RPT packed rows, part tid & 7) takes 8 packed bytes = 8 k of every plane
      if (PL == 2 || tid < RPT * 8) {
        int p = nt * RPT + (tid >> 3);
        p = p < P ? p : P - 1;
        const int kb = k0 + (tid & 7) * 8;
        const uint8_t* src = a.w + (size_t)p * K + kb;
        uint2 v = make_uint2(0u, 0u);
        if (kb + 8 <= K) {
          const U2u u = *reinterpret_cast<const U2u*>(src);
          v = make_uint2(u.x, u.y);
        } else {  // ragged end of the last K-tile: byte by byte, nothing is read beyond the row
          for (int b = 0; b < 8; ++b)
            if (kb + b < K) (b < 4 ? v.x : v.y) |= (uint32_t)src[b] << (8 * (b & 3));
        }
        rw = make_uint4(v.x, v.y, 0u, 0u);
        int g = kb / a.C;  // the 8 k of a chunk lie in one group (C % 8 == 0, or one group per channel)
        g = g < a.G ? g : a.G - 1;
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {
          const size_t idx = (size_t)(pl * P + p) * a.G + g;
          rs[pl] = E::to_f32(reinterpret_cast<const T*>(a.scale)[idx]);
          if constexpr (INT_SHIFT)
            rz[pl] = (float)(int8_t) reinterpret_cast<const uint8_t*>(a.shift)[idx];
          else
            rz[pl] = E::to_f32(reinterpret_cast<const T*>(a.shift)[idx]);
        }
      }
    } else {
      int n = nt * BN + (tid >> 2);
      n = n < N ? n : N - 1;
      const int kb = k0 + (tid & 3) * 16;
      const uint8_t* src = a.w + (size_t)n * K + kb;
      if (QH_CONV_ABLATE & 4) {
        rw = make_uint4(kb, kb, kb, kb);
      } else if (kb + 16 <= K) {
        const U4u u = *reinterpret_cast<const U4u*>(src);
        rw = make_uint4(u.x, u.y, u.z, u.w);
      } else {  // ragged end of the last K-tile: zero bytes (fp8 0.0, int8 0) behind K, nothing is read beyond the row
        uint32_t d[4] = {0u, 0u, 0u, 0u};
        for (int b = 0; b < 16; ++b)
          if (kb + b < K) d[b >> 2] |= (uint32_t)src[b] << (8 * (b & 3));
        rw = make_uint4(d[0], d[1], d[2], d[3]);
      }
    }
  };
  auto write_lds = [&](int buf) {
    if (QH_CONV_ABLATE & 16) {  // keep the loaded registers alive
      if constexpr (PAIR) {
        uint32_t t = rw.x ^ rw.y ^ rw.z ^ rw.w;
#pragma unroll
        for (int q = 0; q < 8; ++q) t ^= g_pair[q] ^ g_sel[q];
        if (t == 0x12345678u) smem[tid] = 1;
      }
      return;
    }
    uint8_t* sa = smem + buf * 2 * TILE_BYTES;
    uint8_t* sb = sa + TILE_BYTES;
    if constexpr (PAIR) {
      uint32_t d[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) d[q] = __builtin_amdgcn_perm(g_pair[q], 0u, g_sel[q]);  // [first pixel | second pixel] of tap q, realigned, padding zeroed
      const int row = 2 * (tid & 63);
      // first pixel: the low halves of the eight dwords; second pixel: the high halves
      *reinterpret_cast<uint4*>(sa + lds_off(row, wave)) =
          make_uint4(__builtin_amdgcn_perm(d[1], d[0], 0x05040100u), __builtin_amdgcn_perm(d[3], d[2], 0x05040100u),
                     __builtin_amdgcn_perm(d[5], d[4], 0x05040100u), __builtin_amdgcn_perm(d[7], d[6], 0x05040100u));
      *reinterpret_cast<uint4*>(sa + lds_off(row + 1, wave)) =
          make_uint4(__builtin_amdgcn_perm(d[1], d[0], 0x07060302u), __builtin_amdgcn_perm(d[3], d[2], 0x07060302u),
                     __builtin_amdgcn_perm(d[5], d[4], 0x07060302u), __builtin_amdgcn_perm(d[7], d[6], 0x07060302u));
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        *reinterpret_cast<uint4*>(sa + lds_off(tid & 127, (tid >> 7) + 4 * j)) =
            make_uint4(pack16(g_raw[j][0], g_raw[j][1]), pack16(g_raw[j][2], g_raw[j][3]), pack16(g_raw[j][4], g_raw[j][5]), pack16(g_raw[j][6], g_raw[j][7]));
    }
    if constexpr (PL == 2) {
      uint4 lo, hi;
      convert8_i4r<DT, INT_SHIFT>(make_uint2(rw.x, rw.y), rs[0], rz[0], rs[1], rz[1], lo, hi);
      const int row = tid >> 3, part = tid & 7;
      *reinterpret_cast<uint4*>(sb + lds_off(row, part)) = lo;
      *reinterpret_cast<uint4*>(sb + lds_off(64 + row, part)) = hi;
    } else if constexpr (PL == 4) {
      if (tid < RPT * 8) {
        uint4 o[4];
        convert8_i2r<DT, INT_SHIFT>(make_uint2(rw.x, rw.y), rs, rz, o);
        const int row = tid >> 3, part = tid & 7;
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) *reinterpret_cast<uint4*>(sb + lds_off(pl * RPT + row, part)) = o[pl];
      }
    } else {
      uint4 c0, c1;
      convert16<DT, FMT>(rw, c0, c1);
      const int row = tid >> 2, part = tid & 3;
      *reinterpret_cast<uint4*>(sb + lds_off(row, 2 * part)) = c0;
      *reinterpret_cast<uint4*>(sb + lds_off(row, 2 * part + 1)) = c1;
    }
  };

  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  fill_ktab(0);
  if (nk > 1) fill_ktab(1);
  __syncthreads();
  issue_loads(0);
  write_lds(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) issue_loads(kt + 1);
    // table of tile kt + 2 into the buffer whose last reader was tile kt's gather (an iteration ago); visible after this iteration's barrier
    if (kt + 2 < nk) fill_ktab(kt + 2);
    const uint8_t* sa = smem + cur * 2 * TILE_BYTES;
    const uint8_t* sb = sa + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < ((QH_CONV_ABLATE & 2) ? 0 : 2); ++kk) {
      V8 fa[4], fb[2];
      const int kc = kk * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const V8*>(sa + lds_off(wm * 64 + i * 16 + (lane & 15), kc));
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const V8*>(sb + lds_off(wn * 32 + j * 16 + (lane & 15), kc));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Mma<DT>::run(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < nk) write_lds(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  if (S > 1) {  // park the partial sums: one 1 KiB store per wave and fragment
    f32x4* mine = reinterpret_cast<f32x4*>(a.partials) + ((size_t)(sp * gridDim.y + blockIdx.y) * gridDim.x + nt) * (8 * 8 * 64) + (wave * 8) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) mine[(i * 2 + j) * 64] = acc[i][j];
    return;
  }
  if (epilogue_through_lds(a))
    store_tile_lds<DT, PL>(a, acc, m0, nt, wm, wn, lane, tid, smem);
  else
    store_tile<DT, PL>(a, acc, m0, nt, wm, wn, lane);
}

// ---- ROW form (r5): windows three taps wide at dilation 1 along the width (pixel pairs: stride 1, even OW; else one pixel per thread) ------------
// The gather above pays ~3.4 VALU instructions and half a load per gathered element, and the kernel is bound by exactly that instruction stream
// (DESIGN 4.8).  For the layers that dominate convolutional networks - 3 x 3 (any KH x 3) windows walked at stride 1 - the taps (c, i, 0..2) of
// two neighbouring output pixels are FOUR neighbouring input elements: one 8-byte load per (pixel pair, window row r = c KH + i) brings six
// operand elements.  A K-tile is 32 window rows = 96 k; inside a tile k is ordered [tap j][row] (any order both operands share is a GEMM), so
// the staging is a 4 x 4 transposition of 16-bit elements:
//   * wave w, lane p: pixel pair (m0 + 2p, + 1), rows 4w .. 4w + 3 of the tile.  Everything that depends on the row - channel, tap row, byte
//     offset, "behind the last row" - is wave-uniform and computed on the SCALAR unit (no table in LDS, no wave filling it while seven wait);
//     per row the vector unit adds the pixel's base, extracts the row's padding bit (v_bfe_i32) and ors both into the offset: a row over the
//     top / bottom padding becomes offset 0xFFFFFFFF of a range-checked buffer load and reads zeros;
//   * left / right padding is per PIXEL PAIR, not per tap: the 8-byte window is clamped into the image row and two per-thread v_perm selectors
//     realign it and zero the elements over the padding (N0 = e0 e1, N1 = e2 e3: first pixel e0 e1 e2, second pixel e1 e2 e3);
//   * eight v_perm with literal selectors transpose the four rows x four elements into P0 .. P3 (element q of rows 4w .. 4w + 3, 8 bytes each);
//     the first pixel stores P0 P1 P2 into the tap blocks of its LDS row, the second P1 P2 P3: 16 VALU + 6 ds_write_b64 for 24 elements.
//   * weights: thread (channel tid >> 2, part tid & 3) loads 24 contiguous bytes = rows 8 part .. + 7 x 3 taps, converts them during the MFMA
//     phase (any byte of a register is a free operand select) and stores 16 bytes per tap block.
// What it costs (compile-time ablations on (8,C,56,56) -> 128, one workgroup per CU, profiles/r05_qconv2d_rows_ablations*.jsonl): ~1.27 us per
// 96-deep K-tile at C = 128 (the tap gather: 1.43 per 64) - MFMAs + fragment reads 0.28, loads 0.17, staging stores 0.12, conversion 0.03, the bare
// loop (two barriers, scalar row arithmetic, v_perm) 0.28, and ~0.4 that only shows with everything present: the phases of a tile run one after
// the other inside a workgroup.  More waves do not change that (sixteen waves in two groups on alternating tiles, in step or one phase apart:
// level at 196 tiles, 3-23 % slower elsewhere - profiles/r05_qconv2d_rows_two_groups_negative.jsonl); a second workgroup on the CU does (~0.8).
// LDS: [128 rows][224 bytes] per operand (96 k = 192 bytes + 32 of padding; 16-byte slot c of row r at c ^ (r >> 2 & 3): conflict-free
// ds_read_b128 fragments, two-way staging stores - scripts/models/conv_rows_model.py), ONE buffer and two barriers per K-tile: 56 KiB, two
// workgroups per CU.  Needs K = 3 cin KH to be a multiple of 8 (8-byte weight pieces never straddle a row's end), KH <= 31, W >= 4.
// (The description above is the pixel-PAIR mapping; the one-pixel mapping for other strides / an odd OW is described at the kernel.)
namespace rows {
constexpr int RT = 32, BKR = 3 * RT;  // window rows / k per K-tile
constexpr int RS = 224;               // LDS bytes per operand row
constexpr int OP_BYTES = BM * RS;
constexpr int LDS_BYTES = 2 * OP_BYTES;
}  // namespace rows

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int FMT>
__device__ __forceinline__ float byte_to_f32(uint32_t d, int b) {  // b: a literal after unrolling
  if constexpr (FMT == W_I8) {
    return (float)(int8_t)((d >> (8 * b)) & 0xFFu);
  } else if constexpr (FMT == W_F8E4M3) {
    switch (b) {
      case 0: return __builtin_amdgcn_cvt_f32_fp8((int)d, 0);
      case 1: return __builtin_amdgcn_cvt_f32_fp8((int)d, 1);
      case 2: return __builtin_amdgcn_cvt_f32_fp8((int)d, 2);
      default: return __builtin_amdgcn_cvt_f32_fp8((int)d, 3);
    }
  } else {
    switch (b) {
      case 0: return __builtin_amdgcn_cvt_f32_bf8((int)d, 0);
      case 1: return __builtin_amdgcn_cvt_f32_bf8((int)d, 1);
      case 2: return __builtin_amdgcn_cvt_f32_bf8((int)d, 2);
      default: return __builtin_amdgcn_cvt_f32_bf8((int)d, 3);
    }
  }
}

// SINGLE (r5): ONE output pixel per thread (tile row tid & 127) and EIGHT window rows (8 (tid >> 7) ..): any stride along the width and odd OW -
// the downsampling 3 x 3 layers and 7 x 7 / 13 x 13 feature maps the pair form cannot take.  An 8-byte window still brings the pixel's three taps
// of a row (twice the load instructions of the pair form per element, a third fewer than the tap gather), the transposition is 8 rows x 3 elements
// into three 16-byte pieces.
// DB (r6): TWO LDS buffers (112 KiB: one workgroup per CU) and ONE barrier per K-tile - tile t+1 is staged into the other buffer while slower waves are still
// multiplying tile t, so the phases of a tile overlap across the waves of the workgroup instead of running one after the other ("the phases of a
// tile run one after the other inside a workgroup ... a second workgroup on the CU does [help]").  For grids that leave every workgroup a CU of its own anyway.
template <int DT, int FMT, bool BUF, bool SINGLE, bool DB = false>
__global__ void __launch_bounds__(NT, DB ? 1 : 2) qconv2d_rows_kernel(const Args a) {
  using namespace rows;
  constexpr int NR = SINGLE ? 8 : 4;  // window rows per thread and K-tile
  using V8 = typename Mma<DT>::V8;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // [A tile | B tile]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // 64 pixels x 32 channels per wave
  const int m0 = blockIdx.y * BM, nt = blockIdx.x;
  const int M = a.M, N = a.N, K = a.K;
  const int R = a.cin * a.KH;  // window rows
  const int S = a.S, sp = blockIdx.z;
  const int nk_all = (R + RT - 1) / RT;
  const int kt_lo = (int)((long)sp * nk_all / S), nk = (int)((long)(sp + 1) * nk_all / S) - kt_lo;  // >= 1 (the launcher keeps S <= nk_all)
  const uint8_t* xb = reinterpret_cast<const uint8_t*>(a.x);
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)(((long)M / (a.OH * a.OW)) * a.cin * a.H * a.W * 2), 0x00020000);
  constexpr int WB = FMT == W_DENSE16 ? 2 : 1;  // bytes per weight
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.w), 0, N * K * WB, 0x00020000);

  // ---- the lane's pixel pair (the same in every wave) ------------------------------------------------------------------------------------------
  int px_base;            // byte offset of element (b, 0, oh sh - ph, ws): the window's first row, clamped first column
  uint32_t nok = 0x80000000u;  // bit i: tap row i of the pair lies over the top / bottom padding; bit 31 (KH <= 31): rows behind the last one
  uint32_t selN0, selN1;  // v_perm selectors: the clamped window -> (e0, e1) and (e2, e3), elements over the left / right padding zeroed
  {
    const int L = a.OH * a.OW;
    int m = SINGLE ? m0 + (tid & 127) : m0 + 2 * lane;
    m = m < M ? m : (SINGLE ? M - 1 : M - 2);  // (pairs: M is even, OW is)
    const int b = m / L, l = m - b * L, oh = l / a.OW, ow = l - oh * a.OW;
    const int iw0 = ow * a.sw - a.pw;  // column of e0; e_j at iw0 + j: the (first) pixel takes e0 e1 e2, the second pixel of a pair (sw = 1) e1 e2 e3
    int ws = iw0 < 0 ? 0 : iw0;
    ws = ws > a.W - 4 ? a.W - 4 : ws;  // an element inside the image is inside the clamped window (W >= 4)
    const int delta = iw0 - ws;
    px_base = 2 * ((b * a.cin * a.H + (oh * a.sh - a.ph)) * a.W + ws);
    for (int i = 0; i < a.KH; ++i) {
      const int ih = oh * a.sh - a.ph + i * a.dh;
      if (ih < 0 || ih >= a.H) nok |= 1u << i;
    }
    auto half_sel = [&](int j) -> uint32_t {  // window position q = bytes 2q, 2q + 1 of {D1, D0}
      const int iw = iw0 + j, q = j + delta;
      return iw >= 0 && iw < a.W ? (uint32_t)(((2 * q + 1) << 8) | (2 * q)) : 0x0c0cu;
    };
    selN0 = half_sel(0) | (half_sel(1) << 16);
    selN1 = half_sel(2) | ((SINGLE ? 0x0c0cu : half_sel(3)) << 16);
  }

  // ---- LDS addresses (constant over the K loop: one buffer) --------------------------------------------------------------------------------------
  // rows 2 lane and 2 lane + 1 share their swizzle; tap block j at + 64 j (the swizzle only touches the slot inside a block)
  // (SINGLE: row tid & 127, eight rows = the 16-byte slot tid >> 7 of every tap block)
  const uint32_t awr = SINGLE ? (uint32_t)((tid & 127) * RS + (((tid >> 7) ^ ((tid >> 2) & 3)) << 4))
                              : (uint32_t)((2 * lane) * RS + (((wave >> 1) ^ ((lane >> 1) & 3)) << 4) + (wave & 1) * 8);
  const uint32_t bwr = (uint32_t)(OP_BYTES + (tid >> 2) * RS + (((tid & 3) ^ ((tid >> 4) & 3)) << 4));
  const uint32_t frag_slot = (uint32_t)((((lane >> 4) ^ ((lane >> 2) & 3)) << 4));
  const uint32_t ard = (uint32_t)((wm * 64 + (lane & 15)) * RS) + frag_slot;
  const uint32_t brd = (uint32_t)(OP_BYTES + (wn * 32 + (lane & 15)) * RS) + frag_slot;

  // ---- weights: channel tid >> 2, bytes 24 (tid & 3) .. + 23 of the K-tile's 96 ---------------------------------------------------------------------
  int wn_row = nt * BN + (tid >> 2);
  wn_row = wn_row < N ? wn_row : N - 1;
  const uint32_t woff = (uint32_t)(wn_row * K + 24 * (tid & 3)) * WB;
  const int wk = 24 * (tid & 3);

  u32x2 D[NR], wq[3];  // gathered windows / weight bytes of the K-tile in flight
  uint4 wd[3];        // (dense weight: its 24 elements)
  uint4 cw[3];        // its converted weights, one 16-byte piece per tap block
  auto issue = [&](int t) {
    const int T = kt_lo + t;
    const int krem = K - T * BKR;  // bytes of this tile inside the weight row (>= 96 except in a ragged last tile)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const uint32_t off = wk + 8 * j < krem ? woff + (uint32_t)(T * BKR + 8 * j) * WB : 0xFFFFFFFFu;
      if constexpr (FMT == W_DENSE16) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)off, 0, 0));
        wd[j] = make_uint4(v.x, v.y, v.z, v.w);
      } else {
        wq[j] = (QH_CONV_ABLATE & 4) ? u32x2{off, off} : __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(wrsrc, (int)off, 0, 0));
      }
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      // scalar: row -> (channel, tap row) -> byte offset; behind the last row: -1
      const uint32_t r = (uint32_t)(T * RT + (SINGLE ? 8 * (wave >> 1) : 4 * wave) + u);
      const uint32_t c = __umulhi(r, a.kh_magic) + (a.kh_magic ? 0u : r);  // (magic 0: KH = 1)
      const uint32_t i = r - c * (uint32_t)a.KH;
      const uint32_t roff = 2u * ((c * (uint32_t)a.H + i * (uint32_t)a.dh) * (uint32_t)a.W);
      const uint32_t pad = (uint32_t)__builtin_amdgcn_sbfe(nok, (int)r < R ? i : 31u, 1);  // -1: over the padding, or behind the last row (bit 31)
      constexpr uint32_t gone = 0u;
      if constexpr (BUF) {
        const uint32_t addr = ((uint32_t)px_base + roff) | pad | gone;
        if (QH_CONV_ABLATE & 1)
          D[u] = u32x2{addr, addr};
        else if (QH_CONV_ABLATE & 256)  // 4-byte aligned windows (wrong elements, same lines)
          D[u] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, (int)(addr & ~3u), 0, 0));
        else if (QH_CONV_ABLATE & 512)  // one dword per lane
          D[u] = u32x2{(uint32_t)__builtin_amdgcn_raw_buffer_load_b32(xrsrc, (int)(addr & ~3u), 0, 0), addr};
        else
          D[u] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, (int)addr, 0, 0));
      } else {  // global loads (experiments): element 0 for a row that is not there, zeroed afterwards
        const uint32_t keep = ~(pad | gone);
        const U2u v = *reinterpret_cast<const U2u*>(xb + (((uint32_t)px_base + roff) & keep));
        D[u] = u32x2{v.x & keep, v.y & keep};
      }
    }
  };
  // 24 weight bytes in (row, tap) order -> per tap block j the eight rows' values, in the activation dtype (exact)
  auto convert = [&](int j) {
    if constexpr (FMT == W_DENSE16) {  // 24 sixteen-bit elements in (row, tap) order: element 3 rl + j of each of the eight rows, two per v_perm
      const uint32_t in[12] = {wd[0].x, wd[0].y, wd[0].z, wd[0].w, wd[1].x, wd[1].y, wd[1].z, wd[1].w, wd[2].x, wd[2].y, wd[2].z, wd[2].w};
      uint32_t o[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int e0 = 3 * (2 * h) + j, e1 = 3 * (2 * h + 1) + j;
        const uint32_t sel = ((e0 & 1) ? 0x0302u : 0x0100u) | ((e1 & 1) ? 0x07060000u : 0x05040000u);
        o[h] = __builtin_amdgcn_perm(in[e1 >> 1], in[e0 >> 1], sel);
      }
      cw[j] = make_uint4(o[0], o[1], o[2], o[3]);
      return;
    }
    const uint32_t in[6] = {wq[0].x, wq[0].y, wq[1].x, wq[1].y, wq[2].x, wq[2].y};
    if (QH_CONV_ABLATE & 64) {
      cw[j] = make_uint4(in[j], in[j + 1], in[j + 2], in[j + 3]);
      return;
    }
    uint32_t o[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int e0 = 3 * (2 * h) + j, e1 = 3 * (2 * h + 1) + j;
      o[h] = pack_exact<DT>(byte_to_f32<FMT>(in[e0 >> 2], e0 & 3), byte_to_f32<FMT>(in[e1 >> 2], e1 & 3));
    }
    cw[j] = make_uint4(o[0], o[1], o[2], o[3]);
  };
  auto write = [&](auto bo_tag) {
    constexpr int BO = decltype(bo_tag)::value;  // byte offset of the LDS buffer the tile is staged into
    if (QH_CONV_ABLATE & 16) {  // keep the loaded registers alive
      uint32_t t = 0;
#pragma unroll
      for (int u = 0; u < NR; ++u) t ^= D[u].x ^ D[u].y;
#pragma unroll
      for (int j = 0; j < 3; ++j) t ^= cw[j].x ^ cw[j].y ^ cw[j].z ^ cw[j].w;
      if (t == 0x12345678u) smem[tid] = 1;
      return;
    }
    uint32_t n0[NR], n1[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      n0[u] = __builtin_amdgcn_perm(D[u].y, D[u].x, selN0);
      n1[u] = __builtin_amdgcn_perm(D[u].y, D[u].x, selN1);
    }
    constexpr uint32_t LO = 0x05040100u, HI = 0x07060302u;  // the low / high halves of two registers
    if constexpr (SINGLE) {  // tap j of rows 0 .. 7: 16 bytes of tap block j
      uint8_t* sa = smem + BO + awr;
      uint32_t t0[4], t1[4], t2[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        t0[h] = __builtin_amdgcn_perm(n0[2 * h + 1], n0[2 * h], LO);
        t1[h] = __builtin_amdgcn_perm(n0[2 * h + 1], n0[2 * h], HI);
        t2[h] = __builtin_amdgcn_perm(n1[2 * h + 1], n1[2 * h], LO);
      }
      *reinterpret_cast<uint4*>(sa) = make_uint4(t0[0], t0[1], t0[2], t0[3]);
      *reinterpret_cast<uint4*>(sa + 64) = make_uint4(t1[0], t1[1], t1[2], t1[3]);
      *reinterpret_cast<uint4*>(sa + 128) = make_uint4(t2[0], t2[1], t2[2], t2[3]);
      uint8_t* sb = smem + BO + bwr;
#pragma unroll
      for (int j = 0; j < 3; ++j) *reinterpret_cast<uint4*>(sb + 64 * j) = cw[j];
      return;
    }
    const uint2 p0 = make_uint2(__builtin_amdgcn_perm(n0[1], n0[0], LO), __builtin_amdgcn_perm(n0[3], n0[2], LO));
    const uint2 p1 = make_uint2(__builtin_amdgcn_perm(n0[1], n0[0], HI), __builtin_amdgcn_perm(n0[3], n0[2], HI));
    const uint2 p2 = make_uint2(__builtin_amdgcn_perm(n1[1], n1[0], LO), __builtin_amdgcn_perm(n1[3], n1[2], LO));
    const uint2 p3 = make_uint2(__builtin_amdgcn_perm(n1[1], n1[0], HI), __builtin_amdgcn_perm(n1[3], n1[2], HI));
    uint8_t* sa = smem + BO + awr;
    *reinterpret_cast<uint2*>(sa) = p0;
    *reinterpret_cast<uint2*>(sa + 64) = p1;
    *reinterpret_cast<uint2*>(sa + 128) = p2;
    *reinterpret_cast<uint2*>(sa + RS) = p1;
    *reinterpret_cast<uint2*>(sa + RS + 64) = p2;
    *reinterpret_cast<uint2*>(sa + RS + 128) = p3;
    uint8_t* sb = smem + BO + bwr;
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<uint4*>(sb + 64 * j) = cw[j];
  };

  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the MFMA phase, ordered by hand (with a fence behind the loads hipcc otherwise serialises fragment read -> wait -> two MFMAs; without one it
  // sinks the next tile's loads behind the MFMAs): the fragments of k-step kk + 1 are read while the MFMAs of k-step kk run, fences between the steps
  V8 fa[2][4], fb[2][2];
  auto read_frags = [&](int kk, auto bo_tag) {
    constexpr int BO = decltype(bo_tag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[kk & 1][i] = *reinterpret_cast<const V8*>(smem + BO + ard + i * 16 * RS + kk * 64);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[kk & 1][j] = *reinterpret_cast<const V8*>(smem + BO + brd + j * 16 * RS + kk * 64);
  };
  auto mma_phase = [&](auto bo_tag) {
    if (QH_CONV_ABLATE & 2) return;
    read_frags(0, bo_tag);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      if (kk < 2) read_frags(kk + 1, bo_tag);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Mma<DT>::run(fa[kk & 1][i], fb[kk & 1][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, rows::LDS_BYTES>;
  issue(0);
#pragma unroll
  for (int j = 0; j < 3; ++j) convert(j);
  write(B0{});
  __syncthreads();
  if constexpr (DB) {
    // tile t is multiplied out of buffer `cur` while tile t + 1 is staged into `nxt`; the one barrier of the tile orders both hand-overs: nxt complete before
    // anybody reads it, every wave done with cur before the tile after next is staged into it
    auto step = [&](int t, auto cur, auto nxt) {
      issue(t + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_phase(cur);
#pragma unroll
      for (int j = 0; j < 3; ++j) convert(j);
      __builtin_amdgcn_sched_barrier(0);
      write(nxt);
      __syncthreads();
    };
    int t = 0;
    for (; t + 2 < nk; t += 2) {
      step(t, B0{}, B1{});
      step(t + 1, B1{}, B0{});
    }
    if (t + 1 < nk) {
      step(t, B0{}, B1{});
      mma_phase(B1{});
    } else {
      mma_phase(B0{});
    }
  } else {
  for (int t = 0; t + 1 < nk; ++t) {
    // the next tile's loads first; its weights are converted AFTER the MFMAs, so that no vmcnt wait on loads issued a moment ago stands in front
    // of them.  (Measured against hipcc's own order, which interleaved the conversion: the same time within 1-3 % either way,
    // profiles/r05_qconv2d_rows_ablations*.jsonl - the hand-placed order is kept because it does not move with the compiler.)
    issue(t + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_phase(B0{});
#pragma unroll
    for (int j = 0; j < 3; ++j) convert(j);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // every wave has read tile t
    write(B0{});
    __syncthreads();
  }
  mma_phase(B0{});
  }

  if (S > 1) {
    f32x4* mine = reinterpret_cast<f32x4*>(a.partials) + ((size_t)(sp * gridDim.y + blockIdx.y) * gridDim.x + nt) * (8 * 8 * 64) + (wave * 8) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) mine[(i * 2 + j) * 64] = acc[i][j];
    return;
  }
  if (QH_CONV_ABLATE & 128) {  // no epilogue (the accumulators stay alive)
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t != 1.2345e30f) return;
  }
  if (epilogue_through_lds(a))
    store_tile_lds<DT, 1>(a, acc, m0, nt, wm, wn, lane, tid, smem);
  else
    store_tile<DT, 1>(a, acc, m0, nt, wm, wn, lane);
}

// split-K tail: one WAVE per (output tile, wave slot of the tile kernel) adds that slot's eight fragments over the S partial tiles in split order
// (deterministic), four splits' loads in flight together, and runs the epilogue.  (First form: one 512-thread workgroup per tile with one split per
// loop iteration - 14 workgroups each waiting 12 times for a round trip cost more than the convolution itself.)
template <int DT, int PL>
__global__ void __launch_bounds__(64) qconv2d_reduce_kernel(const Args a) {
  const int lane = threadIdx.x, wave = blockIdx.z, S = a.S;
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4* base = reinterpret_cast<const f32x4*>(a.partials) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (8 * 8 * 64) + (wave * 8) * 64 + lane;
  const size_t split_stride = (size_t)gridDim.y * gridDim.x * (8 * 8 * 64);
  for (int sp0 = 0; sp0 < S; sp0 += 4) {
    f32x4 v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int sp = sp0 + u < S ? sp0 + u : S - 1;
#pragma unroll
      for (int f = 0; f < 8; ++f) v[u][f] = base[sp * split_stride + f * 64];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (sp0 + u < S) {
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[f >> 1][f & 1][r] += v[u][f][r];
      }
  }
  store_tile<DT, PL>(a, acc, blockIdx.y * BM, blockIdx.x, wave >> 2, wave & 3, lane);
}

// K split: the tile kernel is bound by its gather per K-tile (~1.9 us per workgroup and K-tile whatever M is), so what matters is how many
// workgroups run at once: split until the grid reaches ~2 workgroups per CU, keeping at least 4 K-tiles per split (r5, after the gather and the
// epilogue got cheaper: profiles/r05_qconv2d_split_sweep.jsonl - 3 per split over-split 26-49-tile grids by 10-14 %).  1 = no split (and no workspace).
static thread_local bool g_last_rows = false;  // the last launch of this thread took the row form (last_kernel() name, tests)

static int pick_split(int64_t M, int64_t N, int64_t K) {
  const int forced = env_int("QUANTO_HIP_CONV_SPLIT", 0);  // experiments
  const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN), nk = (K + BK - 1) / BK;
  if (forced > 0) return (int)(forced <= nk ? forced : nk);
  // measured (profiles/r04_qconv2d_forced_split.jsonl): at 196 tiles a split of 2 costs more in partial sums than the second workgroup per CU
  // brings while K is short (9 K-tiles 23.9 -> 31.9 us, 18 K-tiles 43.0 -> 44.7) and pays from ~32 K-tiles on (192 tiles x 45: 81.6 -> 72.9,
  // 256 tiles x 49: 86.8 -> 84.3)
  if (tiles > 128) return tiles <= 256 && nk >= 32 ? 2 : 1;
  int s = 1;
  while (tiles * (s + 1) <= 512 && nk / (s + 1) >= 4 && s < 64) ++s;
  return s;
}
static size_t split_workspace(int64_t M, int64_t N, int S) { return S <= 1 ? 0 : (size_t)S * ((M + BM - 1) / BM) * ((N + BN - 1) / BN) * (BM * BN * 4); }

// row form: three taps wide, stride 1 / dilation 1 along the width, even OW (QUANTO_HIP_CONV_ROWS=0: the tap gather, 2: global loads - experiments)
// (pixel pairs need stride 1 and an even OW; anything else three taps wide takes one pixel per thread; QUANTO_HIP_CONV_ROWS=3: one pixel per thread everywhere)
static bool rows_eligible(int64_t cin, int64_t KH, int64_t KW, int64_t W, int64_t OW, int sw, int dw) {
  return env_int("QUANTO_HIP_CONV_ROWS", 1) != 0 && KW == 3 && dw == 1 && W >= 4 && KH <= 31 && (cin * KH) % 8 == 0;
}
static bool rows_pairs(int64_t OW, int sw) { return sw == 1 && OW % 2 == 0 && env_int("QUANTO_HIP_CONV_ROWS", 1) != 3; }
template <int DT, int FMT>
static int launch_rows(Args a, int ntiles, int mtiles, hipStream_t stream) {  // a.S: the split the workspace allows; a.partials set
  g_last_rows = true;
  const int nk_rows = (a.cin * a.KH + rows::RT - 1) / rows::RT;
  a.S = a.S < nk_rows ? a.S : nk_rows;
  // two LDS buffers where every workgroup has a CU to itself anyway (QUANTO_HIP_CONV_ROWS_DB: 0 never, 2 always - experiments)
  const int dbk = env_int("QUANTO_HIP_CONV_ROWS_DB", 1);
  const bool db = dbk == 2 || (dbk == 1 && (int64_t)ntiles * mtiles * a.S <= 256 && nk_rows / a.S >= 2);
#define QH_ROWS(BUF, SINGLE, DB)                                                                                                                       \
  do {                                                                                                                                                 \
    constexpr int lds = (DB ? 2 : 1) * rows::LDS_BYTES;                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qconv2d_rows_kernel<DT, FMT, BUF, SINGLE, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    hipLaunchKernelGGL((qconv2d_rows_kernel<DT, FMT, BUF, SINGLE, DB>), dim3(ntiles, mtiles, a.S), dim3(NT), lds, stream, a);                          \
  } while (0)
  if (!rows_pairs(a.OW, a.sw)) {
    if (db) QH_ROWS(true, true, true); else QH_ROWS(true, true, false);
  } else if (env_int("QUANTO_HIP_CONV_ROWS", 1) == 2) {
    QH_ROWS(false, false, false);
  } else {
    if (db) QH_ROWS(true, false, true); else QH_ROWS(true, false, false);
  }
#undef QH_ROWS
  if (a.S > 1) hipLaunchKernelGGL((qconv2d_reduce_kernel<DT, 1>), dim3(ntiles, mtiles, 8), dim3(64), 0, stream, a);
  return launch_status();
}

template <int DT, int FMT, bool INT_SHIFT, bool WIDE, bool PAIR>
static void launch_k(const Args& a, int ntiles, int mtiles, hipStream_t stream) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qconv2d_mfma_kernel<DT, FMT, INT_SHIFT, WIDE, PAIR>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipLaunchKernelGGL((qconv2d_mfma_kernel<DT, FMT, INT_SHIFT, WIDE, PAIR>), dim3(ntiles, mtiles, a.S), dim3(NT), LDS_BYTES, stream, a);
}
template <int DT, int FMT, bool INT_SHIFT, bool WIDE>
static int launch_w(Args a, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  constexpr int PL = planes_of(FMT);
  const int ntiles = PL > 1 ? (a.N / PL + BN / PL - 1) / (BN / PL) : (a.N + BN - 1) / BN, mtiles = (a.M + BM - 1) / BM;
  g_last_rows = false;
  int S = pick_split(a.M, a.N, a.K);
  if (S > 1 && (!workspace || workspace_bytes < split_workspace(a.M, a.N, S) || reinterpret_cast<uintptr_t>(workspace) % 16)) S = 1;
  a.S = S;
  a.partials = reinterpret_cast<float*>(workspace);
  if constexpr (PL == 1 && !WIDE) {
    if (rows_eligible(a.cin, a.KH, a.KW, a.W, a.OW, a.sw, a.dw)) {
      a.S = S;
      return launch_rows<DT, FMT>(a, ntiles, mtiles, stream);
    }
  }
  // two output pixels per load wherever the geometry allows it (QUANTO_HIP_CONV_PAIR=0: experiments)
  const bool pair = a.sw == 1 && a.OW % 2 == 0 && a.W >= 2 && env_int("QUANTO_HIP_CONV_PAIR", 1) != 0;
  if (pair)
    launch_k<DT, FMT, INT_SHIFT, WIDE, true>(a, ntiles, mtiles, stream);
  else
    launch_k<DT, FMT, INT_SHIFT, WIDE, false>(a, ntiles, mtiles, stream);
  if (S > 1) hipLaunchKernelGGL((qconv2d_reduce_kernel<DT, PL>), dim3(ntiles, mtiles, 8), dim3(64), 0, stream, a);
  return launch_status();
}
template <int DT, int FMT, bool INT_SHIFT>
static int launch(Args a, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  return a.KH * a.KW > 31 ? launch_w<DT, FMT, INT_SHIFT, true>(a, workspace, workspace_bytes, stream)
                          : launch_w<DT, FMT, INT_SHIFT, false>(a, workspace, workspace_bytes, stream);
}

static bool geometry_ok(int64_t B, int64_t cin, int64_t H, int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW) {
  const int64_t K = cin * KH * KW;
  // one validity bit per tap (two mask words, one bit kept free); byte offsets into x and element offsets into y / w in 31 bits; grid.y
  return B >= 1 && OH >= 1 && OW >= 1 && K >= 1 && K < (1ll << 24) && KH * KW <= 127 && B * cin * H * W < (1ll << 30) && B * OC * OH * OW < (1ll << 31) &&
         OC * K < (1ll << 31) && (B * OH * OW + BM - 1) / BM <= 65535;  // (the K split is at most 64: grid.z)
}

}  // namespace conv

bool conv2d_last_was_rows() { return conv::g_last_rows; }

bool qbytes_conv2d_supported(int64_t B, int64_t cin, int64_t H, int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int a_dtype,
                             int b_dtype, int out_dtype) {
  const bool bd = b_dtype == QUANTO_HIP_I8 || b_dtype == QUANTO_HIP_F8_E4M3FN || b_dtype == QUANTO_HIP_F8_E5M2;
  return bd && a_dtype == out_dtype && (out_dtype == QUANTO_HIP_BF16 || out_dtype == QUANTO_HIP_F16) && conv::geometry_ok(B, cin, H, W, OC, KH, KW, OH, OW);
}

// scratch bytes the K split of a convolution wants (0: not split); the same for every weight format (128 x 128 tiles either way)
size_t conv2d_workspace(int64_t M, int64_t N, int64_t K) { return conv::split_workspace(M, N, conv::pick_split(M, N, K)); }

// int4 / int2 weights in the row form (r5): the weight is tiny next to the activations and every pixel tile would dequantize all of it again
// (196 times for (8,128,56,56) -> 128: the sub-byte tap kernel spends a third of its time there) - so it is dequantized ONCE into the caller's
// workspace by the fused dequantize_qbits kernel (bit-identical to the reference's dequantize()) and the row form multiplies by that dense weight:
// what the reference does, as two launches, without im2col.  Workspace: [dense weight, 256-byte multiple | K-split partials].
size_t conv2d_dense_weight_bytes(int64_t N, int64_t K) { return ((size_t)N * K * 2 + 255) / 256 * 256; }
bool conv2d_rows_eligible(int64_t cin, int64_t KH, int64_t KW, int64_t W, int64_t OW, int sw, int dw, int64_t OC) {
  return conv::rows_eligible(cin, KH, KW, W, OW, sw, dw) && KH * KW <= 31 && OC * cin * KH * KW < (1ll << 30);
}
int qdense_conv2d_rows(const void* x, const void* wdense, const void* bias, void* y, int64_t B, int64_t cin, int64_t H, int64_t W, int64_t OC, int64_t KH,
                       int64_t KW, int64_t OH, int64_t OW, int sh, int sw, int ph, int pw, int dh, int dw, int dtype, void* workspace, size_t workspace_bytes,
                       hipStream_t stream) {
  if (!conv2d_rows_eligible(cin, KH, KW, W, OW, sw, dw, OC) || !conv::geometry_ok(B, cin, H, W, OC, KH, KW, OH, OW) ||
      (dtype != QUANTO_HIP_BF16 && dtype != QUANTO_HIP_F16))
    return QUANTO_HIP_ENOTSUP;
  conv::Args a{x, reinterpret_cast<const uint8_t*>(wdense), nullptr, nullptr, bias, y, (int)(B * OH * OW), (int)OC, (int)(cin * KH * KW), 0, 0,
               (int)cin, (int)H, (int)W, (int)KH, (int)KW, (int)OH, (int)OW, sh, sw, ph, pw, dh, dw, 1, nullptr, conv::div_magic((int)(KH * KW)), conv::div_magic((int)KW), conv::div_magic((int)KH)};
  using namespace conv;
  int S = pick_split(a.M, a.N, a.K);
  if (S > 1 && (!workspace || workspace_bytes < split_workspace(a.M, a.N, S) || reinterpret_cast<uintptr_t>(workspace) % 16)) S = 1;
  a.S = S;
  a.partials = reinterpret_cast<float*>(workspace);
  const int ntiles = (a.N + BN - 1) / BN, mtiles = (a.M + BM - 1) / BM;
  return dtype == QUANTO_HIP_BF16 ? launch_rows<QUANTO_HIP_BF16, W_DENSE16>(a, ntiles, mtiles, stream) : launch_rows<QUANTO_HIP_F16, W_DENSE16>(a, ntiles, mtiles, stream);
}

int qbytes_conv2d_mfma(const void* x, const void* w, const void* s, const void* bias, void* y, int64_t B, int64_t cin, int64_t H, int64_t W, int64_t OC,
                       int64_t KH, int64_t KW, int64_t OH, int64_t OW, int sh, int sw, int ph, int pw, int dh, int dw, int a_dtype, int b_dtype,
                       int out_dtype, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!qbytes_conv2d_supported(B, cin, H, W, OC, KH, KW, OH, OW, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  const conv::Args a{x, reinterpret_cast<const uint8_t*>(w), s, nullptr, bias, y, (int)(B * OH * OW), (int)OC, (int)(cin * KH * KW), 0, 0,
                     (int)cin, (int)H, (int)W, (int)KH, (int)KW, (int)OH, (int)OW, sh, sw, ph, pw, dh, dw, 1, nullptr, conv::div_magic((int)(KH * KW)), conv::div_magic((int)KW), conv::div_magic((int)KH)};
  using namespace conv;
#define QH_CASE(DT, FMT) return launch<DT, FMT, false>(a, workspace, workspace_bytes, stream)
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

// int4 / int2: group sizes that are multiples of 8 (a staging chunk of 8 k must not straddle groups) or per-channel scales; OC a multiple of the
// values per byte (the planes of the generic packed layout)
bool qbits_conv2d_supported(int64_t B, int64_t cin, int64_t H, int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, const PackedGeom& g,
                            int dtype) {
  return (g.bits == 4 || g.bits == 2) && g.N == OC && g.K == cin * KH * KW && OC % g.vpi == 0 && (g.C % 8 == 0 || g.G == 1) &&
         (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) &&
         OC * g.G < (1ll << 31) && conv::geometry_ok(B, cin, H, W, OC, KH, KW, OH, OW);
}

int qbits_conv2d_mfma(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t B, int64_t cin,
                      int64_t H, int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int sh, int sw, int ph, int pw, int dh, int dw,
                      const PackedGeom& g, int dtype, bool int_shift, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!qbits_conv2d_supported(B, cin, H, W, OC, KH, KW, OH, OW, g, dtype)) return QUANTO_HIP_ENOTSUP;
  const conv::Args a{x, packed, scale, shift, bias, y, (int)(B * OH * OW), (int)OC, (int)(cin * KH * KW), (int)g.C, (int)g.G,
                     (int)cin, (int)H, (int)W, (int)KH, (int)KW, (int)OH, (int)OW, sh, sw, ph, pw, dh, dw, 1, nullptr, conv::div_magic((int)(KH * KW)), conv::div_magic((int)KW), conv::div_magic((int)KH)};
  using namespace conv;
#define QH_CASE(DT, FMT) return int_shift ? launch<DT, FMT, true>(a, workspace, workspace_bytes, stream) : launch<DT, FMT, false>(a, workspace, workspace_bytes, stream)
  if (g.bits == 4) {
    if (dtype == QUANTO_HIP_BF16) QH_CASE(QUANTO_HIP_BF16, W_I4R);
    QH_CASE(QUANTO_HIP_F16, W_I4R);
  }
  if (dtype == QUANTO_HIP_BF16) QH_CASE(QUANTO_HIP_BF16, W_I2R);
  QH_CASE(QUANTO_HIP_F16, W_I2R);
#undef QH_CASE
}

}  // namespace qh
