// F.conv2d with groups = in_channels (depthwise, channel multiplier OC / cin >= 1) and an int8 / fp8 weight (r6).
//
//   y[b, oc, oh, ow] = scale[oc] * sum_{i,j} x[b, oc / mult, oh*sh - ph + i*dh, ow*sw - pw + j*dw] * w[oc, 0, i, j]   (+ bias[oc])
//
// What QConv2d.forward (nn/qconv2d.py:54-55) reaches for a depthwise layer through WeightQBytesTensor's dispatch; the reference dequantizes the weight
// per call (qfallback) and runs a float convolution.  A depthwise convolution has KH*KW products per output and no reuse across channels: there is no GEMM
// in it and nothing for the matrix cores - it is a stencil bound by how fast a CU moves activation rows through its vector L1.  Layout of the work:
//   * a thread owns PX = 4 neighbouring output columns of one output row (one store of 8 bytes), a workgroup of 256 threads a run of such quads (of one (b, oc)
//     plane, or of consecutive small planes): consecutive lanes read consecutive input columns - every activation line is fetched once per workgroup row and re-used from L1 for the
//     KH rows above / below;
//   * the plane's KH*KW weight bytes are decoded once per thread into registers for 3 x 3 / 5 x 5 / 7 x 7 windows (other windows read the tap's byte per tap:
//     one address per workgroup), products and sums in fp32 in (i, j) order, scale applied to the sum, one rounding to the output dtype (then bias + one more rounding, the reference's order) - the
//     arithmetic contract of the dense convolution kernel (qconv_mfma.hip).
#include <type_traits>

#include "qh_common.h"

namespace qh {
namespace dw {

struct Args {
  const void* x;       // [B, C, H, W]
  const uint8_t* w;    // [OC, 1, KH, KW] one byte per weight
  const void* scale;   // [OC]
  const void* bias;    // [OC] or null
  void* y;             // [B, OC, OH, OW]
  int B, C, H, W, OC, mult, KH, KW, OH, OW, sh, sw, ph, pw, dh, dw;
  int quads;           // ceil(OW / 4): column quads per output row
};

constexpr int PX = 4, THREADS = 256;

// KS: compile-time window side for the square windows depthwise layers use (3, 5, 7: weights in registers, tap loops unrolled); 0 = any window, the tap's
// weight byte read per tap (one address per workgroup: an L1 broadcast)
template <int DT, int WDT, int KS>
__global__ void __launch_bounds__(THREADS) qconv2d_depthwise_kernel(const Args a) {
  using E = Elem<DT>;
  using T = typename E::T;
  // one grid over (plane, oh, quad): planes smaller than a workgroup (7 x 7: 14 quads) share workgroups - a lane's neighbours may then belong to the next
  // channel, whose nine weight bytes are another L1 line at worst
  const int64_t item = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  const int per_plane = a.OH * a.quads;
  const int plane = (int)(item / per_plane);     // b * OC + oc
  if (plane >= a.B * a.OC) return;
  const int rem = (int)(item - (int64_t)plane * per_plane);
  const int oc = plane % a.OC, b = plane / a.OC;
  const int c = oc / a.mult;
  const int oh = rem / a.quads, q = rem - oh * a.quads;
  const int ow0 = q * PX;
  const T* xp = reinterpret_cast<const T*>(a.x) + ((size_t)b * a.C + c) * a.H * a.W;
  const uint8_t* wp = a.w + (size_t)oc * a.KH * a.KW;
  const int KH = KS ? KS : a.KH, KW = KS ? KS : a.KW;
  float wr[KS ? KS * KS : 1];
  if constexpr (KS != 0) {
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) wr[t] = decode8<WDT>(wp[t]);
  }
  float acc[PX] = {0.f, 0.f, 0.f, 0.f};
  const int ih0 = oh * a.sh - a.ph;
  const int iwb = ow0 * a.sw - a.pw;
  auto row_taps = [&](int i, const T* row) {
#pragma unroll
    for (int j = 0; j < (KS ? KS : 1 << 20); ++j) {
      if (KS == 0 && j >= KW) break;
      const float wv = KS ? wr[KS ? i * KS + j : 0] : decode8<WDT>(wp[i * KW + j]);
      const int iw0 = iwb + j * a.dw;
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        const int iw = iw0 + p * a.sw;
        const float xv = (iw >= 0 && iw < a.W) ? E::to_f32(row[iw]) : 0.f;
        acc[p] = __builtin_fmaf(xv, wv, acc[p]);
      }
    }
  };
  if constexpr (KS != 0) {
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int ih = ih0 + i * a.dh;
      if (ih >= 0 && ih < a.H) row_taps(i, xp + (size_t)ih * a.W);
    }
  } else {
    for (int i = 0; i < KH; ++i) {
      const int ih = ih0 + i * a.dh;
      if (ih >= 0 && ih < a.H) row_taps(i, xp + (size_t)ih * a.W);
    }
  }
  const float sc = E::to_f32(reinterpret_cast<const T*>(a.scale)[oc]);
  const bool has_bias = a.bias != nullptr;
  const float bv = has_bias ? E::to_f32(reinterpret_cast<const T*>(a.bias)[oc]) : 0.f;
  T out[PX];
#pragma unroll
  for (int p = 0; p < PX; ++p) {
    float v = acc[p] * sc;
    asm volatile("" : "+v"(v));  // the product is rounded to fp32 before anything else happens to it (no fused multiply-add with the bias)
    if (has_bias) v = E::to_f32(E::from_f32(v)) + bv;
    out[p] = E::from_f32(v);
  }
  T* yp = reinterpret_cast<T*>(a.y) + ((size_t)plane * a.OH + oh) * a.OW + ow0;
  if (ow0 + PX <= a.OW && (reinterpret_cast<uintptr_t>(yp) & 7) == 0) {
    *reinterpret_cast<uint2*>(yp) = *reinterpret_cast<const uint2*>(out);
  } else {
#pragma unroll
    for (int p = 0; p < PX; ++p)
      if (ow0 + p < a.OW) yp[p] = out[p];
  }
}

// ---- r6: strip form for the large feature maps (W % 8 == 0: 224 / 112 / 56 ...), 3 x 3 and 5 x 5 windows, stride 1 or 2, dilation 1 --------------------
// The quad kernel above issues one 2-byte load per (tap column, output): 18 load instructions per four outputs at stride 1 - it is bound by load ISSUE
// ((8,144,56,56) 19.7 us for 14.4 MB: 0.09 of the HBM rate).  Here a thread owns 8 neighbouring output columns of RO consecutive output rows and reads every input
// row it needs as whole 16-byte chunks: the chunk under its columns and the neighbours on either side (L1 hits: they are the main chunk of the adjacent thread) -
// 18 loads of 16 bytes per 32 outputs instead of 144 of 2 bytes, and every input row serves up to KS output rows from registers.  Same arithmetic contract: fp32
// products accumulated in (i, j) order, scale on the sum, one rounding to the output dtype, the reference's bias order.  The window offset inside the loaded
// chunks depends on the padding: PW is a template parameter so that every element index is a compile-time register index.
template <int DT, int KS, int SW, int PW, int RO, int CE>
__global__ void __launch_bounds__(THREADS) qconv2d_depthwise_strip_kernel(const Args a, int wdt, int strips, int rgroups) {
  using E = Elem<DT>;
  using T = typename E::T;
  // CE: elements per chunk = output columns per thread - 8 (16-byte chunks, W % 8 == 0) or 4 (8-byte chunks, W % 4 == 0: the 28 x 28 maps)
  constexpr int CO = CE;                           // output columns per thread
  constexpr int NCH = SW == 1 ? 3 : 4;             // chunks per input row: local element e is input column CE * (q * SW - 1) + e
  constexpr int NR = (RO - 1) * SW + KS;           // input rows a thread reads (stride is the same in both directions)
  constexpr int OFF = CE - PW;                     // local element of (output column 0, tap column 0)
  static_assert(CE == 8 || CE == 4, "chunks of 16 or 8 bytes");
  static_assert(OFF >= 0 && (CO - 1) * SW + KS - 1 + OFF < CE * NCH, "window outside the loaded chunks");
  using Chunk = std::conditional_t<CE == 8, uint4, uint2>;
  const int64_t item = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  const int per_plane = rgroups * strips;
  const int plane = (int)(item / per_plane);      // b * OC + oc
  if (plane >= a.B * a.OC) return;
  const int rem = (int)(item - (int64_t)plane * per_plane);
  const int oc = plane % a.OC, b = plane / a.OC;
  const int c = oc / a.mult;
  const int rg = rem / strips, q = rem - rg * strips;
  const int oh0 = rg * RO, ow0 = q * CO;
  const Chunk* xp = reinterpret_cast<const Chunk*>(reinterpret_cast<const T*>(a.x) + ((size_t)b * a.C + c) * a.H * a.W);  // W % CE == 0: rows are whole chunks
  const int wch = a.W / CE;                        // chunks per input row
  const int ih0 = oh0 * SW - a.ph;
  const int gc0 = q * SW - 1;

  // every load of the thread up front: NR x NCH independent requests
  Chunk raw[NR][NCH];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int ih = ih0 + r;
    const bool row_ok = ih >= 0 && ih < a.H;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int gc = gc0 + k;
      raw[r][k] = Chunk{};
      if (row_ok && gc >= 0 && gc < wch) raw[r][k] = xp[(size_t)ih * wch + gc];
    }
  }
  const uint8_t* wp = a.w + (size_t)oc * (KS * KS);
  float wr[KS * KS];
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) {
    const uint8_t wb = wp[t];
    wr[t] = wdt == QUANTO_HIP_I8 ? decode8<QUANTO_HIP_I8>(wb) : wdt == QUANTO_HIP_F8_E4M3FN ? decode8<QUANTO_HIP_F8_E4M3FN>(wb) : decode8<QUANTO_HIP_F8_E5M2>(wb);
  }
  float acc[RO][CO];
#pragma unroll
  for (int ro = 0; ro < RO; ++ro)
#pragma unroll
    for (int p = 0; p < CO; ++p) acc[ro][p] = 0.f;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    // the elements of this row the windows touch, as fp32 (compile-time indices: the rest is never unpacked)
    float xv[CE * NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      uint32_t d[CE / 2];
      __builtin_memcpy(d, &raw[r][k], sizeof(d));
#pragma unroll
      for (int e = 0; e < CE; ++e) {
        const uint16_t h = (uint16_t)(d[e >> 1] >> ((e & 1) * 16));
        xv[k * CE + e] = E::to_f32(__builtin_bit_cast(T, h));
      }
    }
#pragma unroll
    for (int ro = 0; ro < RO; ++ro) {
      const int i = r - ro * SW;  // window row of output row ro that this input row is
      if (i < 0 || i >= KS) continue;
#pragma unroll
      for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int p = 0; p < CO; ++p) acc[ro][p] = __builtin_fmaf(xv[p * SW + j + OFF], wr[i * KS + j], acc[ro][p]);
    }
  }
  const float sc = E::to_f32(reinterpret_cast<const T*>(a.scale)[oc]);
  const bool has_bias = a.bias != nullptr;
  const float bv = has_bias ? E::to_f32(reinterpret_cast<const T*>(a.bias)[oc]) : 0.f;
#pragma unroll
  for (int ro = 0; ro < RO; ++ro) {
    const int oh = oh0 + ro;
    if (oh >= a.OH) break;
    T out[CO];
#pragma unroll
    for (int p = 0; p < CO; ++p) {
      float v = acc[ro][p] * sc;
      asm volatile("" : "+v"(v));  // the product is rounded to fp32 before anything else happens to it (no fused multiply-add with the bias)
      if (has_bias) v = E::to_f32(E::from_f32(v)) + bv;
      out[p] = E::from_f32(v);
    }
    T* yp = reinterpret_cast<T*>(a.y) + ((size_t)plane * a.OH + oh) * a.OW + ow0;
    const uintptr_t ya = reinterpret_cast<uintptr_t>(yp);
    if (CO == 8 && ow0 + CO <= a.OW && (ya & 15) == 0) {
      *reinterpret_cast<uint4*>(yp) = *reinterpret_cast<const uint4*>(out);
    } else if (ow0 + CO <= a.OW && (ya & 7) == 0) {
#pragma unroll
      for (int h = 0; h < CO / 4; ++h) reinterpret_cast<uint2*>(yp)[h] = reinterpret_cast<const uint2*>(out)[h];
    } else {
#pragma unroll
      for (int p = 0; p < CO; ++p)
        if (ow0 + p < a.OW) yp[p] = out[p];
    }
  }
}

// the strip form's shapes: square 3 x 3 / 5 x 5 windows, equal strides of 1 or 2, no dilation, W a multiple of 4 (rows of whole 16- or 8-byte chunks, every
// plane then starts on a chunk as well), "same" padding (3 x 3 also without padding)
static bool strip_eligible(const Args& a) {
  if (a.KH != a.KW || (a.KH != 3 && a.KH != 5) || a.sh != a.sw || (a.sw != 1 && a.sw != 2) || a.dh != 1 || a.dw != 1) return false;
  if ((a.W & 3) != 0 || (reinterpret_cast<uintptr_t>(a.x) & 15) != 0) return false;
  const bool same = a.pw == a.KW / 2, none = a.pw == 0 && a.KW == 3;
  return (same || none) && env_int("QUANTO_HIP_DW_STRIP", 1) != 0;
}

template <int DT, int KS, int SW, int PW, int RO, int CE>
static int launch_strip_ce(const Args& a, int wdt, hipStream_t stream) {
  const int strips = (a.OW + CE - 1) / CE, rgroups = (a.OH + RO - 1) / RO;
  const int64_t items = (int64_t)a.B * a.OC * rgroups * strips;
  hipLaunchKernelGGL((qconv2d_depthwise_strip_kernel<DT, KS, SW, PW, RO, CE>), dim3((unsigned)((items + THREADS - 1) / THREADS)), dim3(THREADS), 0, stream, a, wdt,
                     strips, rgroups);
  return launch_status();
}
template <int DT, int KS, int SW, int PW, int RO>
static int launch_strip(const Args& a, int wdt, hipStream_t stream) {
  return (a.W & 7) == 0 ? launch_strip_ce<DT, KS, SW, PW, RO, 8>(a, wdt, stream) : launch_strip_ce<DT, KS, SW, PW, RO, 4>(a, wdt, stream);
}

template <int DT>
static int launch_strip_dt(const Args& a, int wdt, hipStream_t stream) {
  if (a.KW == 5) return a.sw == 1 ? launch_strip<DT, 5, 1, 2, 4>(a, wdt, stream) : launch_strip<DT, 5, 2, 2, 2>(a, wdt, stream);
  if (a.pw == 1) return a.sw == 1 ? launch_strip<DT, 3, 1, 1, 4>(a, wdt, stream) : launch_strip<DT, 3, 2, 1, 2>(a, wdt, stream);
  return a.sw == 1 ? launch_strip<DT, 3, 1, 0, 4>(a, wdt, stream) : launch_strip<DT, 3, 2, 0, 2>(a, wdt, stream);
}

template <int DT, int WDT>
static int launch(const Args& a, hipStream_t stream) {
  const int64_t items = (int64_t)a.B * a.OC * a.OH * a.quads;
  const dim3 grid((unsigned)((items + THREADS - 1) / THREADS));
  const int ks = a.KH == a.KW ? a.KH : 0;
  if (ks == 3)
    hipLaunchKernelGGL((qconv2d_depthwise_kernel<DT, WDT, 3>), grid, dim3(THREADS), 0, stream, a);
  else if (ks == 5)
    hipLaunchKernelGGL((qconv2d_depthwise_kernel<DT, WDT, 5>), grid, dim3(THREADS), 0, stream, a);
  else if (ks == 7)
    hipLaunchKernelGGL((qconv2d_depthwise_kernel<DT, WDT, 7>), grid, dim3(THREADS), 0, stream, a);
  else
    hipLaunchKernelGGL((qconv2d_depthwise_kernel<DT, WDT, 0>), grid, dim3(THREADS), 0, stream, a);
  return launch_status();
}

}  // namespace dw

bool qbytes_conv2d_depthwise_supported(int64_t B, int64_t C, int64_t H, int64_t W, int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int sh, int sw,
                                       int ph, int pw, int dh, int dw, int a_dtype, int b_dtype, int out_dtype) {
  const bool wd = b_dtype == QUANTO_HIP_I8 || b_dtype == QUANTO_HIP_F8_E4M3FN || b_dtype == QUANTO_HIP_F8_E5M2;
  const bool ad = (a_dtype == QUANTO_HIP_BF16 || a_dtype == QUANTO_HIP_F16) && out_dtype == a_dtype;
  if (!wd || !ad || B < 1 || C < 1 || OC < C || OC % C != 0 || KH < 1 || KW < 1 || sh < 1 || sw < 1 || dh < 1 || dw < 1 || ph < 0 || pw < 0) return false;
  if (OH != (H + 2 * ph - dh * (KH - 1) - 1) / sh + 1 || OW != (W + 2 * pw - dw * (KW - 1) - 1) / sw + 1 || OH < 1 || OW < 1) return false;
  return B * C * H * W < (1ll << 31) && B * OC * OH * OW < (1ll << 31) && KH * KW <= 4096 && H < (1 << 20) && W < (1 << 20);
}

int qbytes_conv2d_depthwise(const void* x, const void* w, const void* scales, const void* bias, void* y, int64_t B, int64_t C, int64_t H, int64_t W,
                            int64_t OC, int64_t KH, int64_t KW, int64_t OH, int64_t OW, int sh, int sw, int ph, int pw, int dh, int dw, int a_dtype,
                            int b_dtype, int out_dtype, hipStream_t stream) {
  if (!qbytes_conv2d_depthwise_supported(B, C, H, W, OC, KH, KW, OH, OW, sh, sw, ph, pw, dh, dw, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  const dw::Args a{x, reinterpret_cast<const uint8_t*>(w), scales, bias, y, (int)B, (int)C, (int)H, (int)W, (int)OC, (int)(OC / C), (int)KH, (int)KW, (int)OH,
                   (int)OW, sh, sw, ph, pw, dh, dw, (int)((OW + dw::PX - 1) / dw::PX)};
#define QH_DW(DT)                                                                                       \
  if (b_dtype == QUANTO_HIP_I8) return dw::launch<DT, QUANTO_HIP_I8>(a, stream);                         \
  if (b_dtype == QUANTO_HIP_F8_E4M3FN) return dw::launch<DT, QUANTO_HIP_F8_E4M3FN>(a, stream);           \
  return dw::launch<DT, QUANTO_HIP_F8_E5M2>(a, stream)
  if (dw::strip_eligible(a)) {
    set_last_kernel("conv2d_depthwise_strip");
    return a_dtype == QUANTO_HIP_BF16 ? dw::launch_strip_dt<QUANTO_HIP_BF16>(a, b_dtype, stream) : dw::launch_strip_dt<QUANTO_HIP_F16>(a, b_dtype, stream);
  }
  set_last_kernel("conv2d_depthwise");
  if (a_dtype == QUANTO_HIP_BF16) { QH_DW(QUANTO_HIP_BF16); }
  QH_DW(QUANTO_HIP_F16);
#undef QH_DW
}

}  // namespace qh
