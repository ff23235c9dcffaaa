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
  if (a_dtype == QUANTO_HIP_BF16) { QH_DW(QUANTO_HIP_BF16); }
  QH_DW(QUANTO_HIP_F16);
#undef QH_DW
}

}  // namespace qh
