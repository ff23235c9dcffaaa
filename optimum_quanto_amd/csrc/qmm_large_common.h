// Shared pieces of the large-tile MFMA GEMMs (qmm_mfma_large.hip: 16x16x32 MFMA; the 32x32x16 experiment of r2 is kept as scripts/probes/qmm_mfma_large32.hip).
#pragma once
#include "qh_common.h"

namespace qh {
namespace lt {

constexpr int BK = 64;
constexpr int STAGES = 3;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// LDS-DMA, 16 bytes per lane, wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset (inline asm: through the builtin hipcc puts a vmcnt(0) in front of every ds_read that follows).
// M0 is written and not restored: on gfx9+ the compiler only needs M0 for constructs this kernel does not contain
// (movrel, GWS, sendmsg, its own LDS-DMA builtins), and two SALU instructions per piece matter in a one-wave-per-SIMD
// instruction stream where every issue slot next to an MFMA is accounted for.
#ifndef QH_GLDS_POLICY
#define QH_GLDS_POLICY ""  // cache policy bits of the operand DMA (probes: " sc1", " nt", " sc0 sc1": profiles/r06_glds_cache_policy_ab.jsonl)
#endif
__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1" QH_GLDS_POLICY
      :
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

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
};
template <>
struct Mma<QUANTO_HIP_F16> {
  using V8 = f16x8;
  static __device__ __forceinline__ f32x4 run(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));  // exact for int8 / fp8 values
  }
};

enum { W_I8 = 0, W_F8E4M3 = 1, W_F8E5M2 = 2, W_DENSE = 3, W_F8E4M3FNUZ = 4 };  // W_DENSE: weights already in the activation dtype (weights-direct loop only)

__device__ __forceinline__ int swz_a(int row) {
  const int q = (row + 4) & 15;
  return ((((q >> 3) ^ 1) << 2) | ((q >> 1) & 3));
}
__device__ __forceinline__ int swz_w(int row) { return (-(row >> 2)) & 3; }

// bytes (2p, 2p+1) of `word` -> two 16-bit elements.  int8: 3 VALU ops (2 x v_cvt_f32_i32 with SDWA byte select + one packed
// conversion).  fp8 / bf8: ONE op - gfx950's v_cvt_scalef32_pk_{bf16,f16}_{fp8,bf8} converts a pair straight to the 16-bit type
// (scale 1.0: exact, every e4m3 / e5m2 value is representable in bf16 and fp16) instead of cvt_pk_f32_fp8 + a packed narrowing.
template <int DT, int FMT>
__device__ __forceinline__ uint32_t convert_pair(uint32_t word, int p) {
  if constexpr (FMT == W_I8) {
    const float f0 = p == 0 ? (float)(int8_t)(word & 0xFFu) : (float)(int8_t)((word >> 16) & 0xFFu);
    const float f1 = p == 0 ? (float)(int8_t)((word >> 8) & 0xFFu) : (float)(int8_t)(word >> 24);
    return Mma<DT>::pack(f0, f1);
  } else if constexpr (FMT == W_F8E4M3FNUZ) {
    return DT == QUANTO_HIP_BF16 ? fnuz_pair_bf16(word, p) : fnuz_pair_f16(word, p);  // qh_common.h: fn / 2 + three patched patterns
  } else if constexpr (FMT == W_F8E4M3) {
    if constexpr (DT == QUANTO_HIP_BF16)
      return __builtin_bit_cast(uint32_t, p == 0 ? __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)word, 1.0f, false)
                                                 : __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)word, 1.0f, true));
    else
      return __builtin_bit_cast(uint32_t, p == 0 ? __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)word, 1.0f, false)
                                                 : __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)word, 1.0f, true));
  } else {
    if constexpr (DT == QUANTO_HIP_BF16)
      return __builtin_bit_cast(uint32_t, p == 0 ? __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8((int)word, 1.0f, false)
                                                 : __builtin_amdgcn_cvt_scalef32_pk_bf16_bf8((int)word, 1.0f, true));
    else
      return __builtin_bit_cast(uint32_t, p == 0 ? __builtin_amdgcn_cvt_scalef32_pk_f16_bf8((int)word, 1.0f, false)
                                                 : __builtin_amdgcn_cvt_scalef32_pk_f16_bf8((int)word, 1.0f, true));
  }
}

struct Args {
  const void* x;
  const uint8_t* w;
  const void* scale;
  const void* bias;
  void* y;
  int M, N, K;
  int group_m;  // tile raster: groups of group_m tile rows, column-major inside a group (see tile_coords)
  // split-K (S > 1): workgroup b computes K-range b % S of tile b / S; fp32 partial sums go to `partials` and the last
  // workgroup of a tile to arrive adds them in split order (same protocol and workspace contract as qbits_skinny.hip)
  int S;
  int* counters;    // [tiles], zero on entry, zero on exit
  float* partials;  // [tiles * S][NJ * MI][threads] float4
};

// XCD-aware tile order.  Consecutive workgroup ids land on different XCDs (id % 8), so first give every XCD a contiguous
// band of tile indices; inside the index space walk groups of `group_m` tile rows column by column, so that a band of
// B = tiles/8 consecutive indices is a (group_m x B/group_m) rectangle: its activation panels (group_m) and weight panels
// (B/group_m) are what that XCD's L2 has to fetch.  group_m ~ sqrt(B * bytes_per_weight_row / bytes_per_activation_row)
// minimises the fetched bytes (cfg4, 128-tiles: 294 MB of fabric traffic per launch with row-major order).
__device__ __forceinline__ void tile_coords(int bid, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int per_group = group_m * tiles_n;
  const int g = t / per_group, in_g = t - g * per_group;
  const int rows = tiles_m - g * group_m < group_m ? tiles_m - g * group_m : group_m;
  tn = in_g / rows;
  tm = g * group_m + (in_g - tn * rows);
}

}  // namespace lt
}  // namespace qh
