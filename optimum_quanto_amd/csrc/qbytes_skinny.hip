// qbytes_mm for small batches (2 < M <= 256 in passes of 64, e.g. batched decode with int8 / fp8 weights): weight-streaming MFMA kernel.
//
// The 8-bit sibling of qbits_skinny.hip - HBM-bound like the GEMV, products on the matrix cores so that the cost per weight
// byte does not grow with M:
//   * a block of 4 waves owns 64 output features (16 weight rows per wave) and streams them over its K-range in tiles of 128
//     k through a 4..8-stage LDS-DMA ring (`global_load_lds_dwordx4`, counted vmcnt, one s_barrier per pair of tiles);
//   * lane (i = lane & 15, g = lane >> 4) of a wave fetches the 32 bytes k = 32g .. 32g+31 of row i with two ds_read_b128
//     and converts them in registers (int8: 2 x v_cvt_f32_i32 SDWA + v_cvt_pk_bf16_f32 per pair; fp8: v_cvt_pk_f32_fp8 +
//     pack) into the A operands of the tile's four k-steps: step t uses bytes 8t .. 8t+7 of the lane's 32, i.e.
//     k = 32g + 8t .. +7.  The activation fragment of step t is read with the same k assignment (16-byte chunk 4g + t of
//     the token's 256-byte row) - an MFMA is invariant under a k permutation applied to both operands;
//   * the per-channel scale (and bias) is applied once, to the fp32 accumulator, in the epilogue - no per-group work;
//   * K is split across workgroups when N alone cannot occupy the chip, and M > 64 runs in passes of 64 rows: same scheme,
//     same workspace contract (zeroed arrival counters, system-coherent partial sums) as qbits_skinny.hip.
#include <cstdlib>

#include "qh_common.h"

namespace qh {
namespace skinny8 {

constexpr int BK = 128;  // k per tile = bytes per weight row and tile

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  asm volatile(  // M0 is written and not restored (qmm_large_common.h: nothing else in this kernel needs it)
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off"
      :
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// non-temporal flavour for the weight stream: every weight byte is read once per pass (MI355X_MICROARCH.md "nt-weights":
// issued -> landed 18 % sooner on one-shot streams)
__device__ __forceinline__ void glds16_nt(const void* gsrc, uint32_t lds_dst) {
  asm volatile(  // M0 is written and not restored (qmm_large_common.h: nothing else in this kernel needs it)
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off nt"
      :
      : "v"(gsrc), "s"(lds_dst)
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
  static __device__ __forceinline__ uint32_t pack(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b)); }
};

enum { W_I8 = 0, W_F8E4M3 = 1, W_F8E5M2 = 2, W_F8E4M3FNUZ = 3 };

// bytes (2p, 2p+1) of `word` -> two 16-bit elements (exact: every int8 / fp8 value is representable in bf16 and fp16);
// fp8 / bf8 in one op with gfx950's v_cvt_scalef32_pk_{bf16,f16}_{fp8,bf8} at scale 1.0 (see qmm_large_common.h)
template <int DT, int FMT>
__device__ __forceinline__ uint32_t convert_pair(uint32_t word, int p) {
  if constexpr (FMT == W_I8) {
    const float f0 = p == 0 ? (float)(int8_t)(word & 0xFFu) : (float)(int8_t)((word >> 16) & 0xFFu);
    const float f1 = p == 0 ? (float)(int8_t)((word >> 8) & 0xFFu) : (float)(int8_t)(word >> 24);
    return Mma<DT>::pack(f0, f1);
  } else if constexpr (FMT == W_F8E4M3FNUZ) {
    return DT == QUANTO_HIP_BF16 ? fnuz_pair_bf16(word, p) : fnuz_pair_f16(word, p);  // qh_common.h
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

template <int MAXN, int PER>
__device__ __forceinline__ void wait_vmcnt(int younger_tiles) {
  if constexpr (MAXN > 0) {
    if (younger_tiles * PER >= MAXN) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXN) : "memory");
      return;
    }
    wait_vmcnt<MAXN - PER, PER>(younger_tiles);
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

struct Args {
  const void* x;      // [M, K] activations (M <= 64 per launch)
  const uint8_t* w;   // [N, K] one byte per weight
  const void* scale;  // [N]
  const void* bias;   // [N] or null
  void* y;            // [M, N]
  int M, N, K;
  int S;              // K split (see qbits_skinny.hip)
  int* counters;
  float* partials;
  int nt;             // non-temporal weight DMA (single-pass calls)
};

// Several Linears sharing their input in one launch (MULTI, see qbits_skinny.hip): the feature blocks of all segments form
// one grid, a block takes weight / scale / bias / output pointers and N from its segment.
constexpr int MAX_SEGS = QUANTO_HIP_MAX_MULTI;
struct Segs {
  const uint8_t* w[MAX_SEGS];
  const void* scale[MAX_SEGS];
  const void* bias[MAX_SEGS];
  void* y[MAX_SEGS];
  int N[MAX_SEGS];
  int first_fb[MAX_SEGS];  // first feature block of each segment (INT_MAX for unused slots)
};

template <int DT, int FMT, int TF, int STAGES, bool MULTI = false>
__global__ void __launch_bounds__(256) qbytes_skinny_kernel(Args a, const Segs segs) {
  using E = Elem<DT>;
  using T = typename E::T;
  using V8 = typename Mma<DT>::V8;
  constexpr int WAVES = 4;
  constexpr int ROWS = 16 * WAVES;    // weight rows (= output features) per block
  constexpr int W_BYTES = ROWS * BK;  // 8 KiB: 2 KiB per wave
  constexpr int X_BYTES = TF * 16 * BK * 2;
  constexpr int STAGE_BYTES = W_BYTES + X_BYTES;
  constexpr int XP = TF * 4 / WAVES;  // 1 KiB activation DMA pieces per wave and tile (TF = 1: waves 0..TF*4-1 only)
  constexpr int XPI = XP > 0 ? XP : 1;
  constexpr int PER = 2 + XPI;        // DMA instructions per wave and tile (upper bound used for vmcnt accounting)
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = a.S;
  const int fbg = S > 1 ? blockIdx.x / S : blockIdx.x, sp = S > 1 ? blockIdx.x - fbg * S : 0;  // global feature block
  int fb = fbg;
  if constexpr (MULTI) {
    int seg = 0;
#pragma unroll
    for (int i = 1; i < MAX_SEGS; ++i) seg += fbg >= segs.first_fb[i];
    fb = fbg - segs.first_fb[seg];
    a.w = segs.w[seg];
    a.scale = segs.scale[seg];
    a.bias = segs.bias[seg];
    a.y = segs.y[seg];
    a.N = segs.N[seg];
  }
  const int M = a.M, N = a.N, K = a.K;
  const int n_blk = fb * ROWS;
  const int nk = K / BK / S;
  const int kt0 = sp * nk;

  // ---- per-lane DMA sources ---------------------------------------------------------------------------------------------
  // weights: the wave's 16 rows x 128 B as two 1 KiB pieces of 8 rows; lane -> row lane>>3, position lane&7 holds chunk pos ^ (row & 7)
  const uint8_t* wsrc[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = wave * 16 + h * 8 + (lane >> 3), c = (lane & 7) ^ (r & 7);
    const int n = n_blk + r < N ? n_blk + r : N - 1;
    wsrc[h] = a.w + (size_t)n * K + c * 16 + (size_t)kt0 * BK;
  }
  // activations: 1 KiB pieces of 4 token rows (256 B each); piece u' covers tile rows 4u' .. 4u'+3
  const uint8_t* xsrc[XPI];
  constexpr int XPIECES = TF * 4;
#pragma unroll
  for (int u = 0; u < XPI; ++u) {
    const int piece = XP > 0 ? wave * XP + u : wave;  // TF = 1 with 4 waves: one piece per wave
    const int row = 4 * piece + (lane >> 4);
    const int c = (lane & 15) ^ (row & 15);
    const int m = row < M ? row : M - 1;
    xsrc[u] = reinterpret_cast<const uint8_t*>(reinterpret_cast<const T*>(a.x) + (size_t)m * K + c * 8 + (size_t)kt0 * BK);
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  auto issue = [&](int kt, int stage) {
    const uint32_t st = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE_BYTES);
    if (a.nt) {
      glds16_nt(wsrc[0] + (size_t)kt * BK, st + (wave * 2 + 0) * 1024);
      glds16_nt(wsrc[1] + (size_t)kt * BK, st + (wave * 2 + 1) * 1024);
    } else {
      glds16(wsrc[0] + (size_t)kt * BK, st + (wave * 2 + 0) * 1024);
      glds16(wsrc[1] + (size_t)kt * BK, st + (wave * 2 + 1) * 1024);
    }
#pragma unroll
    for (int u = 0; u < XPI; ++u) {
      const int piece = XP > 0 ? wave * XP + u : wave;
      if (piece < XPIECES) glds16(xsrc[u] + (size_t)kt * (BK * 2), st + W_BYTES + piece * 1024);
    }
  };
  // r6: the lane's four scale / bias values are requested HERE, in front of the first DMA (the oldest entries of the in-order vector-memory queue), not
  // after the K loop: the epilogue used to open with a global round trip (~1 us of a 7-10 us call).  As asm: hipcc would drain the DMA queue at a load it sees.
  uint32_t sc_raw[4], bv_raw[4];
  {
    const int nq = n_blk + wave * 16 + 4 * (lane >> 4);
    // no bias: the same loads from the scale vector (unused) - a branch-free sequence, so that nothing but these asm statements ever writes the
    // destination registers before the counted wait (tests/test_build_invariants.py reads the listing for exactly that)
    const T* bsrc = reinterpret_cast<const T*>(a.bias != nullptr ? a.bias : a.scale);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nq + r < N ? nq + r : N - 1;
      asm volatile("global_load_ushort %0, %1, off" : "=v"(sc_raw[r]) : "v"(reinterpret_cast<const T*>(a.scale) + n) : "memory");
      asm volatile("global_load_ushort %0, %1, off" : "=v"(bv_raw[r]) : "v"(bsrc + n) : "memory");
    }
  }
#pragma unroll
  for (int t = 0; t < STAGES - 2; ++t)
    if (t < nk) issue(t, t);

  // ---- fragment read offsets ------------------------------------------------------------------------------------------------
  const int fi = lane & 15, fg = lane >> 4;
  const int wrow = wave * 16 + fi;
  int woff[2];  // 16-byte chunks 2g and 2g+1 of the lane's row
#pragma unroll
  for (int h = 0; h < 2; ++h) woff[h] = wrow * 128 + (((2 * fg + h) ^ (wrow & 7)) << 4);
  int xoff[TF][4];  // k-step t: chunk 4g + t of the token's row
#pragma unroll
  for (int tf = 0; tf < TF; ++tf) {
    const int row = tf * 16 + fi;
#pragma unroll
    for (int t = 0; t < 4; ++t) xoff[tf][t] = W_BYTES + row * 256 + (((4 * fg + t) ^ (row & 15)) << 4);
  }

  f32x4 acc[TF];
#pragma unroll
  for (int tf = 0; tf < TF; ++tf) acc[tf] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute_tile = [&](const uint8_t* st) {
    uint4 wr[2];
    wr[0] = *reinterpret_cast<const uint4*>(st + woff[0]);
    wr[1] = *reinterpret_cast<const uint4*>(st + woff[1]);
    // every activation fragment of the tile up front, pinned by the scheduling barrier (r5, as qbits_skinny.hip: hipcc otherwise reads one
    // fragment, waits, issues its MFMAs, TF x 4 times per tile - one exposed LDS latency each)
    V8 xb[TF][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int tf = 0; tf < TF; ++tf) xb[tf][t] = *reinterpret_cast<const V8*>(st + xoff[tf][t]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // k-step t: bytes 8t .. 8t+7 of the lane's 32 = dwords (2t, 2t+1) -> four operand dwords
      const uint32_t d0 = (t & 1) ? wr[t >> 1].z : wr[t >> 1].x, d1 = (t & 1) ? wr[t >> 1].w : wr[t >> 1].y;
      uint32_t op[4];
      op[0] = convert_pair<DT, FMT>(d0, 0);
      op[1] = convert_pair<DT, FMT>(d0, 1);
      op[2] = convert_pair<DT, FMT>(d1, 0);
      op[3] = convert_pair<DT, FMT>(d1, 1);
      const V8 wa = __builtin_bit_cast(V8, make_uint4(op[0], op[1], op[2], op[3]));
#pragma unroll
      for (int tf = 0; tf < TF; ++tf) acc[tf] = Mma<DT>::run(wa, xb[tf][t], acc[tf]);
    }
  };

  // Tiles are consumed in pairs per barrier; ring of STAGES (even) stages, tiles kt .. kt+STAGES-1 in flight
  int cur = 0;
  for (int kt = 0; kt < nk; kt += 2) {
    const bool pair = kt + 1 < nk;
    const int last = pair ? kt + 1 : kt;
    const int younger = nk - 1 - last < STAGES - 4 ? nk - 1 - last : STAGES - 4;
    wait_vmcnt<(STAGES - 4) * PER, PER>(younger);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int nxt = cur + 1 == STAGES ? 0 : cur + 1;
    {
      const int s0 = cur >= 2 ? cur - 2 : cur + STAGES - 2, s1 = s0 + 1 == STAGES ? 0 : s0 + 1;
      if (kt + STAGES - 2 < nk) issue(kt + STAGES - 2, s0);
      if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, s1);
    }
    compute_tile(smem + cur * STAGE_BYTES);
    if (pair) compute_tile(smem + nxt * STAGE_BYTES);
    cur = nxt + 1 == STAGES ? 0 : nxt + 1;
  }

  // ---- split-K reduction (see qbits_skinny.hip for the coherence argument) -----------------------------------------------------
  if (S > 1) {
    float* mine = a.partials + ((size_t)blockIdx.x * TF * 256 + tid) * 4;  // fragment-major: whole lines per store instruction
#pragma unroll
    for (int tf = 0; tf < TF; ++tf)
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(mine + tf * (256 * 4)), "v"(acc[tf]) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) *flag = __hip_atomic_fetch_add(a.counters + fbg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (*flag != S - 1) return;
    if (tid == 0) __hip_atomic_store(a.counters + fbg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) acc[tf] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < S; ++q) {
      const float* theirs = a.partials + ((size_t)(fbg * S + q) * TF * 256 + tid) * 4;
      f32x4 v[TF];
#pragma unroll
      for (int tf = 0; tf < TF; ++tf) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[tf]) : "v"(theirs + tf * (256 * 4)) : "memory");
#pragma unroll
      for (int tf = 0; tf < TF; ++tf) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[tf])::"memory");
#pragma unroll
      for (int tf = 0; tf < TF; ++tf)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tf][r] += v[tf][r];
    }
  }

  // ---- epilogue: per-channel scale on the accumulator, optional bias; a lane holds 4 consecutive features of one token ----------
  T* yg = reinterpret_cast<T*>(a.y);
  const int n0 = n_blk + wave * 16 + 4 * fg;
  float sc[4], bv[4];
  const bool has_bias = a.bias != nullptr;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // long since complete (requested ahead of the first tile)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    asm volatile("" : "+v"(sc_raw[r]), "+v"(bv_raw[r]));
    sc[r] = E::to_f32(__builtin_bit_cast(T, (uint16_t)sc_raw[r]));
    bv[r] = has_bias ? E::to_f32(__builtin_bit_cast(T, (uint16_t)bv_raw[r])) : 0.f;
  }
#pragma unroll
  for (int tf = 0; tf < TF; ++tf) {
    const int m = tf * 16 + fi;
    if (m < M) {
      T out[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[tf][r] * sc[r];
        asm volatile("" : "+v"(v));  // product rounded to fp32 first, with and without bias (no single-rounding v_fma_mixlo_f16)
        if (has_bias) v = E::to_f32(E::from_f32(v)) + bv[r];
        out[r] = E::from_f32(v);
      }
      if (n0 + 3 < N && (N & 3) == 0) {
        *reinterpret_cast<uint2*>(yg + (size_t)m * N + n0) = *reinterpret_cast<const uint2*>(out);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n0 + r < N) yg[(size_t)m * N + n0 + r] = out[r];
      }
    }
  }
}

constexpr int lds_bytes(int tf, int stages) { return stages * (64 * BK + tf * 16 * BK * 2); }

template <int DT, int FMT, int TF, int STAGES>
static int launch_s(const Args& a, hipStream_t stream, const Segs* segs = nullptr, int total_fb = 0) {
  constexpr int lds = lds_bytes(TF, STAGES);
  if (segs) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbytes_skinny_kernel<DT, FMT, TF, STAGES, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((qbytes_skinny_kernel<DT, FMT, TF, STAGES, true>), dim3(total_fb * a.S), dim3(256), lds, stream, a, *segs);
    return launch_status();
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbytes_skinny_kernel<DT, FMT, TF, STAGES>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((qbytes_skinny_kernel<DT, FMT, TF, STAGES>), dim3((a.N + 63) / 64 * a.S), dim3(256), lds, stream, a, Segs{});
  return launch_status();
}

template <int DT, int FMT>
static int launch_tf(const Args& a, hipStream_t stream, const Segs* segs = nullptr, int total_fb = 0) {
  if (segs) {  // the multi-Linear launch keeps the default ring
    if (a.M <= 16) return launch_s<DT, FMT, 1, 4>(a, stream, segs, total_fb);
    if (a.M <= 32) return launch_s<DT, FMT, 2, 4>(a, stream, segs, total_fb);
    return launch_s<DT, FMT, 4, 4>(a, stream, segs, total_fb);
  }
  // ring depth: 4 stages (48 / 64 KiB: two to three blocks per CU, which hide each other's barrier and reduction stalls - see the
  // measurements in qbits_skinny.hip) unless the experiment knob asks for the deep ring (8 stages, one block per CU)
  const bool deep = env_int("QUANTO_HIP_SKINNY_LDS_KB", 50) >= 100;
  if (a.M <= 16) return deep ? launch_s<DT, FMT, 1, 8>(a, stream) : launch_s<DT, FMT, 1, 4>(a, stream);
  if (a.M <= 32) return deep ? launch_s<DT, FMT, 2, 8>(a, stream) : launch_s<DT, FMT, 2, 4>(a, stream);
  return deep ? launch_s<DT, FMT, 4, 6>(a, stream) : launch_s<DT, FMT, 4, 4>(a, stream);
}

}  // namespace skinny8

static int skinny8_split(int64_t N, int64_t K) {
  const int forced = env_int("QUANTO_HIP_SKINNY_SPLIT", 0);  // experiments
  const int blocks = (int)((N + 63) / 64), G = (int)(K / skinny8::BK);
  int s = 1;  // same rule as qbits_skinny.hip: 250-500 blocks, at least 8 tiles per block
  while (s < 8 && blocks * s * 2 <= 512 && G % (s * 2) == 0 && G / (s * 2) >= 8) s *= 2;
  if (forced > 0 && G % forced == 0) s = forced;
  if ((size_t)((N + 63) / 64) * 4 > QUANTO_HIP_WS_COUNTER_BYTES) s = 1;  // one counter per feature block of 64
  return s;
}
// fixed-size counter region shared by all split-K kernels of the library (see qbits_skinny.hip)
static size_t skinny8_counter_bytes(int64_t) { return QUANTO_HIP_WS_COUNTER_BYTES; }

bool qbytes_skinny_supported(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  const bool bd = b_dtype == QUANTO_HIP_I8 || b_dtype == QUANTO_HIP_F8_E4M3FN || b_dtype == QUANTO_HIP_F8_E5M2 || b_dtype == QUANTO_HIP_F8_E4M3FNUZ;
  return bd && a_dtype == out_dtype && (out_dtype == QUANTO_HIP_BF16 || out_dtype == QUANTO_HIP_F16) && K % skinny8::BK == 0 && M >= 1 &&
         M <= QUANTO_HIP_SKINNY_MAX_M && N >= 1 && N < (1 << 30) && K < (1 << 30);
}

// [counters (zero on entry, zero on exit) | fp32 partial sums]; 0 when the problem is not split
size_t qbytes_skinny_workspace(int64_t M, int64_t N, int64_t K) {
  const int S = skinny8_split(N, K);
  if (S == 1) return 0;
  const int tf = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
  return skinny8_counter_bytes(N) + (size_t)((N + 63) / 64) * S * 256 * tf * 16;
}

int qbytes_mm_skinny(const void* x, const void* w, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int a_dtype,
                     int b_dtype, int out_dtype, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!qbytes_skinny_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16) return QUANTO_HIP_EALIGN;
  int S = skinny8_split(N, K);
  if (S > 1 && (!workspace || workspace_bytes < qbytes_skinny_workspace(M, N, K) || reinterpret_cast<uintptr_t>(workspace) % 16)) S = 1;
  for (int64_t m0 = 0; m0 < M; m0 += 64) {
    const int64_t rows = M - m0 < 64 ? M - m0 : 64;
    skinny8::Args a{reinterpret_cast<const uint8_t*>(x) + (size_t)m0 * K * 2, reinterpret_cast<const uint8_t*>(w), s, bias,
                    reinterpret_cast<uint8_t*>(y) + (size_t)m0 * N * 2, (int)rows, (int)N, (int)K, S, reinterpret_cast<int*>(workspace),
                    S > 1 ? reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(workspace) + skinny8_counter_bytes(N)) : nullptr,
                    env_int("QUANTO_HIP_SKINNY_NT", M <= 64 ? 1 : 0)};
    int r;
#define QH_FMT(DT)                                                                              \
  r = b_dtype == QUANTO_HIP_I8 ? skinny8::launch_tf<DT, skinny8::W_I8>(a, stream)               \
      : b_dtype == QUANTO_HIP_F8_E4M3FN ? skinny8::launch_tf<DT, skinny8::W_F8E4M3>(a, stream)  \
      : b_dtype == QUANTO_HIP_F8_E4M3FNUZ ? skinny8::launch_tf<DT, skinny8::W_F8E4M3FNUZ>(a, stream) \
                                        : skinny8::launch_tf<DT, skinny8::W_F8E5M2>(a, stream)
    if (out_dtype == QUANTO_HIP_BF16) {
      QH_FMT(QUANTO_HIP_BF16);
    } else {
      QH_FMT(QUANTO_HIP_F16);
    }
#undef QH_FMT
    if (r != QUANTO_HIP_OK) return r;
  }
  return QUANTO_HIP_OK;
}

// ---- several Linears with a shared input in one launch (3 <= M <= 64), see qbits_skinny.hip ------------------------------------
static int64_t multi_total(int nseg, const int64_t* N) {
  int64_t t = 0;
  for (int i = 0; i < nseg; ++i) t += N[i];
  return t;
}

bool qbytes_skinny_multi_supported(int nseg, const int64_t* N, int64_t M, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  if (nseg < 1 || nseg > skinny8::MAX_SEGS || M < 1 || M > 64) return false;
  for (int i = 0; i < nseg; ++i)
    if (N[i] <= 0 || N[i] % 64) return false;
  return qbytes_skinny_supported(M, multi_total(nseg, N), K, a_dtype, b_dtype, out_dtype);
}

size_t qbytes_skinny_multi_workspace(int nseg, const int64_t* N, int64_t M, int64_t K) { return qbytes_skinny_workspace(M, multi_total(nseg, N), K); }

int qbytes_mm_skinny_multi(const void* x, int nseg, const void* const* w, const void* const* s, const void* const* bias, void* const* y,
                           const int64_t* N, int64_t M, int64_t K, int a_dtype, int b_dtype, int out_dtype, void* workspace,
                           size_t workspace_bytes, hipStream_t stream) {
  if (!qbytes_skinny_multi_supported(nseg, N, M, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  const int64_t total = multi_total(nseg, N);
  uintptr_t align = reinterpret_cast<uintptr_t>(x);
  skinny8::Segs segs;
  int fb = 0;
  for (int i = 0; i < skinny8::MAX_SEGS; ++i) {
    const int j = i < nseg ? i : 0;  // unused slots repeat segment 0 and are never selected
    segs.w[i] = reinterpret_cast<const uint8_t*>(w[j]);
    segs.scale[i] = s[j];
    segs.bias[i] = bias ? bias[j] : nullptr;
    segs.y[i] = y[j];
    segs.N[i] = (int)N[j];
    segs.first_fb[i] = i < nseg ? fb : 0x7FFFFFFF;
    if (i < nseg) {
      fb += (int)(N[i] / 64);
      align |= reinterpret_cast<uintptr_t>(w[i]);
    }
  }
  if (align % 16) return QUANTO_HIP_EALIGN;
  int S = skinny8_split(total, K);
  if (S > 1 && (!workspace || workspace_bytes < qbytes_skinny_workspace(M, total, K) || reinterpret_cast<uintptr_t>(workspace) % 16)) S = 1;
  skinny8::Args a{x, segs.w[0], s[0], segs.bias[0], y[0], (int)M, (int)N[0], (int)K, S, reinterpret_cast<int*>(workspace),
                  S > 1 ? reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(workspace) + skinny8_counter_bytes(total)) : nullptr,
                  env_int("QUANTO_HIP_SKINNY_NT", 1)};
  int r;
#define QH_FMT(DT)                                                                                        \
  r = b_dtype == QUANTO_HIP_I8 ? skinny8::launch_tf<DT, skinny8::W_I8>(a, stream, &segs, fb)              \
      : b_dtype == QUANTO_HIP_F8_E4M3FN ? skinny8::launch_tf<DT, skinny8::W_F8E4M3>(a, stream, &segs, fb) \
      : b_dtype == QUANTO_HIP_F8_E4M3FNUZ ? skinny8::launch_tf<DT, skinny8::W_F8E4M3FNUZ>(a, stream, &segs, fb) \
                                        : skinny8::launch_tf<DT, skinny8::W_F8E5M2>(a, stream, &segs, fb)
  if (out_dtype == QUANTO_HIP_BF16) {
    QH_FMT(QUANTO_HIP_BF16);
  } else {
    QH_FMT(QUANTO_HIP_F16);
  }
#undef QH_FMT
  return r;
}

}  // namespace qh
