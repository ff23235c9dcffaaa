// qbytes_mm MFMA GEMM, v2: 256x256x64 tile, 8 waves, LDS-DMA staging, in-register weight conversion.
//
// y[M,N] = (x[M,K] @ q[N,K]^T) * scale[N]   with x bf16/fp16 and q int8 / fp8 (1 byte per weight).
//
// Why this shape (MI355X: 256 CUs, 160 KiB LDS, MFMA 16x16x32 = 16 cycles per SIMD):
//   * both operands are K-contiguous, so both are staged by `global_load_lds` (16 B per lane, no VGPR round trip, no
//     ds_write pass).  The weight tile stays in its 1-byte storage format in LDS: half the LDS footprint and half the
//     ds_read traffic of a bf16 image, and the stage is a pure DMA pipeline (3 stages of 48 KiB in flight, counted
//     vmcnt, one raw s_barrier per K-tile, loads always one to two tiles ahead, never drained inside the loop);
//   * each wave converts the weight fragments it is about to use in registers (int8: v_cvt_f32_i32 with SDWA byte
//     select + v_cvt_pk_bf16_f32; fp8: v_cvt_pk_f32_fp8 + pack) - 1.5 VALU ops per weight next to 64 MFMAs per
//     K-tile and wave, exact because every int8 / fp8 value is representable in bf16 and fp16;
//   * the LDS image is linear per DMA instruction (wave-uniform base + lane*16), so the bank-conflict swizzle is
//     applied to the per-lane GLOBAL source address and undone on the fragment read (both are the same involution):
//       activations: 128-byte rows, 16-byte chunk c of row r lives at chunk position c ^ (r & 7);
//       weights:      64-byte rows, chunk c of row r lives at position c ^ ((-(r >> 2)) & 3);
//   * a weight lane reads ONE ds_read_b128 per fragment and K-tile (bytes k = 16g .. 16g+15 of its row, g = lane>>4)
//     and uses the low half for the first MFMA k-step and the high half for the second; the activation fragments are
//     read with the matching k assignment (chunk 2g + kk) - MFMA results are invariant under a k permutation applied
//     to both operands.
#include "qh_common.h"

namespace qh {

namespace v2 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int A_BYTES = BM * BK * 2;   // 32 KiB
constexpr int W_BYTES = BN * BK;       // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
constexpr int STAGES = 3;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// LDS-DMA of 16 bytes per lane: LDS[m0 + lane*16 .. +16) <- *gsrc.  Issued through inline asm on purpose: hipcc
// (ROCm 7.2) tracks the builtin form as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front of the next ds_read,
// which drains the whole prefetch pipeline every K-tile.  The asm form is invisible to that bookkeeping, so the
// counted `s_waitcnt vmcnt(N)` + `s_barrier` below are the ONLY ordering between the DMA and the fragment reads.
// `lds_dst` must be wave-uniform (it is moved to M0; M0 is saved/restored because the compiler owns it).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
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
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));  // exact for int8 / fp8 values
  }
};

enum { W_I8 = 0, W_F8E4M3 = 1, W_F8E5M2 = 2 };

// LDS swizzles (chunk position = logical 16-byte chunk ^ swz(row)); derived for the ds_read_b128 service groups of
// gfx950 ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32) so that every fragment read is conflict free:
//  * activations (128-byte rows, 8 chunks, fragment lanes read chunk 2*(lane>>4)+kk): rows 4..11 of each 16-row
//    fragment take positions {0..3} ^ c, rows 0..3 and 12..15 take {4..7} ^ c, distinct inside each row parity;
//  * weights (64-byte rows, 4 chunks, lanes read chunk lane>>4).
__device__ __forceinline__ int swz_a(int row) {
  const int q = (row + 4) & 15;
  return ((((q >> 3) ^ 1) << 2) | ((q >> 1) & 3));
}
__device__ __forceinline__ int swz_w(int row) { return (-(row >> 2)) & 3; }

#ifdef QH_PHASE_TIMING
#define QH_STAGGER (a.variant != 2)
#define QH_VARIANT4 (a.variant == 4)
#define QH_PRIO_UP() do { if (a.variant == 0) __builtin_amdgcn_s_setprio(1); } while (0)
#define QH_PRIO_DOWN() do { if (a.variant == 0) __builtin_amdgcn_s_setprio(0); } while (0)
#define QH_STAMP(i) do { if (stamp_on) { asm volatile("s_waitcnt lgkmcnt(0)"); stamps[i] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define QH_STAGGER true
#define QH_VARIANT4 false
#define QH_PRIO_UP() __builtin_amdgcn_s_setprio(1)
#define QH_PRIO_DOWN() __builtin_amdgcn_s_setprio(0)
#define QH_STAMP(i) do { } while (0)
#endif

struct Args {
  const void* x;
  const uint8_t* w;
  const void* scale;
  const void* bias;
  void* y;
  int M, N, K;
#ifdef QH_PHASE_TIMING
  unsigned long long* dbg;
  int variant;  // priority experiment: 0 = flip prio around every MFMA phase, 1 = static prio 1 for waves 4-7, 2 = none
#endif
};

// One output dword of a converted operand: bytes (2p, 2p+1) of `word` -> two 16-bit elements.  3 VALU ops
// (int8: 2x v_cvt_f32_i32 with SDWA byte select + v_cvt_pk_bf16_f32), sized to fit the issue gap of one 16x16x32 MFMA.
template <int DT, int FMT>
__device__ __forceinline__ uint32_t convert_pair(uint32_t word, int p /* 0 or 1 */) {
  float f0, f1;
  if constexpr (FMT == W_I8) {
    f0 = p == 0 ? (float)(int8_t)(word & 0xFFu) : (float)(int8_t)((word >> 16) & 0xFFu);
    f1 = p == 0 ? (float)(int8_t)((word >> 8) & 0xFFu) : (float)(int8_t)(word >> 24);
  } else if constexpr (FMT == W_F8E4M3) {
    const f32x2 v = p == 0 ? __builtin_amdgcn_cvt_pk_f32_fp8((int)word, false) : __builtin_amdgcn_cvt_pk_f32_fp8((int)word, true);
    f0 = v.x;
    f1 = v.y;
  } else {
    const f32x2 v = p == 0 ? __builtin_amdgcn_cvt_pk_f32_bf8((int)word, false) : __builtin_amdgcn_cvt_pk_f32_bf8((int)word, true);
    f0 = v.x;
    f1 = v.y;
  }
  return Mma<DT>::pack(f0, f1);
}

template <int DT, int FMT>
__global__ void __launch_bounds__(512, 1) qbytes_mfma_v2_kernel(const Args a) {
  using E = Elem<DT>;
  using T = typename E::T;
  using V8 = typename Mma<DT>::V8;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int M = a.M, N = a.N, K = a.K;
  const int nk = K / BK;

  // XCD-aware tile order: consecutive block ids land on different XCDs (block b -> XCD b % 8); give each XCD a
  // contiguous band of tiles so that neighbouring tiles (which share an activation or weight panel) share an L2.
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int nwg = tiles_n * tiles_m;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane DMA source pointers (advance by BK per tile) ----------------------------------------
  const T* xg = reinterpret_cast<const T*>(a.x);
  const uint8_t* asrc[4];
  const uint8_t* wsrc[2];
  int adst[4], wdst[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int R = (j * 8 + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz_a(R);
    int m = m0 + R;
    m = m < M ? m : M - 1;
    asrc[j] = reinterpret_cast<const uint8_t*>(xg + (size_t)m * K + c * 8);
    adst[j] = (j * 8 + wave) * 1024;  // wave-uniform: the DMA adds lane*16
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int R = (j * 8 + wave) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz_w(R);
    int n = n0 + R;
    n = n < N ? n : N - 1;
    wsrc[j] = a.w + (size_t)n * K + c * 16;
    wdst[j] = A_BYTES + (j * 8 + wave) * 1024;
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  auto issue = [&](int kt, int stage) {
    const uint32_t st = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE_BYTES);
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(asrc[j] + (size_t)kt * (BK * 2), st + adst[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(wsrc[j] + (size_t)kt * BK, st + wdst[j]);
  };

  // ---- fragment read offsets (constant per lane) ------------------------------------------------------
  // activation fragment mf (16 rows) of this wave, k-half kk: logical chunk 2*(lane>>4) + kk of row R
  // weight fragment j (16 rows): ONE 16-byte read per K-tile, logical chunk lane>>4
  int aoff[8][2], boff[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int R = wm * 128 + i * 16 + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) aoff[i][kk] = R * 128 + ((((lane >> 4) * 2 + kk) ^ swz_a(R)) << 4);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int R = wn * 64 + j * 16 + (lane & 15);
    boff[j] = A_BYTES + R * 64 + (((lane >> 4) ^ swz_w(R)) << 4);
  }

  // acc[j][mf]: weight fragment j is the MFMA A operand (rows n), activation fragment mf the B operand (cols m):
  // D[n][m], so a lane ends up with 4 CONSECUTIVE output features of one token -> 8-byte stores in the epilogue.
  f32x4 acc[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline.  A K-tile is consumed in four 16-MFMA blocks S0..S3 = (k-half, token-half).  Activation
  // fragments ping-pong between two register sets (xa / xb) one block ahead of their use.  The weight tile is read
  // once per K-tile (wraw, 16 bytes per fragment) and converted in 3-instruction pieces placed in the issue gaps of
  // the MFMAs: the k-half-1 operand (w1) during S0, the NEXT tile's k-half-0 operand (w0) during S3.
  // Two barriers per K-tile: B1 (before S2) makes tile kt+1 visible so its weight bytes can be fetched during S2;
  // B2 (before S3) retires every wave's reads of tile kt, after which its stage is refilled by the DMA of tile kt+3.
  uint4 wraw[4];
  uint32_t w0[4][4], w1[4][4];
  V8 xa[4], xb[4];
  auto read_w = [&](const uint8_t* st) {
#pragma unroll
    for (int j = 0; j < 4; ++j) wraw[j] = *reinterpret_cast<const uint4*>(st + boff[j]);
  };
  auto read_x = [&](V8(&f)[4], const uint8_t* st, int kk, int h) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const V8*>(st + aoff[h * 4 + i][kk]);
  };
  auto wword = [&](int j, int kk, int d) -> uint32_t {  // source dword of output dword d of fragment j, k-half kk
    const uint32_t lo = kk == 0 ? wraw[j].x : wraw[j].z, hi = kk == 0 ? wraw[j].y : wraw[j].w;
    return d < 2 ? lo : hi;
  };
  auto as_v8 = [&](const uint32_t(&w)[4]) { return __builtin_bit_cast(V8, make_uint4(w[0], w[1], w[2], w[3])); };

  // Alternating-phase schedule.  Waves 0-3 ("G0", token rows 0..127) and waves 4-7 ("G1") share the four SIMDs
  // pairwise (wave w and w+4 sit on the same SIMD).  Each wave alternates LOAD phases (LDS-DMA issue, fragment
  // ds_reads, weight conversion - no MFMA) and COMPUTE phases (back-to-back MFMAs with every operand already in
  // registers); every phase ends with a workgroup barrier and G1 runs exactly one phase behind G0, so while one wave
  // of a SIMD computes its partner loads.  A load phase drains its ds_reads (lgkmcnt(0)) BEFORE its barrier, so "every
  // wave passed the barrier" implies "every read issued so far has returned": the stage of tile kt-1 can be refilled
  // from the first load phase of tile kt on, and tile kt+1 is visible to all from the second load phase of tile kt on.
#ifdef QH_PHASE_TIMING
  const int grp = a.variant == 3 ? (wave >> 2) ^ 1 : wave >> 2;  // variant 3: the younger waves lead
  if (a.variant == 1 && grp == 1) __builtin_amdgcn_s_setprio(1);
#else
  const int grp = wave >> 2;
#endif
  auto issue_piece = [&](int kt, int stage, int piece) {
    const uint32_t stb = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE_BYTES);
    if (piece < 4)
      glds16(asrc[piece] + (size_t)kt * (BK * 2), stb + adst[piece]);
    else
      glds16(wsrc[piece - 4] + (size_t)kt * BK, stb + wdst[piece - 4]);
  };
  auto cvt_half = [&](uint32_t(&w)[4][4], int kk, int half) {
#pragma unroll
    for (int q = half * 8; q < half * 8 + 8; ++q) w[q >> 2][q & 3] = convert_pair<DT, FMT>(wword(q >> 2, kk, q & 3), q & 1);
  };
  auto end_load_phase = [&]() {
    // Pin the schedule first: hipcc otherwise sinks register-only VALU work (the conversions) below the barrier, into
    // the head of the compute phase, where it delays the first MFMAs of a wave that should only be issuing MFMAs.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto end_compute_phase = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  issue(0, 0);
  if (nk > 1) issue(1, 1);
  if (nk > 1)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // tile 0 visible
  asm volatile("" ::: "memory");
  read_w(smem);
  cvt_half(w0, 0, 0);
  cvt_half(w0, 0, 1);
  __builtin_amdgcn_sched_barrier(0);
  if (grp == 1 && QH_STAGGER) end_compute_phase();  // G1 starts one phase late

  // Four phases per K-tile and wave: La Ca Lb Cb, 32 MFMAs per compute phase (k-half 0, then k-half 1, all 8 token
  // fragments), G1 one phase behind G0.
  //   La: DMA pieces 0-2 of tile kt+2 | x(kk0) -> xa,xb | convert w1 (k-half 1 of this tile) | wait: own share of tile kt+1
  //   Lb: DMA pieces 3-5              | next tile's weight bytes -> wraw, x(kk1) -> xa,xb | convert next tile's w0
  int cur = 0;  // stage of tile kt
#ifdef QH_PHASE_TIMING
  unsigned long long stamps[17];
  int stamp_base = 0;
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const uint8_t* st = smem + cur * STAGE_BYTES;
    const int nxt = cur == STAGES - 1 ? 0 : cur + 1;
    const int nxt2 = nxt == STAGES - 1 ? 0 : nxt + 1;
    const uint8_t* sn = smem + nxt * STAGE_BYTES;
    const bool more = kt + 1 < nk, more2 = kt + 2 < nk;
#ifdef QH_PHASE_TIMING
    const bool stamp_on = (kt == 20 || kt == 21) && blockIdx.x == 7;
    stamp_base = (kt - 20) * 17;
#endif
    QH_STAMP(0);
    // ---- La ----
    if (more2) {
      issue_piece(kt + 2, nxt2, 0);
      issue_piece(kt + 2, nxt2, 1);
      issue_piece(kt + 2, nxt2, 2);
    }
    read_x(xa, st, 0, 0);
    read_x(xb, st, 0, 1);
    cvt_half(w1, 1, 0);
    cvt_half(w1, 1, 1);
    if (more) {
      if (more2)
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // all of tile kt+1; the 3 pieces of tile kt+2 may be in flight
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    QH_STAMP(1);
    end_load_phase();
    QH_STAMP(2);
    __builtin_amdgcn_sched_barrier(0);
    // ---- Ca ----
    QH_PRIO_UP();
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q & 3][q >> 2] = Mma<DT>::run(as_v8(w0[q & 3]), xa[q >> 2], acc[q & 3][q >> 2]);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q & 3][4 + (q >> 2)] = Mma<DT>::run(as_v8(w0[q & 3]), xb[q >> 2], acc[q & 3][4 + (q >> 2)]);
    QH_PRIO_DOWN();
    QH_STAMP(3);
    end_compute_phase();
    QH_STAMP(4);
    __builtin_amdgcn_sched_barrier(0);
    // ---- Lb ----
    if (more) read_w(sn);  // w1 of this tile was converted in La: wraw is free
    if (more2) {
      issue_piece(kt + 2, nxt2, 3);
      issue_piece(kt + 2, nxt2, 4);
      issue_piece(kt + 2, nxt2, 5);
    }
    read_x(xa, st, 1, 0);
    read_x(xb, st, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    cvt_half(w0, 0, 0);  // next tile's k-half-0 operand (garbage but unused on the last tile)
    cvt_half(w0, 0, 1);
    QH_STAMP(5);
    end_load_phase();
    QH_STAMP(6);
    __builtin_amdgcn_sched_barrier(0);
    // ---- Cb ----
    QH_PRIO_UP();
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q & 3][q >> 2] = Mma<DT>::run(as_v8(w1[q & 3]), xa[q >> 2], acc[q & 3][q >> 2]);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q & 3][4 + (q >> 2)] = Mma<DT>::run(as_v8(w1[q & 3]), xb[q >> 2], acc[q & 3][4 + (q >> 2)]);
    QH_PRIO_DOWN();
    QH_STAMP(7);
    end_compute_phase();
    QH_STAMP(8);
    __builtin_amdgcn_sched_barrier(0);
#ifdef QH_PHASE_TIMING
    if (stamp_on && lane == 0) for (int i = 0; i < 17; ++i) a.dbg[wave * 34 + stamp_base + i] = stamps[i];
    if (stamp_on && lane == 0 && kt == 21) a.dbg[wave * 34 + 33] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
#endif
    cur = nxt;
  }
  if (grp == 0 && QH_STAGGER) end_compute_phase();  // G0 finishes one phase early: same barrier count for every wave

  // ---- epilogue: per-channel scale on the fp32 accumulator, optional bias, full-line stores ------------------
  // The stage memory is free once every wave has left the K loop.  Each wave parks its 128x64 result (16 KiB, rows of
  // 128 bytes, 8-byte chunks XOR-swizzled by ((row & 7) << 1) so that both the ds_write_b64 below and the ds_read_b128
  // that follows stay (nearly) conflict free) and streams it out as whole 128-byte lines: 8 rows x 128 B per store
  // instruction instead of 16 rows x 32 B.
  T* yg = reinterpret_cast<T*>(a.y);
  const bool has_bias = a.bias != nullptr;
  const bool full = (m0 + BM <= M) && (n0 + BN <= N) && (N % 8 == 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  uint8_t* park = smem + wave * (128 * 128);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int nb = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;  // 4 consecutive output features nb..nb+3
    float sc[4], bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nb + r < N ? nb + r : N - 1;
      sc[r] = E::to_f32(reinterpret_cast<const T*>(a.scale)[n]);
      bv[r] = has_bias ? E::to_f32(reinterpret_cast<const T*>(a.bias)[n]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      T out[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[j][i][r] * sc[r];
        if (has_bias) v = E::to_f32(E::from_f32(v)) + bv[r];
        out[r] = E::from_f32(v);
      }
      const int row = i * 16 + (lane & 15);
      const int chunk = (j * 4 + (lane >> 4)) ^ ((row & 7) << 1);
      *reinterpret_cast<uint2*>(park + row * 128 + chunk * 8) = *reinterpret_cast<const uint2*>(out);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: no barrier needed
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int row = t * 8 + (lane >> 3);
    const int c16 = lane & 7;  // 16-byte column chunk = 8 output features
    const uint4 v = *reinterpret_cast<const uint4*>(park + row * 128 + (((c16 * 2) ^ ((row & 7) << 1)) * 8));
    const int m = m0 + wm * 128 + row;
    const int n = n0 + wn * 64 + c16 * 8;
    if (full) {
      *reinterpret_cast<uint4*>(yg + (size_t)m * N + n) = v;
    } else if (m < M) {
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (n + r < N) yg[(size_t)m * N + n + r] = e[r];
    }
  }
}

template <int DT, int FMT>
static int launch(const Args& a, hipStream_t stream) {
  static bool attr_done = false;
  constexpr int lds = STAGES * STAGE_BYTES;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbytes_mfma_v2_kernel<DT, FMT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const int tiles = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM);
  hipLaunchKernelGGL((qbytes_mfma_v2_kernel<DT, FMT>), dim3(tiles), dim3(512), lds, stream, a);
  return launch_status();
}

}  // namespace v2

// Large-tile kernel: worth it when the grid fills the chip with 256x256 tiles.
bool qbytes_mfma_v2_supported(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  const bool bd = b_dtype == QUANTO_HIP_I8 || b_dtype == QUANTO_HIP_F8_E4M3FN || b_dtype == QUANTO_HIP_F8_E5M2;
  return bd && a_dtype == out_dtype && (out_dtype == QUANTO_HIP_BF16 || out_dtype == QUANTO_HIP_F16) && K % v2::BK == 0 &&
         K >= 2 * v2::BK && M >= 1 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30);
}

int qbytes_mm_mfma_v2(const void* x, const void* w, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int a_dtype,
                      int b_dtype, int out_dtype, hipStream_t stream) {
  if (!qbytes_mfma_v2_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16) return QUANTO_HIP_EALIGN;
  v2::Args a{x, reinterpret_cast<const uint8_t*>(w), s, bias, y, (int)M, (int)N, (int)K};
#define QH_CASE(DT, FMT) return v2::launch<DT, FMT>(a, stream)
  if (out_dtype == QUANTO_HIP_BF16) {
    if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_BF16, v2::W_I8);
    if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_BF16, v2::W_F8E4M3);
    QH_CASE(QUANTO_HIP_BF16, v2::W_F8E5M2);
  }
  if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_F16, v2::W_I8);
  if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_F16, v2::W_F8E4M3);
  QH_CASE(QUANTO_HIP_F16, v2::W_F8E5M2);
#undef QH_CASE
}

}  // namespace qh
