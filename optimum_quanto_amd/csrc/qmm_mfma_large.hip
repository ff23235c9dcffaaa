// qbytes_mm MFMA GEMM, v3: 256x256x64 tile, FOUR waves, each owning a 128x128 output block (256 accumulator registers).
//
// y[M,N] = (x[M,K] @ q[N,K]^T) * scale[N]   with x bf16/fp16 and q int8 / fp8 (1 byte per weight).
//
// Why four big waves instead of v2's eight 128x64 waves: the LDS read traffic of a K-tile is set by the per-wave block
// shape, (rows_m + rows_n) * BK bytes per wave.  v2 moves 8 x 20 KiB = 160 KiB of fragments (+48 KiB of DMA writes)
// through the 128 B/clk LDS per K-tile - 1664 cycles next to 2048 cycles of MFMA work per SIMD, i.e. the LDS is as
// busy as the matrix pipe and every imperfect overlap shows (measured 40% of the MFMA peak).  128x128 blocks read
// 4 x 24 KiB = 96 KiB: 1150 cycles including the DMA writes, 56% of the MFMA time.
//
// One wave per SIMD means no partner wave hides anything: the K loop is a single software-pipelined instruction stream
// in which every MFMA (16 cycles of matrix pipe, one 4-cycle issue slot) is followed by at most three other
// instructions.  Per K-tile and wave: 128 MFMA + 192 VALU (weight conversion) + 24 ds_read_b128 + 12 LDS-DMA issues.
//   step s = (kk, i), 16 per K-tile: 8 MFMAs acc[j][i] += W_kk[j] * x(i, kk), j = 0..7
//     phase kk=0 converts the k-half-1 operand of THIS tile (W1[i]) in the MFMA issue gaps,
//     phase kk=1 converts the k-half-0 operand of the NEXT tile (W0[i]); the raw 16-byte weight fragments of the next
//     tile are fetched from LDS as soon as their register is dead; activation fragments stream two steps ahead.
//   ONE workgroup barrier per K-tile (at the tile boundary): before it every wave waits for its own DMA share of tile
//   kt+1 (issued a full tile earlier); after it the stage of tile kt-1 is refilled with tile kt+2.
// LDS image, swizzles, XCD-aware tile order and the D[n][m] accumulator orientation are those of the 128x128 kernel (qmm_mfma.hip).
#include <cstdlib>
#include <type_traits>

#include "qmm_large_common.h"

namespace qh {
namespace lt {

#ifdef QH_LT_STAMPS
__device__ unsigned long long g_stamps[256 * 8];  // per workgroup: s_memrealtime (100 MHz) at entry / loop start / loop end / exit
#define QH_LT_STAMP(i) do { if (threadIdx.x == 0) g_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define QH_LT_STAMP(i) do { } while (0)
#endif

// f(integral_constant<int, 0>) ... f(integral_constant<int, R-1>): the weights-direct loop is unrolled over its ring slots
template <int... Ps, class F>
__device__ __forceinline__ void for_each_slot_impl(std::integer_sequence<int, Ps...>, F&& f) {
  (f(std::integral_constant<int, Ps>{}), ...);
}
template <int R, class F>
__device__ __forceinline__ void for_each_slot(F&& f) {
  for_each_slot_impl(std::make_integer_sequence<int, R>{}, f);
}

// Tile configurations (BM x BN workgroup tile, WM x WN waves, each wave (BM/WM) x (BN/WN)):
//   256 x 256, 2 x 4 waves of 128 x 64  - the default for grids that fill the chip: two interleaving streams per SIMD
//   256 x 256, 2 x 2 waves of 128 x 128 - one stream per SIMD, least LDS traffic (experiments: QUANTO_HIP_LARGE_CFG=1)
//   256 x 256, 1 x 8 waves of 256 x 32  - every weight fragment converted once per workgroup, twice the activation reads
//                                         (experiments: QUANTO_HIP_LARGE_CFG=3).  All three 256-tile layouts run 4096^3
//                                         within 2 % of each other (102 - 104 us)
//   128 x 128, 1 x 4 waves of 128 x 32  - four times the workgroups for prefill shapes whose 256-tiles cannot fill 256 CUs
//                                         (e.g. M = 512).  All waves side by side along the features: every converted
//                                         weight fragment feeds 8 MFMAs (1.5 VALU ops per MFMA, as in the 256-tiles); a
//                                         2 x 2 layout converts twice as much and is VALU-issue bound (measured 105 us
//                                         vs this layout on cfg4)
//
// WD ("weights direct", WD = ring depth): the weight bytes never touch the LDS.  Every lane loads the 16 bytes of its own
// fragment row from global memory into a register ring of WD K-tiles, WD-1 tiles ahead of their use, and the LDS holds WD
// activation-only stages (128-tile: WD x 16 KiB instead of 3 x 24 KiB).  Activation DMA and weight loads are then in flight
// for WD-2 whole tiles instead of one.  It pays only where a workgroup has its CU to itself (one wave per SIMD, nobody to
// hide the DMA latency, and 512 registers / 160 KiB of LDS for one workgroup): see the launcher for the numbers.  Tried and rejected for the 256-tile (two waves per SIMD; a ring of two weight tiles, loaded in
// the second phase of tile t-2 and waited for in the middle of tile t-1, because 128 accumulators leave no room for four):
// 4096^3 121 -> 126 us, 8192^3 783 -> 811 us.
template <int DT, int FMT, int BM, int BN, int WM, int WN, int WD = 0>
__global__ void __launch_bounds__(WM * WN * 64, 1) qbytes_mfma_large_kernel(const Args a) {
  constexpr int NWAVES = WM * WN;
  constexpr int MI = BM / WM / 16;              // 16-token fragments per wave
  constexpr int NJ = BN / WN / 16;              // 16-feature fragments per wave
  constexpr int STEPS = 2 * MI;                 // (k-half, token fragment) steps per K-tile
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = WD ? 0 : BN * BK, STAGE_BYTES = A_BYTES + W_BYTES;
  constexpr int APIECES = BM / 8 / NWAVES;      // activation DMA pieces (8 rows x 128 B) per wave and K-tile
  constexpr int WPIECES = WD ? 0 : BN / 16 / NWAVES;  // weight DMA pieces (16 rows x 64 B) per wave and K-tile
  constexpr int NPIECES = APIECES + WPIECES;
  constexpr int ND = (NJ * 4 + MI - 1) / MI;    // converted dwords per step (one phase converts NJ*4 dwords in MI steps)
  constexpr int DSTEPS = WD ? 1 : NPIECES % 6 == 0 ? 6 : 3;  // the DMA of tile kt+2 is issued over the first DSTEPS steps of tile kt
  constexpr int PPS = NPIECES / DSTEPS;         // pieces per step
  static_assert(WD || (NPIECES % DSTEPS == 0 && PPS <= NJ), "unsupported tile configuration");
  static_assert(STEPS % 4 == 0 && ND <= NJ && APIECES + NJ <= STEPS, "unsupported tile configuration");
  using E = Elem<DT>;
  using T = typename E::T;
  using V8 = typename Mma<DT>::V8;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  QH_LT_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int M = a.M, N = a.N, K = a.K;
  const int S = a.S;
  const int tile_id = S > 1 ? blockIdx.x / S : blockIdx.x, sp = S > 1 ? blockIdx.x - tile_id * S : 0;
  const int nk = K / BK / S;  // K-tiles of this workgroup's K-range
  const int kt0 = sp * nk;

  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  int tm, tn;
  tile_coords(tile_id, tiles_m, tiles_n, a.group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- DMA: per K-tile 32 activation pieces (8 rows x 128 B) + 16 weight pieces (16 rows x 64 B) of 1 KiB; 8 + 4 per wave
  uint32_t asrc[APIECES], wsrc[WPIECES > 0 ? WPIECES : 1];
#pragma unroll
  for (int j = 0; j < APIECES; ++j) {
    const int R = (j * NWAVES + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz_a(R);
    int m = m0 + R;
    m = m < M ? m : M - 1;
    asrc[j] = (uint32_t)(((size_t)m * K + c * 8 + (size_t)kt0 * BK) * 2);
  }
#pragma unroll
  for (int j = 0; j < WPIECES; ++j) {
    const int R = (j * NWAVES + wave) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz_w(R);
    int n = n0 + R;
    n = n < N ? n : N - 1;
    wsrc[j] = (uint32_t)((size_t)n * K + c * 16 + (size_t)kt0 * BK);
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.x);
  auto issue_piece = [&](int kt, int stage, int piece) {
    const uint32_t stb = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE_BYTES);
    if (piece < APIECES)
      glds16(xbase + (size_t)kt * (BK * 2), asrc[piece], stb + (piece * NWAVES + wave) * 1024);
    else
      glds16(a.w + (size_t)kt * BK, wsrc[piece - APIECES], stb + A_BYTES + ((piece - APIECES) * NWAVES + wave) * 1024);
  };

  // ---- fragment read offsets: fragment i / j only adds a compile-time multiple of 2048 / 1024 bytes ----------------
  const int ra = wm * (MI * 16) + (lane & 15), rw = wn * (NJ * 16) + (lane & 15);
  int aoff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    // lane group g reads 16-byte chunk 2g + kk of the row for k-half kk (any bijection works as long as the weight operand uses
    // the same one).  W_DENSE: chunk 4 kk + g, so that the four lane groups of one weight load cover 64 contiguous bytes
    const int chunk = FMT == W_DENSE ? kk * 4 + (lane >> 4) : (lane >> 4) * 2 + kk;
    aoff[kk] = ra * 128 + ((chunk ^ swz_a(ra)) << 4);
  }
  const int boff = A_BYTES + rw * 64 + (((lane >> 4) ^ swz_w(rw)) << 4);

  f32x4 acc[NJ][MI];  // acc[j][i]: features j*16.., tokens i*16..
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // r6: per-feature scale / bias of the tile wait in LDS behind the ring.  The epilogue used to fetch them from global memory AFTER the
  // K loop: a full round trip in front of the first output byte, on the critical path of every tile (all tiles end together).  One 2-byte
  // load per thread, issued in front of the prologue's DMA (oldest entry of the in-order vector-memory queue: complete at the prologue's
  // counted wait), parked behind the ring before the first barrier.  As asm: a load hipcc can see makes it wait vmcnt(0) at the store.
  constexpr int RING_BYTES = (WD != 0 ? WD : STAGES) * STAGE_BYTES;
  static_assert(NWAVES * 64 >= 2 * BN, "one table entry per thread");
  uint32_t tab_val = 0;
  const bool tab_scale = tid < BN ? a.scale != nullptr : a.bias != nullptr;
  if (tid < 2 * BN && tab_scale) {
    int n = n0 + (tid < BN ? tid : tid - BN);
    n = n < N ? n : N - 1;
    const T* src = reinterpret_cast<const T*>(tid < BN ? a.scale : a.bias) + n;
    asm volatile("global_load_ushort %0, %1, off" : "=v"(tab_val) : "v"(src) : "memory");
  }
  auto park_table = [&]() {  // after the prologue's vmcnt wait, before its barrier
    asm volatile("" : "+v"(tab_val));
    if (tid < 2 * BN) {
      const uint16_t one = DT == QUANTO_HIP_BF16 ? 0x3F80 : 0x3C00;
      reinterpret_cast<uint16_t*>(smem + RING_BYTES)[tid] = tab_scale ? (uint16_t)tab_val : (tid < BN ? one : (uint16_t)0);
    }
  };

  uint32_t w0[NJ][4], w1[NJ][4];
  V8 xf[4];              // activation fragments, two steps ahead (ring of 4: STEPS is a multiple of 4, the ring stays aligned)
  auto as_v8 = [&](const uint32_t(&w)[4]) { return __builtin_bit_cast(V8, make_uint4(w[0], w[1], w[2], w[3])); };
  auto read_x = [&](const uint8_t* st, int i, int kk) -> V8 { return *reinterpret_cast<const V8*>(st + aoff[kk] + i * 2048); };
  using yes = std::integral_constant<bool, true>;

  if constexpr (WD != 0) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    constexpr bool DENSE = FMT == W_DENSE;
    constexpr int WB = DENSE ? 2 : 1;      // bytes per weight = 16-byte loads per fragment and K-tile
    constexpr int WLOADS = NJ * WB;        // weight loads per wave and K-tile
    static_assert(APIECES + WLOADS <= STEPS, "unsupported tile configuration");
    // ---- weights: lane (r = lane & 15, g = lane >> 4) of fragment j owns bytes 16g..16g+15 of feature row j*16 + r of the
    // K-tile: k-half 0 operand in .xy, k-half 1 in .zw (the same 16 bytes the LDS path reads back from its weight image).
    // W_DENSE: bytes 16g..16g+15 (k-half 0) and 64+16g.. (k-half 1) of the 128-byte row
    uint32_t wofs[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int n = n0 + wn * (NJ * 16) + j * 16 + (lane & 15);
      n = n < N ? n : N - 1;
      wofs[j] = (uint32_t)(((size_t)n * K + (size_t)kt0 * BK) * WB + (lane >> 4) * 16);
    }
    // rg[t % RING]: weight bytes of tile t, loaded during tile t-RING+1, complete at the end of tile t-2.  W_DENSE: the
    // lane's 32 bytes are the two MFMA operands themselves ([..][0] k-half 0, [..][1] k-half 1), no conversion and no w0 / w1
    constexpr int RING = WD;                 // K-tiles in flight + in use: LDS stages and weight register slots
    constexpr int GROUP = APIECES + WLOADS;  // vector-memory instructions a wave issues per K-tile
    constexpr int AHEAD = (RING - 3) * GROUP;  // tiles kt+3 .. kt+RING-1 may still be in flight when tile kt ends
    static_assert(AHEAD <= 63, "vmcnt is a 6-bit counter");
    u32x4 rg[RING][NJ][WB];
    auto load_w = [&](int kt, u32x4 (&dst)[WB], int j) {
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst[0]) : "v"(wofs[j]), "s"(a.w + (size_t)kt * (BK * WB)) : "memory");
      if constexpr (DENSE)
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(dst[1]) : "v"(wofs[j]), "s"(a.w + (size_t)kt * (BK * WB)) : "memory");
    };
    auto word = [&](const u32x4& r, int kk, int d) -> uint32_t { return kk == 0 ? (d < 2 ? r.x : r.y) : (d < 2 ? r.z : r.w); };

    // ---- prologue: tiles 0..RING-2 in flight, tiles 0 and 1 complete, W0(0) converted, x(0..1, kk0) of tile 0 in registers -
#pragma unroll
    for (int t = 0; t < RING - 1; ++t) {
#pragma unroll
      for (int p = 0; p < APIECES; ++p) issue_piece(t, t, p);
#pragma unroll
      for (int j = 0; j < NJ; ++j) load_w(t, rg[t][j], j);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD) : "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int b = 0; b < WB; ++b) asm volatile("" : "+v"(rg[t][j][b]));
    park_table();
    QH_LT_STAMP(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    QH_LT_STAMP(2);
    if constexpr (!DENSE) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int d = 0; d < 4; ++d) w0[j][d] = convert_pair<DT, FMT>(word(rg[0][j][0], 0, d), d & 1);
    }
    xf[0] = read_x(smem, 0, 0);
    xf[1] = read_x(smem, 1, 0);

    // K-tile kt with kt % RING == P: same step schedule as the LDS-weight loop below; the activation pieces and weight loads
    // of tile kt+RING-1 take one issue slot each in the first APIECES + NJ steps
    auto tile = [&](auto p_tag, int kt, auto dma_tag, auto barrier_tag) {
      constexpr int P = decltype(p_tag)::value;
      constexpr bool DMA = decltype(dma_tag)::value, BARRIER = decltype(barrier_tag)::value;
      constexpr int PN = (P + 1) % RING, PF = (P + RING - 1) % RING;  // next tile's slot; slot refilled during this tile
      const uint8_t* st = smem + P * STAGE_BYTES;
      const uint8_t* sn = smem + PN * STAGE_BYTES;
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const int kk = s / MI, i = s % MI;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if constexpr (DENSE) {
            acc[j][i] = Mma<DT>::run(__builtin_bit_cast(V8, rg[P][j][kk == 0 ? 0 : WB - 1]), xf[s & 3], acc[j][i]);
          } else {
            if (kk == 0)
              acc[j][i] = Mma<DT>::run(as_v8(w0[j]), xf[s & 3], acc[j][i]);
            else
              acc[j][i] = Mma<DT>::run(as_v8(w1[j]), xf[s & 3], acc[j][i]);
            if (j < ND && i * ND + j < NJ * 4) {
              const int c = i * ND + j, f = c >> 2, d = c & 3;
              if (kk == 0)
                w1[f][d] = convert_pair<DT, FMT>(word(rg[P][f][0], 1, d), d & 1);   // this tile's k-half 1
              else
                w0[f][d] = convert_pair<DT, FMT>(word(rg[PN][f][0], 0, d), d & 1);  // next tile's k-half 0
            }
          }
          if (j == (ND < NJ ? ND : 0))
            xf[(s + 2) & 3] = s + 2 < STEPS ? read_x(st, (s + 2) % MI, (s + 2) / MI) : read_x(sn, s + 2 - STEPS, 0);
          if (DMA && j == NJ - 1 && s < APIECES + NJ) {
            if (s < APIECES)
              issue_piece(kt + RING - 1, PF, s);
            else
              load_w(kt + RING - 1, rg[PF][s - APIECES], s - APIECES);  // W_DENSE: two loads in this slot
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (BARRIER) {
        // the activation rows of tile kt+2 feed the fragment prefetch at the end of tile kt+1, its weight registers the
        // conversions of tile kt+1's second phase.  Younger tiles stay in flight: RING-3 of them in the steady state, in
        // the tail (no more DMA; kt % RING == P because nk is a multiple of RING) the RING-3-P that are left.
        constexpr int LEFT = DMA ? AHEAD : (RING - 3 - P > 0 ? (RING - 3 - P) * GROUP : 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LEFT) : "memory");
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int b = 0; b < WB; ++b) asm volatile("" : "+v"(rg[(P + 2) % RING][j][b]));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    };
    QH_LT_STAMP(3);
    int kt = 0;
    for (; kt + RING < nk; kt += RING)  // nk % RING == 0, nk >= 2 * RING (checked by the launcher)
      for_each_slot<RING>([&](auto p) { tile(p, kt + decltype(p)::value, yes{}, yes{}); });
    for_each_slot<RING>([&](auto p) {  // last turn: only its first tile still has something to fetch (tile nk-1)
      constexpr int P = decltype(p)::value;
      tile(p, kt + P, std::integral_constant<bool, P == 0>{}, std::integral_constant<bool, P != RING - 1>{});
    });
  } else {
    uint4 raw[NJ];         // 16 weight bytes per fragment: k-half 0 in .xy, k-half 1 in .zw
    auto rawword = [&](int j, int kk, int d) -> uint32_t {
      const uint32_t lo = kk == 0 ? raw[j].x : raw[j].z, hi = kk == 0 ? raw[j].y : raw[j].w;
      return d < 2 ? lo : hi;
    };
    auto read_raw = [&](const uint8_t* st, int j) -> uint4 { return *reinterpret_cast<const uint4*>(st + boff + j * 1024); };

    // ---- prologue: tiles 0 and 1 in flight, tile 0 visible, W0(0) converted, x(0..1, kk0) of tile 0 in registers -----------
#pragma unroll
    for (int p = 0; p < NPIECES; ++p) issue_piece(0, 0, p);
    if (nk > 1) {
#pragma unroll
      for (int p = 0; p < NPIECES; ++p) issue_piece(1, 1, p);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile 1 too: its weight bytes are fetched during tile 0
    park_table();
    QH_LT_STAMP(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    QH_LT_STAMP(2);
#pragma unroll
    for (int j = 0; j < NJ; ++j) raw[j] = read_raw(smem, j);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int d = 0; d < 4; ++d) w0[j][d] = convert_pair<DT, FMT>(rawword(j, 0, d), d & 1);
    xf[0] = read_x(smem, 0, 0);
    xf[1] = read_x(smem, 1, 0);

    // One K-tile.  The source order below IS the schedule: a sched_barrier after every MFMA group keeps hipcc from
    // clustering the conversions (it otherwise hoists ~50 VALU ops in front of the first MFMA of a tile, which leaves the
    // matrix pipe idle for their whole issue time).
    // Stage-dependent addresses are loop constants: the loop is unrolled over the three stages (tile kt lives in stage kt % 3),
    // every LDS base sits in a register and every DMA destination in an SGPR, so a K-tile carries no address arithmetic at
    // all - each scalar or vector ALU instruction in this loop takes an MFMA issue gap.  Run-time stage bookkeeping was 21 of
    // the 247 instructions per tile and wave: (1024,8192,4096) as 512 paired 128-tiles 68 -> 59 us, 4096^3 101 -> 100 us.
    const uint8_t* xb[STAGES][2];
    const uint8_t* wb[STAGES];
    uint32_t mdst[STAGES][NPIECES];
#pragma unroll
    for (int t = 0; t < STAGES; ++t) {
      xb[t][0] = smem + t * STAGE_BYTES + aoff[0];
      xb[t][1] = smem + t * STAGE_BYTES + aoff[1];
      wb[t] = smem + t * STAGE_BYTES + boff;
#pragma unroll
      for (int p = 0; p < NPIECES; ++p)
        mdst[t][p] = __builtin_amdgcn_readfirstlane(lds_base + t * STAGE_BYTES +
                                                    (p < APIECES ? (p * NWAVES + wave) * 1024 : A_BYTES + ((p - APIECES) * NWAVES + wave) * 1024));
    }
    auto tile = [&](auto p_tag, int kt, auto dma_tag, auto barrier_tag) {
      constexpr int P = decltype(p_tag)::value, PN = (P + 1) % STAGES, PF = (P + 2) % STAGES;
      const bool DMA = dma_tag, BARRIER = barrier_tag;  // integral_constants in the steady state, run-time flags in the tail
      auto rx = [&](int stage, int i, int kk) -> V8 { return *reinterpret_cast<const V8*>(xb[stage][kk] + i * 2048); };
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const int kk = s / MI, i = s % MI;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (kk == 0)
            acc[j][i] = Mma<DT>::run(as_v8(w0[j]), xf[s & 3], acc[j][i]);
          else
            acc[j][i] = Mma<DT>::run(as_v8(w1[j]), xf[s & 3], acc[j][i]);
          if (j < ND && i * ND + j < NJ * 4) {
            // conversion: step i of a phase produces dwords i*ND .. i*ND+ND-1 of the phase's NJ*4 (fragment-major)
            const int c = i * ND + j, f = c >> 2, d = c & 3;
            if (kk == 0)
              w1[f][d] = convert_pair<DT, FMT>(rawword(f, 1, d), d & 1);  // this tile's k-half 1
            else
              w0[f][d] = convert_pair<DT, FMT>(rawword(f, 0, d), d & 1);  // next tile's k-half 0 (raw[f] already holds tile kt+1)
          }
          if (j == (ND < NJ ? ND : 0)) {
            // activation fragment of step s+2 (the first two of the next tile at the end; garbage, unused, on the last tile)
            xf[(s + 2) & 3] = s + 2 < STEPS ? rx(P, (s + 2) % MI, (s + 2) / MI) : rx(PN, s + 2 - STEPS, 0);
          }
          if (j == (ND + 1 < NJ ? ND + 1 : NJ - 1)) {
            // next tile's raw weight bytes: fragment f is dead once the last dword of its k-half 1 is converted, i.e. after
            // phase-0 step (4f+3)/ND; it is reloaded in the following step (phase-1 step 0 for the last fragment)
#pragma unroll
            for (int f = 0; f < NJ; ++f)
              if (s == (4 * f + 3) / ND + 1) raw[f] = *reinterpret_cast<const uint4*>(wb[PN] + f * 1024);
          }
          if (j >= NJ - PPS && s < DSTEPS) {
            if (DMA) {
              const int piece = PPS * s + (j - (NJ - PPS));
              if (piece < APIECES)
                glds16(xbase + (size_t)(kt + 2) * (BK * 2), asrc[piece], mdst[PF][piece]);
              else
                glds16(a.w + (size_t)(kt + 2) * BK, wsrc[piece - APIECES], mdst[PF][piece]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (BARRIER) {
        // tile boundary: the own DMA share of tile kt+2, issued in the first steps of this tile, has landed -> barrier ->
        // everybody's share visible and every wave done with tile kt, whose stage the next tile refills.  vmcnt(0), not
        // "all but the newest pieces": tile kt+1 prefetches its successor's weight bytes and first activation fragments
        // out of that stage while it runs, so tile kt+2 must be complete before tile kt+1 starts.  Tried and rejected:
        // five 24 KiB stages for the 128-tile (one workgroup per CU instead of two: 110 us vs 86 us on cfg4) and a
        // three-step activation prefetch distance (cfg4 95 -> 119 us, 4096^3 unchanged); touching the lines of tile kt+4 with one
        // un-waited 4-byte load per lane so that the DMA hits in L2 (4096^3 127 -> 136 us); the two waves of a SIMD issuing
        // their DMA share in different phases of the tile (4096^3 102 -> 112 us; even the wave-uniform branch that selected
        // the phase, not taken, cost 5 us).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    static_assert(STAGES == 3, "the loop below is unrolled over three stages");
    QH_LT_STAMP(3);
    int kt = 0;
    for (; kt + 4 < nk; kt += 3) {  // three tiles that all still have a tile kt+2 to fetch
      tile(S0{}, kt, yes{}, yes{});
      tile(S1{}, kt + 1, yes{}, yes{});
      tile(S2{}, kt + 2, yes{}, yes{});
    }
    // tail: 2..4 tiles (nk >= 2), kt % 3 == 0; the last two have nothing left to prefetch, the last one no barrier
    const int rem = nk - kt;
    tile(S0{}, kt, rem > 2, true);
    tile(S1{}, kt + 1, rem > 3, rem > 2);
    if (rem > 2) tile(S2{}, kt + 2, false, rem > 3);
    if (rem > 3) tile(S0{}, kt + 3, false, false);
  }
  QH_LT_STAMP(4);

  // ---- epilogue: scale (+bias) on the fp32 accumulator; each wave parks MI*16 tokens x 64 features per pass -------------
  T* yg = reinterpret_cast<T*>(a.y);
  const bool has_bias = a.bias != nullptr;
  const bool full = (m0 + BM <= M) && (n0 + BN <= N) && (N % 8 == 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (S > 1) {
    // split-K reduction: system-coherent 16-byte stores / loads of the partial sums (no L2-wide fence), one arrival counter
    // per tile, the last workgroup to arrive sums in split order - see qbits_skinny.hip for the coherence argument
    // fragment-major layout: every store / load instruction of a wave covers 1 KiB of whole lines (lane-major - NJ*MI*16 bytes per lane -
    // wrote 16 bytes into every second line per instruction, and partial lines are what the write-through path is slow at)
    float* mine = a.partials + ((size_t)blockIdx.x * (NJ * MI) * (NWAVES * 64) + tid) * 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i)
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(mine + (j * MI + i) * (NWAVES * 64 * 4)), "v"(acc[j][i]) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) *flag = __hip_atomic_fetch_add(a.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (*flag != S - 1) return;
    if (tid == 0) __hip_atomic_store(a.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();  // the flag word is part of the parking area used below
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < S; ++q) {
      const float* theirs = a.partials + ((size_t)(tile_id * S + q) * (NJ * MI) * (NWAVES * 64) + tid) * 4;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        f32x4 v[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[i]) : "v"(theirs + (j * MI + i) * (NWAVES * 64 * 4)) : "memory");
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[i])::"memory");
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[j][i][r] += v[i][r];
      }
    }
  }
  constexpr int JP = NJ < 4 ? NJ : 4;  // feature fragments per pass: parked rows of JP*32 bytes
  constexpr int ROWB = JP * 32, LPR = ROWB / 16;  // lanes per parked row on the read side
  // r6: the wave's rows leave in RH halves - the stores of the first 64 rows drain while the second 64 are scaled, converted and parked (one
  // pass over all 128 rows ran the phases of all eight waves in step: first every wave on the VALU / LDS, then every wave on the store path)
  constexpr int RH = MI >= 8 ? 2 : 1, MIH = MI / RH;
  uint8_t* park = smem + wave * (MI * 16 * ROWB);
  const T* tab = reinterpret_cast<const T*>(smem + RING_BYTES);  // [scale x BN | bias x BN], parked by the prologue
#pragma unroll
  for (int p = 0; p < NJ / JP; ++p) {
    float sc[JP][4], bv[JP][4];
#pragma unroll
    for (int jj = 0; jj < JP; ++jj) {
      const int nl = wn * (NJ * 16) + (p * JP + jj) * 16 + (lane >> 4) * 4;  // the lane's four consecutive features inside the tile
      T sct[4], bvt[4];
      *reinterpret_cast<uint2*>(sct) = *reinterpret_cast<const uint2*>(tab + nl);
      *reinterpret_cast<uint2*>(bvt) = *reinterpret_cast<const uint2*>(tab + BN + nl);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sc[jj][r] = E::to_f32(sct[r]);   // 1.0 without a scale (the dense variant)
        bv[jj][r] = E::to_f32(bvt[r]);   // 0.0 without a bias (not added below)
      }
    }
#pragma unroll
    for (int h = 0; h < RH; ++h) {
#pragma unroll
      for (int jj = 0; jj < JP; ++jj) {
        const int j = p * JP + jj;
#pragma unroll
        for (int i = h * MIH; i < (h + 1) * MIH; ++i) {
          T out[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[j][i][r] * sc[jj][r];
            asm volatile("" : "+v"(v));  // product rounded to fp32 first, with and without bias (no single-rounding v_fma_mixlo_f16)
            if (has_bias) v = E::to_f32(E::from_f32(v)) + bv[jj][r];
            out[r] = E::from_f32(v);
          }
          const int row = i * 16 + (lane & 15);
          const int chunk = (jj * 4 + (lane >> 4)) ^ ((row & (LPR - 1)) << 1);
          *reinterpret_cast<uint2*>(park + row * ROWB + chunk * 8) = *reinterpret_cast<const uint2*>(out);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: no barrier needed
#pragma unroll
      for (int t = h * (MIH * 16 * LPR / 64); t < (h + 1) * (MIH * 16 * LPR / 64); ++t) {
        const int row = t * (64 / LPR) + lane / LPR;
        const int c16 = lane % LPR;
        const uint4 v = *reinterpret_cast<const uint4*>(park + row * ROWB + (((c16 * 2) ^ ((row & (LPR - 1)) << 1)) * 8));
        const int m = m0 + wm * (MI * 16) + row;
        const int n = n0 + wn * (NJ * 16) + p * (JP * 16) + c16 * 8;
        if (full) {
          // non-temporal: the 2*M*N output bytes are not re-read by this kernel; streaming them past the L2 shortens the
          // end-of-kernel write-back (measured: 19.4 -> 13.3 us at K = 128, -3 us at K = 4096)
          typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
          __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(yg + (size_t)m * N + n));
        } else if (m < M) {
          const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (n + r < N) yg[(size_t)m * N + n + r] = e[r];
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next pass parks into the rows this one has just read
  }
  QH_LT_STAMP(5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  QH_LT_STAMP(6);
}

enum { CFG_256_8W = 0, CFG_256_4W = 1, CFG_128_4W = 2, CFG_256_1X8 = 3 };

template <int DT, int FMT, int BM, int BN, int WM, int WN, int WD = 0>
static int launch_cfg(const Args& a, hipStream_t stream) {
  constexpr int lds0 = WD != 0 ? WD * (BM * BK * 2) : STAGES * (BM * BK * 2 + BN * BK);
  const int pad = env_int("QUANTO_HIP_LARGE_LDS_PAD", 0);  // experiments
  const int lds = lds0 + 2 * BN * 2 + pad;  // ring + the tile's scale / bias table
  static_assert(lds0 >= BM * BN * 2, "the epilogue parks the output tile in the stage memory");
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN, tiles = tiles_m * tiles_n;
  Args b = a;
  {
    // per-XCD band of B tiles as a (g x B/g) rectangle: fetched bytes ~ g*BM*2 + (B/g)*BN per k -> g = sqrt(B*BN/(2*BM))
    const int band = (tiles + 7) / 8;
    int g = 1;
    while ((g + 1) * (g + 1) * 2 * BM <= band * BN) ++g;
    const int forced = env_int("QUANTO_HIP_GROUP_M", 0);  // experiments
    if (forced > 0) g = forced;
    b.group_m = g < tiles_m ? g : tiles_m;
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbytes_mfma_large_kernel<DT, FMT, BM, BN, WM, WN, WD>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((qbytes_mfma_large_kernel<DT, FMT, BM, BN, WM, WN, WD>), dim3(tiles * b.S), dim3(WM * WN * 64), lds, stream, b);
  return launch_status();
}

template <int DT, int FMT>
static int launch(const Args& a, int cfg, hipStream_t stream) {
  // three 24 KiB stages: two workgroups share a CU (two interleaving streams per SIMD)
  if (cfg == CFG_128_4W) {
    const int wd = env_int("QUANTO_HIP_LARGE_WD", 1);  // 0 off, 1 auto, 5 always
    const int nk = a.K / BK / a.S;
    // weights-direct variant when the K-range of a workgroup is a whole number of 4-tile ring turns
    // and every workgroup has a CU to itself: with two workgroups per CU the partner hides the DMA latency anyway and
    // the LDS-weight loop is faster (bf16 x int8, us, LDS -> WD: 512x8192x8192 77 -> 61, 384x8192x8192 74 -> 57,
    // 1024x4096x4096 44 -> 33; but 1024x8192x4096 as 512 workgroups 68 -> 75, 768x6144x4096 as 288 56 -> 60)
    const int tiles = ((a.M + 127) / 128) * ((a.N + 127) / 128);
    // Ring depth 4.  A ring of 8 (128 KiB of activation stages, six tiles in flight) is NOT faster: (512,8192,8192) 59.7 ->
    // 61.5 us, (256,4096,4096) 27.1 -> 29.0 us - the lone workgroup is not latency bound any more but limited by what one CU
    // pulls through its vector L1: 24 KiB per K-tile in 0.41 us = ~27 B/clk, the same rate the dense variant (32 KiB per
    // K-tile, 0.64 us) and two paired workgroups (48 KiB, 1.06 us) reach.
    if ((wd & 1) && (tiles * a.S <= 256 || (wd & 4)) && nk % 4 == 0 && nk >= 8) return launch_cfg<DT, FMT, 128, 128, 1, 4, 4>(a, stream);
    return launch_cfg<DT, FMT, 128, 128, 1, 4>(a, stream);
  }
  // (tried: 128-tiles as eight waves of 128 x 16 with one workgroup per CU - LDS-bound, cfg4 90-98 us vs 86-90 us;
  //  r3: 128 tokens x 256 features as eight weights-direct waves of 128 x 32 - a third fewer fetched bytes per flop - with K split 2 ways
  //  so that cfg4 still covers 256 CUs: 68.2 us against 59.7 on the same box (unsplit on 128 CUs 86 us: 64 % MFMA utilisation instead
  //  of 50 %, but the partial-tile pass of 32 MB costs more than the better loop gains))
  // r6: the product library carries what the dispatch can reach.  The 2 x 2 layout (four waves of 128 x 128, hipcc-allocated AGPR accumulators:
  // 15-98 spilled VGPRs, 159.9 us on cfg2 against 102.4, profiles/r05_cfg2_wave_layouts_ab.jsonl) is built by probes only (-DQH_LT_EXPERIMENTS,
  // scripts/probes/README.md); e4m3fnuz exists as 1 x 8 only - its patched-pattern converter spills 25-33 VGPRs in the 2 x 4 layout
#ifdef QH_LT_EXPERIMENTS
  if (cfg == CFG_256_4W) return launch_cfg<DT, FMT, 256, 256, 2, 2>(a, stream);
#else
  if (cfg == CFG_256_4W) return QUANTO_HIP_ENOTSUP;
#endif
  if constexpr (FMT == W_F8E4M3FNUZ) {
    return launch_cfg<DT, FMT, 256, 256, 1, 8>(a, stream);
  } else {
    if (cfg == CFG_256_1X8) return launch_cfg<DT, FMT, 256, 256, 1, 8>(a, stream);
    return launch_cfg<DT, FMT, 256, 256, 2, 4>(a, stream);
  }
}

}  // namespace lt

bool qbytes_mfma_large_supported(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  const bool bd = b_dtype == QUANTO_HIP_I8 || b_dtype == QUANTO_HIP_F8_E4M3FN || b_dtype == QUANTO_HIP_F8_E5M2 || b_dtype == QUANTO_HIP_F8_E4M3FNUZ;
  return bd && a_dtype == out_dtype && (out_dtype == QUANTO_HIP_BF16 || out_dtype == QUANTO_HIP_F16) && K % lt::BK == 0 &&
         K >= 2 * lt::BK && M >= 1 && M * K < (1ll << 30) && N * K < (1ll << 31) && M < (1 << 30) && N < (1 << 30);
}

// split-K for the 128-tile configuration when its tiles cover at most half of the CUs.  The partial sums cost 64 KiB of
// system-coherent traffic per workgroup each way.  r3, with the partial tiles laid out fragment-major (whole lines per store
// instruction; lane-major wrote 16 bytes into every second line and made a split cost ~12 us + 1 us per MB), bf16 x int8, us with
// split 1 / 2 / 4: (128,4096,4096) 27.2 / 21.7 / 20.4, (256,4096,4096) 27.5 / 21.9 / 22.9, (512,4096,4096) 28.1 / 25.6 / 34.8,
// (256,8192,8192) 49.8 / 38.5 / 46.7, (512,4096,14336) 85.2 / 59.0 / 67.6; with more tiles than half the CUs it loses:
// (1024,4096,4096) 32.3 / 43.3, cfg4 (512,8192,8192; 256 tiles) 55.7 / 63.7 (r2 layout: 65 / 100).
static int large_split(int64_t M, int64_t N, int64_t K) {
  const int forced = env_int("QUANTO_HIP_LARGE_SPLIT", 0);  // experiments
  const int64_t tiles256 = ((M + 255) / 256) * ((N + 255) / 256), tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
  int s = 1;
  // halves (quarters for a single row of tiles) of a whole number of 4-tile ring turns (weights-direct loop)
  if (tiles128 <= 128 && (K / lt::BK) % 8 == 0 && K / lt::BK >= 16) s = 2;
  if (tiles128 <= 32 && (K / lt::BK) % 16 == 0 && K / lt::BK >= 32) s = 4;
  if (forced > 0 && tiles128 <= 512 && (K / lt::BK) % forced == 0 && K / lt::BK / forced >= 2) s = forced;
  (void)tiles256;
  if ((size_t)tiles128 * 4 > QUANTO_HIP_WS_COUNTER_BYTES) s = 1;  // one counter per 128-tile
  return s;
}
// fixed-size counter region shared by all split-K kernels of the library (see qbits_skinny.hip)
static size_t large_counter_bytes(int64_t, int64_t) { return QUANTO_HIP_WS_COUNTER_BYTES; }
size_t qbytes_mfma_large_workspace(int64_t M, int64_t N, int64_t K) {
  const int S = large_split(M, N, K);
  if (S == 1) return 0;
  return large_counter_bytes(M, N) + (size_t)(((M + 127) / 128) * ((N + 127) / 128)) * S * (128 * 128 * 4);
}

// Dense 16-bit GEMM y = x @ w^T (+ bias) on the weights-direct 128-tile loop, for grids that leave every workgroup a CU of its
// own.  Used by qbits_mm's dequantize + GEMM path; the gain over qmm_native8.hip's 128-tile dense kernel is small
// ((256, 4096, 4096): 47 -> 41 us for the GEMM, 58 -> 54 us with the dequantize pass): both stream 32 KiB per K-tile and CU.
bool dense_mm_wd_supported(int64_t M, int64_t N, int64_t K, int dtype) {
  const int on = env_int("QUANTO_HIP_DENSE_WD", 1);  // experiments
  const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
  return on && (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) && K % (4 * lt::BK) == 0 && K >= 8 * lt::BK && M >= 1 &&
         tiles128 <= 256 && M * K < (1ll << 30) && N * K < (1ll << 30);
}

int dense_mm_wd(const void* x, const void* w, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int dtype, hipStream_t stream) {
  if (!dense_mm_wd_supported(M, N, K, dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16) return QUANTO_HIP_EALIGN;
  lt::Args a{x, reinterpret_cast<const uint8_t*>(w), nullptr, bias, y, (int)M, (int)N, (int)K, 1, 1, nullptr, nullptr};
  if (dtype == QUANTO_HIP_BF16) return lt::launch_cfg<QUANTO_HIP_BF16, lt::W_DENSE, 128, 128, 1, 4, 4>(a, stream);
  return lt::launch_cfg<QUANTO_HIP_F16, lt::W_DENSE, 128, 128, 1, 4, 4>(a, stream);
}

int qbytes_mm_mfma_large(const void* x, const void* w, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int a_dtype,
                      int b_dtype, int out_dtype, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!qbytes_mfma_large_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  int split = large_split(M, N, K);
  if (split > 1 && (!workspace || workspace_bytes < qbytes_mfma_large_workspace(M, N, K) || reinterpret_cast<uintptr_t>(workspace) % 16)) split = 1;
  // 128-tiles as long as all of them are resident at once (two workgroups per CU: 512), 256-tiles beyond.  Measured,
  // bf16 x int8, K = 4096, us with 256-tiles -> 128-tiles: (512,14336) 87 -> 67, (1024,8192) 85 -> 70, (2048,4096) 79 -> 66;
  // but (1280,8192) 82 -> 99, (2048,8192) 107 -> 125
  const int forced = env_int("QUANTO_HIP_LARGE_CFG", -1);  // experiments
  const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
  // e4m3fnuz on 256-tiles: the 1 x 8 layout - the patched-pattern converter (qh_common.h) spills in the 2 x 4 layout (104-116 B of scratch per
  // lane; 0 in 1 x 8, which measures within 2 % of 2 x 4 on the other formats: profiles/r05_cfg2_wave_layouts_ab.jsonl)
  const int cfg256 = b_dtype == QUANTO_HIP_F8_E4M3FNUZ ? lt::CFG_256_1X8 : lt::CFG_256_8W;
  const int cfg = forced >= 0 ? forced : (tiles128 > 512 ? cfg256 : lt::CFG_128_4W);
  if (cfg != lt::CFG_128_4W) split = 1;  // the workspace is sized for 128-tiles
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16) return QUANTO_HIP_EALIGN;
  lt::Args a{x, reinterpret_cast<const uint8_t*>(w), s, bias, y, (int)M, (int)N, (int)K, 1, split, reinterpret_cast<int*>(workspace),
             split > 1 ? reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(workspace) + large_counter_bytes(M, N)) : nullptr};
#define QH_CASE(DT, FMT) return lt::launch<DT, FMT>(a, cfg, stream)
  if (out_dtype == QUANTO_HIP_BF16) {
    if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_BF16, lt::W_I8);
    if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_BF16, lt::W_F8E4M3);
    if (b_dtype == QUANTO_HIP_F8_E4M3FNUZ) QH_CASE(QUANTO_HIP_BF16, lt::W_F8E4M3FNUZ);
    QH_CASE(QUANTO_HIP_BF16, lt::W_F8E5M2);
  }
  if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_F16, lt::W_I8);
  if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_F16, lt::W_F8E4M3);
  if (b_dtype == QUANTO_HIP_F8_E4M3FNUZ) QH_CASE(QUANTO_HIP_F16, lt::W_F8E4M3FNUZ);
  QH_CASE(QUANTO_HIP_F16, lt::W_F8E5M2);
#undef QH_CASE
}

}  // namespace qh
