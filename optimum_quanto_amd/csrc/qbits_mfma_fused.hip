// qbits_mm, fused int4 GEMM for prefill-sized M: packed nibbles -> MFMA operands in registers, scale / shift folded into the fp32
// accumulator per group, no dequantized weight anywhere (neither HBM nor LDS).
//
//   y[m, n] = sum_g ( s[n,g] * sum_{k in g} x[m,k] * (128 + q[n,k])  -  (z[n,g] + 128 s[n,g]) * XS[m,g] ),   XS[m,g] = sum_{k in g} x[m,k]
//
// exact products of the stored integers (0x4300 | q is exactly 128 + q in bf16, 0x6400 | q is 1024 + q in fp16), fp32 accumulation,
// scale and shift applied in fp32 - the same arithmetic as the decode kernels (qbits_gemv.hip, qbits_skinny.hip), so the oracle
// is exact math on the reference's integers / scales, not the reference's twice-rounded bf16 weight.  The reference analogs are the
// CUDA-only AWQ / Marlin GEMMs (library/extensions/cuda/awq/v2/gemm_cuda.cu, cuda/marlin/marlin_cuda_kernel.cu), which dequantize
// in registers with one fp16 rounding per weight.
//
// Where it pays.  dequantize + dense GEMM (c_api.hip, DEQUANT_MFMA) writes N*K*2 bytes and makes the GEMM read 4x the packed
// bytes; for M up to ~1 k the dense GEMM is bound by the bytes a CU pulls per K-tile, not by the matrix pipe.  This kernel streams
// the packed bytes (a quarter of the int8 kernel's weight traffic) and pays VALU instead: the operands (v_perm + v_and_or) and the
// per-group fold (one scaled group accumulator per output) are three times the int8 kernel's VALU work per MFMA, so at large M,
// where the dequantize pass is amortised over thousands of rows, the dense path wins again (dispatch in c_api.hip, measured).
//
// Structure: workgroup = 8 waves as 2 x 4 = 128 tokens x 64 packed rows (= 64 features of the low plane + the 64 features N/2
// further of the high plane), wave = 64 tokens x 32 features; K-tile = one group (128 k).  Activations AND weights travel by LDS-DMA
// into a ring of three 40 KiB stages, two tiles ahead (256-byte activation rows, chunk ^ (row & 15) on the DMA source, undone on the
// read; 128-byte weight rows, chunk ^ (row & 7)).  Scales / shifts of the workgroup's 128 features for its groups are parked in LDS
// once.  Split-K (r2) where the table of all groups does not fit (K = 14336): the groups are split over 2 or 4 workgroups per tile,
// fp32 partial tiles go through the workspace (write-through stores, arrival counter, the last workgroup adds them in split order -
// the protocol of qbits_skinny.hip).  It does not pay as a way to fill idle CUs (pick_split below).
#include <type_traits>

#include "qh_common.h"

namespace qh {
namespace fused4 {

// Workgroup: 8 waves as 2 (token halves) x 4 (blocks of 16 packed rows) = BM tokens x 64 packed rows (128 features), one per CU
// (two waves per SIMD).  Wave: BM/2 tokens x 32 features = MI x 2 accumulator blocks (running + group).
//   BM = 128 (MI = 4): 32 KiB of activations per 8 KiB of packed weights and tile - the form for grids that fill the chip;
//   BM = 64  (MI = 2): r3 - twice the workgroups for short prefills whose 128-token tiles leave CUs idle ((512,4096,4096): 128 ->
//                      256 workgroups), 24 KiB per tile.
constexpr int BK = 128, PR = 64, WAVES = 8, DEPTH = 2, STAGES = 3;
constexpr int W_BYTES = PR * BK;  // 8 KiB
template <int BM>
struct Geo {
  static constexpr int MI = BM / 32;            // 16-token fragments per wave
  static constexpr int X_BYTES = BM * BK * 2, STAGE_BYTES = X_BYTES + W_BYTES;
  static constexpr int XP = BM / 4 / WAVES;     // activation DMA pieces (4 rows x 256 B = 1 KiB) per wave and tile; the weight tile is one piece per wave
  static constexpr int OPS = XP + 1;            // vector-memory instructions per wave and tile: activation pieces + weight piece
  static_assert(W_BYTES == WAVES * 1024 && DEPTH * OPS <= 63 && (MI == 2 || MI == 4), "tile geometry");
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// LDS-DMA, 16 bytes per lane: wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset
__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

template <int DT>
struct Mma;
template <>
struct Mma<QUANTO_HIP_BF16> {
  using V8 = bf16x8;
  static __device__ __forceinline__ f32x4 run(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static constexpr uint32_t MAGIC = 0x43004300u;
  static constexpr float OFFSET = 128.f;
};
template <>
struct Mma<QUANTO_HIP_F16> {
  using V8 = f16x8;
  static __device__ __forceinline__ f32x4 run(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static constexpr uint32_t MAGIC = 0x64006400u;
  static constexpr float OFFSET = 1024.f;
};

template <int DT>
__device__ __forceinline__ uint32_t ONE2() { return DT == QUANTO_HIP_BF16 ? 0x3F803F80u : 0x3C003C00u; }  // (1.0, 1.0)

struct Args {
  const void* x;       // [M, K]
  const uint8_t* w;    // packed [N/2, K]
  const void* scale;   // [N*G]
  const void* shift;   // [N*G]
  const void* bias;    // [N] or null
  void* y;             // [M, N]
  int M, N, K, G;
  int S;               // K split: blockIdx.z handles groups [z * G / S, (z + 1) * G / S)
  int* counters;       // [tiles] arrival counters, zero on entry and on exit (S > 1)
  float* partials;     // [tiles][S][512 lanes][2 * MI] float4
};

template <int DT, bool INT_SHIFT>
__global__ void __launch_bounds__(WAVES * 64, 1) qbits_mfma_fused_kernel(const Args a) {
  using E = Elem<DT>;
  using T = typename E::T;
  using V8 = typename Mma<DT>::V8;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // layout: [STAGES x (activation tile | weight tile)] [sz: G x 2 x 128 features of T]
  T* sz = reinterpret_cast<T*>(smem + STAGES * STAGE_BYTES);
  constexpr int NF = 2 * PR;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int M = a.M, N = a.N, K = a.K, G = a.G;
  const int P = N >> 1;
  const int p0 = blockIdx.x * PR, m0 = blockIdx.y * BM;
  const int S = a.S, sp = blockIdx.z;
  const int nk = G / S;     // one tile per group; this workgroup's groups are kt0 .. kt0 + nk - 1
  const int kt0 = sp * nk;
  const int fi = lane & 15, fg = lane >> 4;

  // ---- memory pipeline.  A tile lasts under a microsecond, a load from L2 / HBM under load 1-2 us: activations AND weights travel
  // by LDS-DMA into a ring of three stages, DEPTH = 2 tiles ahead.  The loop contains no other vector-memory instruction, so the
  // hand-counted s_waitcnt below is the only wait on that queue (hipcc counts only the loads it can see and would drain the DMA
  // queue at each of its own waits).  Every tile issues the same OPS DMA instructions; tiles past the end re-request the last tile
  // (harmless), so the count never changes.  XS[m, g] needs no memory at all: one extra MFMA per step against an all-ones operand
  // accumulates sum_k x[m, k] in the same order and with the same roundings as the products it corrects - measured necessary for
  // fp16, where the 1024 offset leaves only ~13 bits of the fp32 accumulator for the signal: with XS from a separately ordered sum
  // (a pre-kernel) 20 of 51 k outputs missed the 2-ulp gate; the matrix pipe has the slack (the loop is VALU-bound).
  uint32_t xsrc[XP];  // byte offsets from a.x (M * K * 2 < 4 GiB, checked by the launcher)
#pragma unroll
  for (int u = 0; u < XP; ++u) {
    const int row = 4 * (wave * XP + u) + (lane >> 4);
    const int c = (lane & 15) ^ (row & 15);
    int m = m0 + row;
    m = m < M ? m : M - 1;
    xsrc[u] = (uint32_t)(((size_t)m * K + c * 8) * 2);
  }
  uint32_t wsrc;  // this wave's weight piece: packed rows 8*wave .. +7 of the tile, lane -> row lane>>3, position lane&7 holds chunk pos ^ (row & 7)
  {
    const int r = wave * 8 + (lane >> 3), c = (lane & 7) ^ (r & 7);
    int p = p0 + r;
    p = p < P ? p : P - 1;
    wsrc = (uint32_t)((size_t)p * K + c * 16);  // (N/2) * K < 4 GiB, checked by the launcher
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.x);
  auto issue_tile = [&](int kt_tile, int stage) {  // activation pieces + weight piece of one tile: XP + 1 DMA instructions
    const uint32_t st = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE_BYTES);
#pragma unroll
    for (int u = 0; u < XP; ++u) glds16(xbase + (size_t)(kt0 + kt_tile) * (BK * 2), xsrc[u], st + (wave * XP + u) * 1024);
    glds16(a.w + (size_t)(kt0 + kt_tile) * BK, wsrc, st + X_BYTES + wave * 1024);
  };
  const int last = nk - 1;

  // ---- prologue: tiles 0 and 1 requested; tables parked ------------------------------------------------------------------------------
  issue_tile(0, 0);
  issue_tile(nk > 1 ? 1 : 0, 1);
  {
    // thread -> feature tid & 127 (plane = bit 6), groups (tid >> 7), +4, ...: no division in front of the loop
    const int f = tid & (NF - 1);
    int p = p0 + (f & (PR - 1));
    p = p < P ? p : P - 1;
    const size_t row = (size_t)(p + (f >> 6) * P) * G + kt0;
    for (int g = tid >> 7; g < nk; g += (WAVES * 64) >> 7) {
      sz[(g * 2 + 0) * NF + f] = reinterpret_cast<const T*>(a.scale)[row + g];
      if constexpr (INT_SHIFT)
        sz[(g * 2 + 1) * NF + f] = E::from_f32((float)(int8_t) reinterpret_cast<const uint8_t*>(a.shift)[row + g]);
      else
        sz[(g * 2 + 1) * NF + f] = reinterpret_cast<const T*>(a.shift)[row + g];
    }
  }
  // the table loop contains compiler-visible loads with compiler-placed waits: drain once so that the hand-counted waits below
  // start from a known state (tiles 0 and 1 have landed, nothing outstanding)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // ---- fragment read offsets --------------------------------------------------------------------------------------------------------
  // activations: chunk of k-step t for lane group g is 8 (t >> 1) + 2 g + (t & 1) (the weight bytes the lane holds: k = 16 g +
  // 8 (t & 1) .. for t < 2, 64 + 16 g + 8 (t & 1) .. for t >= 2); rows wm*64 + i*16 + fi, so (row & 15) == fi for every fragment
  int xoff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) xoff[t] = (wm * 64 + fi) * 256 + (((8 * (t >> 1) + 2 * fg + (t & 1)) ^ fi) << 4);
  // weights: 16-byte chunks fg and 4 + fg of packed row wn*16 + fi of the tile (128-byte rows, chunk ^ (row & 7))
  int woff[2];
  {
    const int r = wn * 16 + fi;
#pragma unroll
    for (int h = 0; h < 2; ++h) woff[h] = X_BYTES + r * 128 + (((4 * h + fg) ^ (r & 7)) << 4);
  }
  // this lane's 4 consecutive features inside the block: plane j, local packed rows wn*16 + 4*fg + r
  const int floc = wn * 16 + 4 * fg;

  f32x4 acc[2][MI];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint32_t kmask = 0x000F000Fu, kmagic = Mma<DT>::MAGIC;
  asm volatile("" : "+s"(kmask));
  asm volatile("" : "+v"(kmagic));
  const V8 ones = __builtin_bit_cast(V8, make_uint4(ONE2<DT>(), ONE2<DT>(), ONE2<DT>(), ONE2<DT>()));

  // Group accumulators are double-buffered: while tile kt accumulates into one set, the fold of tile kt-1 (scale / shift applied to
  // the other set, 64 VALU + the table reads) is sliced over the 16 MFMA steps of tile kt.  Without this the two waves of a SIMD,
  // re-synchronised by the barrier of every tile, run their MFMA phases together and then their fold phases together, and the
  // matrix pipe and the VALU take turns idling (measured: 1.56 us per tile instead of ~0.8).
  f32x4 accgA[2][MI], accxA[MI], accgB[2][MI], accxB[MI];
  float s4[2][4], z4[2][4];  // scale and (shift + OFFSET * scale) of the lane's 2 x 4 features for the group being folded
  auto load_sz = [&](int g) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      T s4t[4], z4t[4];
      *reinterpret_cast<uint2*>(s4t) = *reinterpret_cast<const uint2*>(sz + (g * 2 + 0) * NF + j * PR + floc);
      *reinterpret_cast<uint2*>(z4t) = *reinterpret_cast<const uint2*>(sz + (g * 2 + 1) * NF + j * PR + floc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s4[j][r] = E::to_f32(s4t[r]);
        const float z = E::to_f32(z4t[r]);
        z4[j][r] = INT_SHIFT ? s4[j][r] * (z + Mma<DT>::OFFSET) : z + Mma<DT>::OFFSET * s4[j][r];
      }
    }
  };
  // slice q (0..15) of the fold of one group: features r = 2 (q & 1), +1 of block (j = q >> 3, i = (q >> 1) & 3)
  auto fold_slice = [&](const f32x4 (&pg)[2][MI], const f32x4 (&px)[MI], int q) {
    const int j = q >> 3, i = (q >> 1) & 3, r0 = (q & 1) * 2;
#pragma unroll
    for (int r = r0; r < r0 + 2; ++r) acc[j][i][r] += s4[j][r] * pg[j][i][r] - z4[j][r] * px[i][0];  // every row of the ones-product holds XS
  };

  int stage = 0;
  // One tile (= one group) accumulating into (cg, cx) while the previous tile's (pg, px) is folded.
  auto tile = [&](int kt, f32x4 (&cg)[2][MI], f32x4 (&cx)[MI], const f32x4 (&pg)[2][MI], const f32x4 (&px)[MI], bool have_prev) {
    // requests of tile kt (issued two tiles ago) have landed once at most the one younger group is outstanding.  (Tiles 0 and 1:
    // completed by the prologue; fewer groups are outstanding than the count allows, the wait falls through.)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * OPS) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the fragment / table reads of the previous tile
    __builtin_amdgcn_s_barrier();  // tile kt visible to all; everybody is done with tile kt-1, whose stage is refilled now
    asm volatile("" ::: "memory");
    {
      const int tn = kt + DEPTH < nk ? kt + DEPTH : last;
      const int sn = stage + DEPTH >= STAGES ? stage + DEPTH - STAGES : stage + DEPTH;
      issue_tile(tn, sn);
    }
    const uint8_t* st = smem + stage * STAGE_BYTES;
    uint4 w[2];
    w[0] = *reinterpret_cast<const uint4*>(st + woff[0]);
    w[1] = *reinterpret_cast<const uint4*>(st + woff[1]);
    if (have_prev) load_sz(kt - 1);

    // 4 * MI steps (k-step t, token fragment i) of three MFMAs (low plane, high plane, ones -> XS); behind them a slice of the NEXT
    // k-step's operand conversion, a slice of the previous group's fold and the activation fragment of step s+2.  The sched_barrier
    // pins that order: left alone, hipcc hoists all fragment reads to the top of the tile and spills.
    uint32_t lo[2][4], hi[2][4];
    auto raw = [&](int t, int d) -> uint32_t {  // dword d (0, 1) of the 8 weight bytes of k-step t
      const uint4& q = w[t >> 1];
      return (t & 1) ? (d ? q.w : q.z) : (d ? q.y : q.x);
    };
    auto convert = [&](int t, int c) {  // operand dword c (0..3 low plane, 4..7 high plane) of k-step t
      const uint32_t src = raw(t, (c & 3) >> 1) >> (c >= 4 ? 4 : 0);
      const uint32_t v = (__builtin_amdgcn_perm(0u, src, (c & 1) ? 0x0C030C02u : 0x0C010C00u) & kmask) | kmagic;
      if (c < 4)
        lo[t & 1][c] = v;
      else
        hi[t & 1][c - 4] = v;
    };
#pragma unroll
    for (int c = 0; c < 8; ++c) convert(0, c);
    V8 xf[3];
    xf[0] = *reinterpret_cast<const V8*>(st + xoff[0]);
    xf[1] = *reinterpret_cast<const V8*>(st + xoff[0] + 4096);
#pragma unroll
    for (int s = 0; s < 4 * MI; ++s) {
      const int t = s / MI, i = s % MI;
      const V8 wl = __builtin_bit_cast(V8, make_uint4(lo[t & 1][0], lo[t & 1][1], lo[t & 1][2], lo[t & 1][3]));
      const V8 wh = __builtin_bit_cast(V8, make_uint4(hi[t & 1][0], hi[t & 1][1], hi[t & 1][2], hi[t & 1][3]));
      cg[0][i] = Mma<DT>::run(wl, xf[s % 3], t == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : cg[0][i]);
      if (t < 3) {
        // the 8 operand dwords of k-step t+1, spread over the MI steps of k-step t
#pragma unroll
        for (int c = i * 8 / MI; c < (i + 1) * 8 / MI; ++c) convert(t + 1, c);
      }
      cg[1][i] = Mma<DT>::run(wh, xf[s % 3], t == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : cg[1][i]);
      if (have_prev) fold_slice(pg, px, s);
      cx[i] = Mma<DT>::run(ones, xf[s % 3], t == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : cx[i]);
      if (s + 2 < 4 * MI) xf[(s + 2) % 3] = *reinterpret_cast<const V8*>(st + xoff[(s + 2) / MI] + ((s + 2) % MI) * 4096);
      __builtin_amdgcn_sched_barrier(0);
    }
    stage = stage + 1 == STAGES ? 0 : stage + 1;
  };
  static_assert(MI == 4, "fold_slice maps 16 slices onto 2 x MI x 2 register pairs");
  tile(0, accgA, accxA, accgB, accxB, false);
  int kt = 1;
  for (; kt + 2 <= nk; kt += 2) {
    tile(kt, accgB, accxB, accgA, accxA, true);
    tile(kt + 1, accgA, accxA, accgB, accxB, true);
  }
  if (kt < nk) {
    tile(kt, accgB, accxB, accgA, accxA, true);  // nk even: the last tile landed in set B
    load_sz(nk - 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) fold_slice(accgB, accxB, q);
  } else {
    load_sz(nk - 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) fold_slice(accgA, accxA, q);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-requested tiles past the end: nothing may land in LDS after the kernel moved on

  // ---- split-K: fp32 partial tiles through the workspace, the last workgroup of a tile adds them in split order (qbits_skinny.hip) ----
  if (S > 1) {
    const int tile_id = blockIdx.y * gridDim.x + blockIdx.x;
    float* mine = a.partials + ((size_t)(tile_id * S + sp) * (WAVES * 64) + tid) * (2 * MI * 4);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i)  // s_nop: gfx9 hazard "VMEM store of > 64 bits, then VALU write of its data VGPRs"
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(mine + (j * MI + i) * 4), "v"(acc[j][i]) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) *flag = __hip_atomic_fetch_add(a.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (*flag != S - 1) return;
    if (tid == 0) __hip_atomic_store(a.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // leave the workspace as found
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < S; ++q) {  // fixed order: the result does not depend on which workgroup arrived last
      const float* theirs = a.partials + ((size_t)(tile_id * S + q) * (WAVES * 64) + tid) * (2 * MI * 4);
      f32x4 v[2 * MI];
#pragma unroll
      for (int e = 0; e < 2 * MI; ++e) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[e]) : "v"(theirs + e * 4) : "memory");
#pragma unroll
      for (int e = 0; e < 2 * MI; ++e) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[e])::"memory");  // ties the uses below to the wait
#pragma unroll
      for (int e = 0; e < 2 * MI; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[e / MI][e % MI][r] += v[e][r];
    }
  }

  // ---- epilogue: 4 consecutive features of one token per (plane, fragment): 8-byte stores --------------------------------------------
  T* yg = reinterpret_cast<T*>(a.y);
  const bool has_bias = a.bias != nullptr;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pl = p0 + floc;            // first of the lane's 4 packed rows
    const int n0 = pl + j * P;           // 4 consecutive output features n0 .. n0+3
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (has_bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = pl + r < P ? E::to_f32(reinterpret_cast<const T*>(a.bias)[n0 + r]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fi;
      if (m < M) {
        T out[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[j][i][r];
          if (has_bias) v = E::to_f32(E::from_f32(v)) + bv[r];
          out[r] = E::from_f32(v);
        }
        if (pl + 3 < P && (N & 3) == 0) {
          *reinterpret_cast<uint2*>(yg + (size_t)m * N + n0) = *reinterpret_cast<const uint2*>(out);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (pl + r < P) yg[(size_t)m * N + n0 + r] = out[r];
        }
      }
    }
  }
}

inline int lds_bytes(int groups) { return STAGES * STAGE_BYTES + groups * 2 * (2 * PR) * 2; }

inline int tiles_of(int64_t M, int64_t N) { return (int)(((N / 2 + PR - 1) / PR) * ((M + BM - 1) / BM)); }

// K split.  Measured (us, unsplit / 2 / 4): (512,4096,4096) 49.7 / 52.6 / 80.8 - a split costs a 64 KiB fp32 partial tile per
// workgroup through the fabric and back, which eats what the idle CUs would give - so K is split only where the scale table of
// all groups does not fit the LDS (K = 14336), and then as far as the tiles leave CUs idle: (256,14336,4096) 76.9 us with 4 splits
// against 153 us for dequantize + dense GEMM, (512,14336,4096) 115.5 against 155.
inline int pick_split(int64_t M, int64_t N, int G) {
  const int forced = env_int("QUANTO_HIP_FUSED4_SPLIT", 0);
  if (forced > 0 && G % forced == 0) return forced;
  if (lds_bytes(G) <= 160 * 1024) return 1;
  const int tiles = tiles_of(M, N);
  int s = 1;
  while (s < 4 && G % (s * 2) == 0 && (lds_bytes(G / s) > 160 * 1024 || tiles * s * 2 <= 256)) s *= 2;
  if ((size_t)tiles * 4 > QUANTO_HIP_WS_COUNTER_BYTES) s = 1;
  return s;
}

template <int DT, bool INT_SHIFT>
static int launch(const Args& a, hipStream_t stream) {
  const int lds = lds_bytes(a.G / a.S);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbits_mfma_fused_kernel<DT, INT_SHIFT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const dim3 grid((unsigned)((a.N / 2 + PR - 1) / PR), (unsigned)((a.M + BM - 1) / BM), (unsigned)a.S);
  hipLaunchKernelGGL((qbits_mfma_fused_kernel<DT, INT_SHIFT>), grid, dim3(WAVES * 64), lds, stream, a);
  return launch_status();
}

}  // namespace fused4

bool qbits_mfma_fused_supported(int64_t M, const PackedGeom& g, int dtype) {
  return g.bits == 4 && g.C == 128 && (g.N % 8 == 0) && (g.K % 128 == 0) && M >= 1 && (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) &&
         g.N < (1 << 30) && g.K < (1 << 30) && M * g.K < (1ll << 31) && g.N * g.K < (1ll << 33) &&
         fused4::lds_bytes((int)g.G / fused4::pick_split(M, g.N, (int)g.G)) <= 160 * 1024;
}

// true when the problem cannot run without the split-K scratch (the scale table of all groups does not fit the LDS: K = 14336)
bool qbits_mfma_fused_needs_workspace(const PackedGeom& g) { return fused4::lds_bytes((int)g.G) > 160 * 1024; }

// [counters (zero on entry, zero on exit) | fp32 partial tiles]; 0 when K is not split (the group sums of x come from the matrix pipe)
size_t qbits_mfma_fused_workspace(int64_t M, const PackedGeom& g) {
  const int S = fused4::pick_split(M, g.N, (int)g.G);
  if (S == 1) return 0;
  return QUANTO_HIP_WS_COUNTER_BYTES + (size_t)fused4::tiles_of(M, g.N) * S * (fused4::WAVES * 64) * (2 * fused4::MI * 16);
}

int qbits_mm_mfma_fused(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t M,
                        const PackedGeom& g, int dtype, bool int_shift, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!qbits_mfma_fused_supported(M, g, dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed)) % 16) return QUANTO_HIP_EALIGN;
  int S = fused4::pick_split(M, g.N, (int)g.G);
  if (S > 1 && (!workspace || workspace_bytes < qbits_mfma_fused_workspace(M, g) || reinterpret_cast<uintptr_t>(workspace) % 16)) {
    S = 1;  // no scratch: unsplit, if the whole scale table fits
    if (fused4::lds_bytes((int)g.G) > 160 * 1024) return QUANTO_HIP_EINVAL;
  }
  fused4::Args a{x, packed, scale, shift, bias, y, (int)M, (int)g.N, (int)g.K, (int)g.G, S, reinterpret_cast<int*>(workspace),
                 S > 1 ? reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(workspace) + QUANTO_HIP_WS_COUNTER_BYTES) : nullptr};
  if (dtype == QUANTO_HIP_BF16)
    return int_shift ? fused4::launch<QUANTO_HIP_BF16, true>(a, stream) : fused4::launch<QUANTO_HIP_BF16, false>(a, stream);
  return int_shift ? fused4::launch<QUANTO_HIP_F16, true>(a, stream) : fused4::launch<QUANTO_HIP_F16, false>(a, stream);
}

}  // namespace qh
