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
// Structure (r3): workgroup = 8 waves, BM tokens x 64 packed rows (= 64 features of the low plane + the 64 features N/2 further of
// the high plane); wave w = ALL BM tokens x 16 features (packed rows 8w .. 8w+7, both planes: lanes 0-7 of every 16 keep the low
// nibbles, lanes 8-15 the high nibbles of the same 8 rows - the decode kernels' mapping); K-tile = one group (128 k).  Every weight
// is turned into an MFMA operand exactly once per workgroup (r2's 2 x 4 layout built every operand in both token halves and spent
// 2.2 VALU per MFMA on it; here 8 VALU serve 4 k-steps x MI MFMAs).  Activations AND weights travel by LDS-DMA into a ring of two
// stages, one tile ahead (256-byte activation rows, chunk ^ (row & 15) on the DMA source, undone on the read; 128-byte weight rows,
// chunk ^ (row & 7)).  Scales / shifts of the workgroup's 128 features for its groups are parked in LDS once (16-byte loads, issued
// in front of the first tiles' DMA).  Split-K where the scale table does not fit (K = 14336 with 128-token tiles) and, with 64-token
// tiles, to fill the chip: fp32 partial tiles through the workspace, write-through stores, arrival counter, the last workgroup adds
// them in split order - the protocol of qbits_skinny.hip.
#include <type_traits>

#include "qh_common.h"

namespace qh {
namespace fused4 {

// Token tile: BM = 128 (MI = 8 fragments per wave) for grids that fill the chip - 32 KiB of activations per 8 KiB of packed weights
// and tile -, BM = 64 (MI = 4) for short prefills whose 128-token tiles would leave CUs idle ((512,4096,4096): 128 -> 256 workgroups).
// Ring of TWO stages, one tile ahead (r3, end of round; three stages / two ahead before): a 64-token workgroup then needs 65 KiB of LDS with
// the tables of 32 groups and TWO workgroups share a CU (the kernel's 106-124 VGPRs always allowed four waves per SIMD) - out of phase, they
// fill each other's barrier and wait slots: (1024,4096,4096) as 512 workgroups of 64 tokens 56.7 -> 49.5 us, (1536,...) 79.3 -> 73.3, (2048,...)
// 98 -> 92.6; grids of one workgroup per CU do not notice the shallower ring ((512,4096,4096) 29.4 -> 29.3, (128,...) unsplit 27.6 -> 26.1).
constexpr int BK = 128, PR = 64, WAVES = 8, DEPTH = 1, STAGES = 2;
constexpr int W_BYTES = PR * BK;  // 8 KiB
template <int BM>
struct Geo {
  static constexpr int MI = BM / 16;            // 16-token fragments (every wave owns all of them)
  static constexpr int X_BYTES = BM * BK * 2, STAGE_BYTES = X_BYTES + W_BYTES;
  static constexpr int XP = BM / 4 / WAVES;     // activation DMA pieces (4 rows x 256 B = 1 KiB) per wave and tile; the weight tile is one piece per wave
  static constexpr int OPS = XP + 1;            // vector-memory instructions per wave and tile: activation pieces + weight piece
  static_assert(W_BYTES == WAVES * 1024 && DEPTH * OPS <= 63 && (MI == 4 || MI == 8), "tile geometry");
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// LDS-DMA, 16 bytes per lane: wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset
__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  // M0 is written and not restored (qmm_large_common.h: nothing else in this kernel needs it; 2 SALU instructions per piece)
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
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
  float* partials;     // [tiles][S][MI][512 lanes] float4: every store / load instruction of a wave covers 1 KiB of whole lines
  // QUANTO_HIP_FUSED4_ABLATE (timing experiments, WRONG results): 1 at most two tiles of the K loop, 2 no output stores, 4 no table fill
  // (r3 also tried 8 = no DMA inside the K loop and 16 = no MFMA steps: (512,4096,4096) 38.8 -> 36.4 / 16.7 us - the step loop, not
  // the DMA, is what a tile waits for; those two knobs changed the code of the loop they were meant to measure and were removed)
  int ablate;
};

template <int DT, bool INT_SHIFT, int BM>
__global__ void __launch_bounds__(WAVES * 64, 1) qbits_mfma_fused_kernel(const Args a) {
  using E = Elem<DT>;
  using T = typename E::T;
  using V8 = typename Mma<DT>::V8;
  constexpr int MI = Geo<BM>::MI, XP = Geo<BM>::XP, OPS = Geo<BM>::OPS, X_BYTES = Geo<BM>::X_BYTES, STAGE_BYTES = Geo<BM>::STAGE_BYTES;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // r3, measured and dropped ("shift term in the tail"): for bf16 the K loop folded only acc += s * acc_g, kept XS[m, g] of all groups and
  // applied  - sum_g (z + 128 s)[n, g] * XS[m, g]  once, after the loop, as two K' = groups GEMMs on the matrix pipe (z, s: bf16 as stored;
  // XS split into three bf16 pieces by truncation: exact products).  Parity-green; the loop lost a third of its VALU work and 9 % of
  // its time (0.74 -> 0.67 us per 64-token tile), the tail cost 3.4 us per workgroup: (512,4096,4096) 28.5 -> 30.2 us.
  // layout: [STAGES x (activation tile | weight tile)] [xs: 2 x BM fp32 group sums of x] [sz: G x 2 x 128 features of T]
  float* xs_slot = reinterpret_cast<float*>(smem + STAGES * STAGE_BYTES);
  constexpr int NF = 2 * PR;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M, N = a.N, K = a.K, G = a.G;
  const int P = N >> 1;
  const int p0 = blockIdx.x * PR, m0 = blockIdx.y * BM;
  const int S = a.S, sp = blockIdx.z;
  const int nk = G / S;     // one tile per group; this workgroup's groups are kt0 .. kt0 + nk - 1
  const int nk_run = (a.ablate & 1) && nk > 2 ? 2 : nk;  // tiles the K loop runs (timing experiments only)
  const int kt0 = sp * nk;
  const int fi = lane & 15, fg = lane >> 4;
  T* sz = reinterpret_cast<T*>(smem + STAGES * STAGE_BYTES + 2 * BM * 4);

  // ---- memory pipeline.  A tile lasts under a microsecond, a load from L2 / HBM under load 1-2 us: activations AND weights travel
  // by LDS-DMA into a ring of STAGES stages, DEPTH tiles ahead.  The loop contains no other vector-memory instruction, so the
  // hand-counted s_waitcnt below is the only wait on that queue (hipcc counts only the loads it can see and would drain the DMA
  // queue at each of its own waits).  Every tile issues the same OPS DMA instructions; tiles past the end re-request the last tile
  // (harmless), so the count never changes.  XS[m, g] comes from the matrix pipe: an extra MFMA per k-step against an all-ones operand
  // accumulates sum_k x[m, k] in the same order and with the same roundings as the products it corrects - measured necessary for
  // fp16, where the 1024 offset leaves only ~13 bits of the fp32 accumulator for the signal: with XS from a separately ordered sum
  // (a pre-kernel) 20 of 51 k outputs missed the 2-ulp gate.  r3: the four waves that share a token half no longer compute it four
  // times (12 of a wave's 48 MFMAs per tile were these): wave (wm, wn) runs the ones-products of fragment wn only and leaves the sums
  // in LDS (xs_slot, double-buffered by tile parity) - the fold of a tile runs one tile later anyway, behind a barrier.
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

  // ---- prologue: tables requested, tiles 0 and 1 requested, ONE wait, tables parked ---------------------------------------------------
  // r3: the scale / shift entries of a feature are contiguous over the groups, so a thread fetches 8 groups of one feature with one
  // 16-byte load (r2: 2-byte loads, four groups apart per iteration, each iteration waiting for its own round trip: 8 dependent
  // round trips in front of every workgroup's first MFMA).  The loads go out BEFORE the first two tiles' DMA and everything is
  // waited for once.  Needs G % 8 == 0, a K-range of whole 8-group chunks and 16-byte aligned tables; otherwise the element loop.
  constexpr int FILL_IT = 4;  // up to 4 x 512 chunks of 8 groups = 128 features x 128 groups
  const int chunks = nk >> 3;
  const bool vec_fill = !INT_SHIFT && (G & 7) == 0 && (nk & 7) == 0 && NF * chunks <= FILL_IT * WAVES * 64 &&
                        ((reinterpret_cast<uintptr_t>(a.scale) | reinterpret_cast<uintptr_t>(a.shift)) & 15) == 0;
  uint4 sv[FILL_IT], zv[FILL_IT];
  if (vec_fill && !(a.ablate & 4)) {
#pragma unroll
    for (int it = 0; it < FILL_IT; ++it) {
      const int e = tid + it * (WAVES * 64);
      if (e < NF * chunks) {
        const int f = e & (NF - 1), c = e >> 7;
        int p = p0 + (f & (PR - 1));
        p = p < P ? p : P - 1;
        const size_t row = (size_t)(p + (f >> 6) * P) * G + kt0 + c * 8;
        sv[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.scale) + row);
        zv[it] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.shift) + row);
      }
    }
  }
  issue_tile(0, 0);
  issue_tile(nk > 1 ? 1 : 0, 1);
  if (a.ablate & 4) {
  } else if (vec_fill) {
#pragma unroll
    for (int it = 0; it < FILL_IT; ++it) {
      const int e = tid + it * (WAVES * 64);
      if (e < NF * chunks) {
        const int f = e & (NF - 1), c = e >> 7;
        const T* se = reinterpret_cast<const T*>(&sv[it]);
        const T* ze = reinterpret_cast<const T*>(&zv[it]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          sz[((c * 8 + j) * 2 + 0) * NF + f] = se[j];
          sz[((c * 8 + j) * 2 + 1) * NF + f] = ze[j];
        }
      }
    }
  } else {
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
  // the table code contains compiler-visible loads with compiler-placed waits: drain once so that the hand-counted waits below
  // start from a known state (tiles 0 and 1 have landed, nothing outstanding)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // ---- fragment read offsets --------------------------------------------------------------------------------------------------------
  // activations: chunk of k-step t for lane group g is 8 (t >> 1) + 2 g + (t & 1) (the weight bytes the lane holds: k = 16 g +
  // 8 (t & 1) .. for t < 2, 64 + 16 g + 8 (t & 1) .. for t >= 2); rows i*16 + fi, so (row & 15) == fi for every fragment
  int xoff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) xoff[t] = fi * 256 + (((8 * (t >> 1) + 2 * fg + (t & 1)) ^ fi) << 4);
  // weights: 16-byte chunks fg and 4 + fg of packed row wave*8 + (fi & 7) of the tile (128-byte rows, chunk ^ (row & 7)); lanes fi and
  // fi + 8 read the same bytes (LDS broadcast) and keep the low / the high nibbles
  int woff[2];
  {
    const int r = wave * 8 + (fi & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) woff[h] = X_BYTES + r * 128 + (((4 * h + fg) ^ (r & 7)) << 4);
  }
  const uint32_t nib_shift = (fi >> 3) * 4;
  // this lane's 4 consecutive features inside the block: plane fg >> 1, local packed rows wave*8 + 4*(fg & 1) + r
  const int floc = (fg >> 1) * PR + wave * 8 + 4 * (fg & 1);

  f32x4 acc[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // operand construction, 8 VALU per k-step (8 weights per lane): the lane's nibble plane is shifted down and masked once per raw
  // dword, then ONE v_perm per pair of weights interleaves their bytes with the exponent byte of 128 (bf16 0x43) / 1024 (fp16 0x64)
  uint32_t nibmask = 0x0F0F0F0Fu, kmagic = Mma<DT>::MAGIC;
  asm volatile("" : "+s"(nibmask));
  asm volatile("" : "+v"(kmagic));
  const V8 ones = __builtin_bit_cast(V8, make_uint4(ONE2<DT>(), ONE2<DT>(), ONE2<DT>(), ONE2<DT>()));

  // Group accumulators are double-buffered: while tile kt accumulates into one set, the fold of tile kt-1 (scale / shift applied to
  // the other set) is sliced over the MFMA steps of tile kt - the two waves of a SIMD, re-synchronised by the barrier of every tile,
  // would otherwise run their MFMA phases together and then their fold phases together.
  f32x4 accgA[MI], accgB[MI];
  float s4[4], z4[4];  // scale and (shift + OFFSET * scale) of the lane's 4 features for the group being folded
  float xsp[MI];       // XS[token of this lane, group being folded] per fragment, from xs_slot
  // XS[m, g] = sum_k x[m, k] comes from the matrix pipe: an MFMA per k-step against an all-ones operand accumulates it in the same
  // order and with the same roundings as the products it corrects - measured necessary for fp16, where the 1024 offset leaves only
  // ~13 bits of the fp32 accumulator for the signal (XS from a separately ordered sum: 20 of 51 k outputs missed the 2-ulp gate).
  // r3: every wave used to run these products for all its fragments (a third of its MFMAs); now wave w runs them for fragment w only
  // and leaves the sums in LDS (xs_slot, double-buffered by tile parity) - the fold runs one tile later anyway, behind a barrier.
  const int my_xs = wave < MI ? wave : -1;
  auto load_sz = [&](int g) {
    T s4t[4], z4t[4];
    *reinterpret_cast<uint2*>(s4t) = *reinterpret_cast<const uint2*>(sz + (g * 2 + 0) * NF + floc);
    *reinterpret_cast<uint2*>(z4t) = *reinterpret_cast<const uint2*>(sz + (g * 2 + 1) * NF + floc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s4[r] = E::to_f32(s4t[r]);
      const float z = E::to_f32(z4t[r]);
      z4[r] = INT_SHIFT ? s4[r] * (z + Mma<DT>::OFFSET) : z + Mma<DT>::OFFSET * s4[r];
    }
  };
  auto load_xs = [&](int kt_prev) {
#pragma unroll
    for (int i = 0; i < MI; ++i) xsp[i] = xs_slot[(kt_prev & 1) * BM + i * 16 + fi];
  };
  // slice q (0 .. 4 MI - 1) of the fold of one group: feature r = q & 3 of fragment i = q >> 2
  // Two FMAs, as asm: left as C++, hipcc SINKS the whole fold (pure arithmetic whose result nobody reads before the next fold)
  // out of the MFMA steps to the end of the loop body, where it runs as one block while the matrix pipe idles (and its SLP
  // vectorizer turns it into v_pk_* there).  The operands were produced a tile ago (pg) or by LDS reads hipcc waits for.
  auto fold_slice = [&](const f32x4 (&pg)[MI], int q) {
    const int i = q >> 2, r = q & 3;
    float v = acc[i][r];
    asm volatile("v_fmac_f32 %0, %1, %2\n\tv_fma_f32 %0, -%3, %4, %0" : "+v"(v) : "v"(s4[r]), "v"(pg[i][r]), "v"(z4[r]), "v"(xsp[i]));
    acc[i][r] = v;
  };

  static_assert(STAGES == 2 && DEPTH == 1, "the stage of a tile is its parity: a compile-time tag below");
  constexpr int XR = 6, XD = 4;  // activation fragments: ring of 6 registers sets, fetched 4 steps ahead of their MFMA (one MFMA per step)
  // One tile (= one group) accumulating into cg while the previous tile's pg is folded.
  // stage_tag: with two stages the stage of tile kt is kt & 1 - known where the tile is instantiated, so every LDS address of the tile is
  // (lane offset register) + immediate instead of one v_add per fragment offset and tile
  auto tile = [&](int kt, f32x4 (&cg)[MI], const f32x4 (&pg)[MI], auto have_prev_tag, auto stage_tag) {
    constexpr bool have_prev = decltype(have_prev_tag)::value;
    constexpr int stage = decltype(stage_tag)::value;
    // requests of tile kt (issued two tiles ago) have landed once at most the one younger group is outstanding.  (Tiles 0 and 1:
    // completed by the prologue; fewer groups are outstanding than the count allows, the wait falls through.)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * OPS) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the fragment / table reads and the XS store of the previous tile
    __builtin_amdgcn_s_barrier();  // tile kt and the XS sums of tile kt-1 visible to all; everybody is done with tile kt-1, whose stage is refilled now
    asm volatile("" ::: "memory");
    {
      const int tn = kt + DEPTH < nk ? kt + DEPTH : last;
      constexpr int sn = 1 - stage;
      issue_tile(tn, sn);
    }
    const uint8_t* st = smem + stage * STAGE_BYTES;
    uint4 w[2];
    w[0] = *reinterpret_cast<const uint4*>(st + woff[0]);
    w[1] = *reinterpret_cast<const uint4*>(st + woff[1]);
    V8 xf[XR];
#pragma unroll
    for (int u = 0; u < XD; ++u) xf[u] = *reinterpret_cast<const V8*>(st + xoff[u / MI] + (u % MI) * 4096);
    if constexpr (have_prev) {
      load_sz(kt - 1);
      load_xs(kt - 1);
    }

    // 4 * MI steps (k-step t, token fragment i) of ONE product MFMA (+ the ones-product in the steps of this wave's XS fragment);
    // behind it a slice of the NEXT k-step's operand construction, a slice of the previous group's fold and the activation fragment
    // of step s + XD.  The sched_barrier pins that order: left alone, hipcc hoists all fragment reads to the top of the tile.
    uint32_t opw[2][4], mk[2];
    auto raw = [&](int t, int d) -> uint32_t {  // dword d (0, 1) of the 8 weight bytes of k-step t
      const uint4& q = w[t >> 1];
      return (t & 1) ? (d ? q.w : q.z) : (d ? q.y : q.x);
    };
    auto conv_op = [&](int t, int o) {  // the 8 VALU ops that build the operand of k-step t: 2 x (shift, mask), 4 x v_perm
      if (o == 0 || o == 2) {
        mk[o >> 1] = raw(t, o >> 1) >> nib_shift;
      } else if (o == 1 || o == 3) {
        mk[o >> 1] &= nibmask;
      } else {
        const int c = o - 4;  // operand dword c: weights 2c, 2c + 1 -> (q, exp, q', exp)
        opw[t & 1][c] = __builtin_amdgcn_perm(kmagic, mk[c >> 1], (c & 1) ? 0x07030502u : 0x07010500u);
      }
    };
#pragma unroll
    for (int o = 0; o < 8; ++o) conv_op(0, o);
#pragma unroll
    for (int s = 0; s < 4 * MI; ++s) {
      const int t = s / MI, i = s % MI;
      const V8 wa = __builtin_bit_cast(V8, make_uint4(opw[t & 1][0], opw[t & 1][1], opw[t & 1][2], opw[t & 1][3]));
      cg[i] = Mma<DT>::run(wa, xf[s % XR], t == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : cg[i]);
      if (t < 3) {
        // the operand of k-step t+1, spread over the MI steps of k-step t (the masks of ops 0..3 are consumed by ops 4..7 of the
        // same k-step, all later in this loop)
#pragma unroll
        for (int o = i * 8 / MI; o < (i + 1) * 8 / MI; ++o) conv_op(t + 1, o);
      }
      if constexpr (have_prev) fold_slice(pg, s);
      if (s + XD < 4 * MI) xf[(s + XD) % XR] = *reinterpret_cast<const V8*>(st + xoff[(s + XD) / MI] + ((s + XD) % MI) * 4096);
      __builtin_amdgcn_sched_barrier(0);
    }
    // XS of this wave's fragment: four ones-products in k-step order (the order of the products they correct), ONE wave-uniform
    // branch per tile; every row of the result holds sum_k x[token, k] of this group: lanes 0..15 (row 0) leave it for the fold
    // one tile later
    if (my_xs >= 0) {
      f32x4 cx = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) cx = Mma<DT>::run(ones, *reinterpret_cast<const V8*>(st + xoff[t] + my_xs * 4096), cx);
      if (lane < 16) xs_slot[(kt & 1) * BM + my_xs * 16 + lane] = cx[0];
    }
  };
  auto final_fold = [&](const f32x4 (&pg)[MI]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the XS sums of the last tile
    asm volatile("" ::: "memory");
    load_sz(nk_run - 1);
    load_xs(nk_run - 1);
#pragma unroll
    for (int q = 0; q < 4 * MI; ++q) fold_slice(pg, q);
  };
  using yes = std::integral_constant<bool, true>;
  using st0 = std::integral_constant<int, 0>;
  using st1 = std::integral_constant<int, 1>;
  tile(0, accgA, accgB, std::integral_constant<bool, false>{}, st0{});
  int kt = 1;
  for (; kt + 2 <= nk_run; kt += 2) {
    tile(kt, accgB, accgA, yes{}, st1{});
    tile(kt + 1, accgA, accgB, yes{}, st0{});
  }
  if (kt < nk_run) {
    tile(kt, accgB, accgA, yes{}, st1{});  // nk even: the last tile landed in set B
    final_fold(accgB);
  } else {
    final_fold(accgA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-requested tiles past the end: nothing may land in LDS after the kernel moved on

  // ---- split-K: fp32 partial tiles through the workspace, the last workgroup of a tile adds them in split order (qbits_skinny.hip) ----
  if (S > 1) {
    const int tile_id = blockIdx.y * gridDim.x + blockIdx.x;
    // fragment-major: lane-major (64 B per lane) made every store instruction write a quarter of each line it touched, and partial
    // lines are what the write-through path is slow at
    float* mine = a.partials + ((size_t)(tile_id * S + sp) * MI * (WAVES * 64) + tid) * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i)  // s_nop: gfx9 hazard "VMEM store of > 64 bits, then VALU write of its data VGPRs"
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(mine + i * (WAVES * 64 * 4)), "v"(acc[i]) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) *flag = __hip_atomic_fetch_add(a.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (*flag != S - 1) return;
    if (tid == 0) __hip_atomic_store(a.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // leave the workspace as found
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fixed order: the result does not depend on which workgroup arrived last.  The loads of up to four splits are in flight
    // together: a system-coherent load is a ~2 us round trip, and r2's loop paid one per split ((128,4096,4096) with 4 splits:
    // ~12 of its 21.7 us were this tail)
    constexpr int QB = BM == 64 ? 4 : 2;  // splits per batch: QB * MI float4 registers
    for (int q0 = 0; q0 < S; q0 += QB) {
      f32x4 v[QB][MI];
#pragma unroll
      for (int j = 0; j < QB; ++j) {
        const int q = q0 + j < S ? q0 + j : S - 1;
        const float* theirs = a.partials + ((size_t)(tile_id * S + q) * MI * (WAVES * 64) + tid) * 4;
#pragma unroll
        for (int e = 0; e < MI; ++e) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[j][e]) : "v"(theirs + e * (WAVES * 64 * 4)) : "memory");
      }
#pragma unroll
      for (int j = 0; j < QB; ++j)
#pragma unroll
        for (int e = 0; e < MI; ++e) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[j][e])::"memory");  // ties the uses below to the wait
#pragma unroll
      for (int j = 0; j < QB; ++j)
        if (q0 + j < S) {
#pragma unroll
          for (int e = 0; e < MI; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[e][r] += v[j][e][r];
        }
    }
  }

  // ---- epilogue: 4 consecutive features of one token per fragment: 8-byte stores -------------------------------------------------------
  T* yg = reinterpret_cast<T*>(a.y);
  const bool has_bias = a.bias != nullptr;
  const int pl = p0 + wave * 8 + 4 * (fg & 1);  // first of the lane's 4 packed rows
  const int n0 = pl + (fg >> 1) * P;            // 4 consecutive output features n0 .. n0+3
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (has_bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = pl + r < P ? E::to_f32(reinterpret_cast<const T*>(a.bias)[n0 + r]) : 0.f;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + i * 16 + fi;
    if (m < M && !(a.ablate & 2)) {
      T out[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][r];
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

inline int lds_bytes(int groups, int bm) { return STAGES * (bm * BK * 2 + W_BYTES) + 2 * bm * 4 + groups * 2 * (2 * PR) * 2; }

inline int tiles_of(int64_t M, int64_t N, int bm) { return (int)(((N / 2 + PR - 1) / PR) * ((M + bm - 1) / bm)); }

// Token tile and K split, chosen together from a small time model fitted to r3's sweeps (profiles/r03_fused_int4_gemm.md; us):
//   t = 5.8 + rounds * groups_per_workgroup * t_tile + tail,   rounds = ceil(workgroups / 256 CUs),
//   t_tile = 0.68 (64-token tiles) / 1.2 (128-token tiles),     tail = 3.5 + 0.5 per MB of fp32 partial tiles when K is split
//   (1.2 per MB until the partial tiles were laid out fragment-major: whole lines per write-through store instruction).
// 128 tokens per workgroup halve the activation bytes per weight byte, but a short prefill then leaves CUs idle ((512,4096,4096) is
// 128 tiles of 128 tokens on 256 CUs: 46.7 us against 28.5 with 64-token tiles); a split costs its tail (a 32 / 64 KiB partial tile
// per workgroup through the fabric and back, arrival counter, one more round trip for the last workgroup), so it pays for few
// tiles or long K only: (128,4096,4096) 27.6 / 20.0 / 16.2 / 17.9 us with 1 / 2 / 4 / 8 splits, (128,14336,4096) 83 / 48 / 34 / 34,
// (256,4096,4096) 27.9 / 21.5 / 20.8 / 28.7, but (512,4096,4096) 29.4 / 36.1.  The scale tables of a workgroup's groups must fit the LDS next to the ring (K = 14336 with 128-token
// tiles needs a split for that alone).
struct Plan {
  int bm, S;
  float us;
};
inline float model_us(int tiles, int nk, int bm, int S) {
  const int wgs = tiles * S, rounds = (wgs + 255) / 256;
  const float tail = S > 1 ? 3.5f + 0.5f * (float)wgs * (float)(bm * 512) * 1e-6f : 0.f;
  return 5.8f + (float)rounds * (float)nk * (bm == 64 ? 0.68f : 1.2f) + tail;
}
inline Plan make_plan(int64_t M, int64_t N, int G) {
  const int fbm = env_int("QUANTO_HIP_FUSED4_BM", 0), fs = env_int("QUANTO_HIP_FUSED4_SPLIT", 0);  // experiments / tests
  Plan best{0, 0, 0.f};
  for (int bm = 64; bm <= 128; bm += 64) {
    if ((fbm == 64 || fbm == 128) && bm != fbm) continue;
    const int tiles = tiles_of(M, N, bm);
    for (int S = 1; S <= 8; S *= 2) {
      if (G % S) break;
      const int nk = G / S;
      if (fs > 0 ? (S != fs) : (S > 1 && nk < 4)) continue;
      if (lds_bytes(nk, bm) > 160 * 1024) continue;
      if (S > 1 && (size_t)tiles * 4 > QUANTO_HIP_WS_COUNTER_BYTES) continue;
      const float us = model_us(tiles, nk, bm, S);
      if (best.bm == 0 || us < best.us * 0.97f) best = Plan{bm, S, us};  // ties: the smaller tile, fewer splits
    }
  }
  return best;  // bm == 0: no configuration fits (a forced split that does not divide the groups, tables that never fit)
}

template <int DT, bool INT_SHIFT, int BM>
static int launch_bm(const Args& a, hipStream_t stream) {
  const int lds = lds_bytes(a.G / a.S, BM);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbits_mfma_fused_kernel<DT, INT_SHIFT, BM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const dim3 grid((unsigned)((a.N / 2 + PR - 1) / PR), (unsigned)((a.M + BM - 1) / BM), (unsigned)a.S);
  hipLaunchKernelGGL((qbits_mfma_fused_kernel<DT, INT_SHIFT, BM>), grid, dim3(WAVES * 64), lds, stream, a);
  return launch_status();
}

template <int DT, bool INT_SHIFT>
static int launch(const Args& a, int bm, hipStream_t stream) {
  return bm == 64 ? launch_bm<DT, INT_SHIFT, 64>(a, stream) : launch_bm<DT, INT_SHIFT, 128>(a, stream);
}

}  // namespace fused4

bool qbits_mfma_fused_supported(int64_t M, const PackedGeom& g, int dtype) {
  if (!(g.bits == 4 && g.C == 128 && (g.N % 8 == 0) && (g.K % 128 == 0) && M >= 1 && (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) &&
        g.N < (1 << 30) && g.K < (1 << 30) && M * g.K < (1ll << 31) && g.N * g.K < (1ll << 33)))
    return false;
  return fused4::make_plan(M, g.N, (int)g.G).bm != 0;
}

// modelled time of the configuration make_plan picks, in units of "one round of 128-token tiles at this K" (46 us at K = 4096): what
// c_api.hip compares with the dequantize + dense GEMM path (flat in M up to ~1 k rows)
float qbits_mfma_fused_cost(int64_t M, const PackedGeom& g) {
  const fused4::Plan p = fused4::make_plan(M, g.N, (int)g.G);
  return p.bm == 0 ? 1e9f : p.us / (46.f * (float)g.K / 4096.f);
}

// true when the problem cannot run without the split-K scratch (no unsplit configuration fits the LDS: the scale tables of K = 14336)
bool qbits_mfma_fused_needs_workspace(const PackedGeom& g) {
  return fused4::lds_bytes((int)g.G, 128) > 160 * 1024 && fused4::lds_bytes((int)g.G, 64) > 160 * 1024;
}

// [counters (zero on entry, zero on exit) | fp32 partial tiles]; 0 when K is not split (the group sums of x come from the matrix pipe)
size_t qbits_mfma_fused_workspace(int64_t M, const PackedGeom& g) {
  const fused4::Plan p = fused4::make_plan(M, g.N, (int)g.G);
  if (p.bm == 0 || p.S == 1) return 0;
  return QUANTO_HIP_WS_COUNTER_BYTES + (size_t)fused4::tiles_of(M, g.N, p.bm) * p.S * (fused4::WAVES * 64) * ((p.bm / 16) * 16);
}

int qbits_mm_mfma_fused(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t M,
                        const PackedGeom& g, int dtype, bool int_shift, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!qbits_mfma_fused_supported(M, g, dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed)) % 16) return QUANTO_HIP_EALIGN;
  fused4::Plan p = fused4::make_plan(M, g.N, (int)g.G);
  if (p.S > 1 && (!workspace || workspace_bytes < qbits_mfma_fused_workspace(M, g) || reinterpret_cast<uintptr_t>(workspace) % 16)) {
    // no scratch: unsplit, with whichever token tile lets the whole scale table fit
    p.S = 1;
    if (fused4::lds_bytes((int)g.G, p.bm) > 160 * 1024) p.bm = 64;
    if (fused4::lds_bytes((int)g.G, p.bm) > 160 * 1024) return QUANTO_HIP_EINVAL;
  }
  const int bm = p.bm, S = p.S;
  fused4::Args a{x, packed, scale, shift, bias, y, (int)M, (int)g.N, (int)g.K, (int)g.G, S, reinterpret_cast<int*>(workspace),
                 S > 1 ? reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(workspace) + QUANTO_HIP_WS_COUNTER_BYTES) : nullptr,
                 env_int("QUANTO_HIP_FUSED4_ABLATE", 0)};
  if (dtype == QUANTO_HIP_BF16)
    return int_shift ? fused4::launch<QUANTO_HIP_BF16, true>(a, bm, stream) : fused4::launch<QUANTO_HIP_BF16, false>(a, bm, stream);
  return int_shift ? fused4::launch<QUANTO_HIP_F16, true>(a, bm, stream) : fused4::launch<QUANTO_HIP_F16, false>(a, bm, stream);
}

}  // namespace qh
