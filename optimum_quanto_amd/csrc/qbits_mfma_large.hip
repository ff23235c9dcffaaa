// qbits_mm for prefill-sized M, no dense weight anywhere: int4-packed weights -> registers -> bf16 / fp16 MFMA operands with the
// REFERENCE's rounding sequence (bit-identical to dequantize_qbits_kernel, i.e. to the reference's dequantize()), 256 x 256 tiles.
//
//   y[M,N] = x[M,K] @ W^T,   W[n,k] = T(T(scale[n,g] * q[n,k]) - shift[n,g])           (tensor/qbits.py:27-49: two roundings to T)
//                            W[n,k] = T(scale[n,g] * (q[n,k] - zp[n,g]))                (integer zero-point: one rounding)
//
// What it replaces: QUANTO_HIP_KERNEL_DEQUANT_MFMA (a dequantize pass that writes N*K*2 bytes into a workspace + a dense GEMM that
// reads them back: 5.1 x the algorithmic traffic at 4096^3, and the structure the reference itself has on ROCm).  The fused exact-math
// kernel (qbits_mfma_fused.hip) pays 2 fp32 FMAs per output and group for its fold and loses beyond ~1.6 rounds of tiles; here the
// per-group scale / shift go into the OPERAND (as the reference does) and the K loop is qmm_mfma_large.hip's: one software-pipelined
// MFMA stream per wave, one barrier per K-tile.  Functional analogs on CUDA: awq/v2/gemm_cuda.cu:924-1026, marlin/marlin_cuda_kernel.cu:191-720.
//
// Workgroup = 8 waves, 256 tokens x 128 PACKED rows (= 256 features: byte (p,k) = W[p,k] | W[p+N/2,k] << 4, so the output tile is two
// 128-wide column blocks, at p0 and at N/2 + p0).  All waves side by side along the features (1 x 8): wave w owns packed rows
// 16w..16w+15 (32 features, both nibble planes) and ALL 256 tokens - every weight is converted exactly once per workgroup, which is
// what makes the conversion affordable (11 VALU per pair of weights: 2 cvt_f32_ubyte, 2 mul, 2 x (cvt_pk + shift) for the first rounding,
// 2 sub, cvt_pk; ~2.9 VALU per MFMA; a 2 x 4 layout would convert everything twice).
// SCALAR fp32 math on purpose (and -fno-slp-vectorize for this file): with v_pk_mul_f32 / v_pk_add_f32 - 4 VALU per pair fewer - one
// operand element of the last 16 lanes came out wrong in one wave every few launches (always a v_pk_add_f32 whose low lane takes the HIGH
// half of its source pair, op_sel:[0,1], issued in the shadow of an MFMA; never reproduced with scalar instructions: r4,
// profiles/r04_packed_fp32_next_to_mfma.md).
//   * activations: LDS-DMA ring of three stages (32 KiB each) (global_load_lds_dwordx4, swizzled as in qmm_mfma_large.hip), fragments two
//     steps ahead through a ring of four;
//   * packed weights ride the same ring (8 KiB per stage); a wave only ever reads back the 16 packed rows it fetched itself: lane
//     (r = lane & 15, g = lane >> 4) takes its 16 bytes (k = 16g..16g+15 of the K-tile, both planes) with one ds_read_b128 per K-tile;
//   * scale / shift: a 32-group window per workgroup in LDS ({s, z} pairs, 32 KiB; K = 4096 never refills), refilled 8 groups at a time while the loop runs (K = 14336
//     has 112 groups); a lane re-reads its two entries once per K-tile;
//   * step (kk, i), 32 per K-tile: 2 MFMAs (both planes) x token fragment i; phase kk = 0 converts this tile's k-half-1 operands,
//     phase kk = 1 the next tile's k-half-0 operands - one pair of weights every step;
//   * epilogue: the tile is parked in LDS and stored as whole 256-byte rows per column block.
// Group sizes: multiples of 64 and per-channel (a K-tile lies inside one group), and - HG, r5 - the other multiples of 32 the reference
// produces for K % 128 != 0 layers (nn/qmodule.py:121-129: 96, 32): there a group boundary may fall on k = 32 of a K-tile.  A lane's 16
// packed bytes are k = 16 fg .. 16 fg + 15 of the tile (its two MFMA k-halves are the lower and the upper 8 of them), so lane groups
// fg = 0, 1 take the scale / shift of the group of the tile's first 32 k and fg = 2, 3 that of its last 32 - one table entry per lane and
// tile, as before, with a lane-dependent group index.  int4 only.
#include <type_traits>

#include "qmm_large_common.h"

namespace qh {
namespace l4 {

using lt::BK;  // 64
using lt::glds16;
using lt::lds_ptr_t;
using lt::Mma;
using lt::swz_a;
using lt::swz_w;

constexpr int BM = 256;           // tokens per workgroup
constexpr int BP = 128;           // packed rows per workgroup (256 features)
constexpr int NW = 8;             // waves
constexpr int MI = BM / 16;       // token fragments per wave (all of them)
constexpr int STEPS = 2 * MI;     // (k-half, token fragment)
constexpr int STAGES = 3;
constexpr int A_BYTES = BM * BK * 2;          // 32 KiB of activations per stage
constexpr int W_BYTES = BP * BK;              // + 8 KiB of packed weights
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
constexpr int APIECES = BM / 8 / NW;          // 4 activation DMA pieces (8 rows x 128 B) per wave and K-tile
constexpr int TG = 32, TH = 8;                // groups in the table window / per refill unit
constexpr int TAB_BYTES = TG * 2 * BP * 4;    // [slot][plane][row] {scale, shift} as two T: 32 KiB
constexpr int OUT_PITCH = 2 * BP * 2 + 16;    // parked output row: 256 features of T + padding
constexpr int LOOP_BYTES = STAGES * STAGE_BYTES + TAB_BYTES;                             // 152 KiB while the K loop runs
constexpr int LDS_BYTES = LOOP_BYTES > BM * OUT_PITCH ? LOOP_BYTES : BM * OUT_PITCH;     // the epilogue parks the output tile (132 KiB) over the ring + table

struct Args {
  const void* x;
  const uint8_t* w;     // packed [N/2, K]
  const void* scale;    // [N * G]
  const void* shift;    // [N * G]: T (float shift) or uint8 / int8 (zero-point)
  const void* bias;     // [N] or null
  void* y;
  int M, N, K;
  int C;                // group size (K for per-channel)
  int G;                // groups per feature
  int group_m;          // tile raster (lt::tile_coords)
};

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// T(v) as an fp32 number: the first of the reference's two roundings, kept in fp32 for the subtraction that follows (bf16: v_cvt_pk_bf16_f32
// + a 16-bit shift.  v_cvt_pk_bf16_f32 with a zero LOW half would be the fp32 image in one instruction, but hipcc only emits that form
// from inline asm, and with it this kernel spills: 138.9 -> 144.6 us at 4096^3).
template <int DT>
__device__ __forceinline__ float round_to_T(float v) {
  if constexpr (DT == QUANTO_HIP_BF16)
    return (float)(__bf16)v;
  else
    return (float)(_Float16)v;
}

// two fp32 -> one dword of T, round to nearest even (lt::Mma<F16>::pack rounds toward zero: exact only for the 8-bit formats it serves)
template <int DT>
__device__ __forceinline__ uint32_t pack_rne(float a, float b) {
  if constexpr (DT == QUANTO_HIP_BF16) {
    bf16x2 r;
    r.x = (__bf16)a;
    r.y = (__bf16)b;
    return __builtin_bit_cast(uint32_t, r);
  } else {
    f16x2 r;
    r.x = (_Float16)a;
    r.y = (_Float16)b;
    return __builtin_bit_cast(uint32_t, r);
  }
}

// Two weights of one feature -> one MFMA operand dword.  `spread`: the lane's nibble plane spread to bytes (plane 0: q, plane 1: 16 q -
// the factor 16 is folded into s / zp below, exactly: powers of two).  `PAIR`: byte pair 0 / 1 of the dword.
template <int DT, bool INT_SHIFT, int PAIR>
__device__ __forceinline__ uint32_t convert_pair4(uint32_t spread, float s, float z) {
  // `spread` is opaque to the optimizer (see convert): these two are v_cvt_f32_ubyte{0,1} / {2,3}, emitted by the compiler
  const float q0 = (float)((spread >> (16 * PAIR)) & 0xFFu), q1 = (float)((spread >> (16 * PAIR + 8)) & 0xFFu);
  if constexpr (INT_SHIFT) {
    return pack_rne<DT>((q0 - z) * s, (q1 - z) * s);  // (q - zp) exact, one rounding (tensor/qbits.py:35-42)
  } else {
    // q * s is exact in fp32 (8 x 4 significant bits), rounded to T as `scale * data` is (tensor/qbits.py:41); `dqt -= shift` (:44) is
    // the fp32 difference of two T values, rounded to T by the pack
    const float r0 = round_to_T<DT>(q0 * s), r1 = round_to_T<DT>(q1 * s);
    return pack_rne<DT>(r0 - z, r1 - z);
  }
}

// FULLM: M is a multiple of 256 - the activation rows of DMA piece j are the lane's piece-0 row + 64 j, a wave-uniform byte offset added to
// the SGPR base (one VGPR for all four pieces).  Ragged M clamps every row to M - 1 and recomputes the lane's offset per piece
// (3 VALU each): four live VGPRs more made hipcc spill INSIDE the loop - among others the registers of weight loads still in flight.
template <int DT, bool INT_SHIFT, bool FULLM, bool HG>
__global__ void __launch_bounds__(NW * 64, 1) qbits_mfma_large_kernel(const Args a) {
  using E = Elem<DT>;
  using T = typename E::T;
  using V8 = typename Mma<DT>::V8;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* tab = reinterpret_cast<uint32_t*>(smem + STAGES * STAGE_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M, N = a.N, K = a.K, C = a.C, G = a.G;
  const int P = N >> 1;
  const int nk = K / BK;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (P + BP - 1) / BP;
  int tm, tn;
  lt::tile_coords(blockIdx.x, tiles_m, tiles_n, a.group_m, tm, tn);
  const int m0 = tm * BM, p0 = tn * BP;

  // ---- activation DMA (qmm_mfma_large.hip's image: 8 rows x 128 B per piece, chunk c of row R stored at position c ^ swz_a(R)) ----
  // piece j of wave w: rows (j * NW + w) * 8 + (lane >> 3); the swizzle only depends on the row modulo 16, i.e. not on j
  const int arow = m0 + wave * 8 + (lane >> 3);
  const uint32_t acol = (uint32_t)(((lane & 7) ^ swz_a(wave * 8 + (lane >> 3))) * 16);
  const uint32_t asrc0 = (uint32_t)(arow < M ? arow : M - 1) * (uint32_t)K * 2u + acol;  // 32-bit arithmetic: M * K < 2^30 (launcher)
  auto a_offset = [&](int j) -> uint32_t {  // the lane's byte offset of piece j (ragged M)
    const int r = arow + j * (NW * 8);
    return (uint32_t)(r < M ? r : M - 1) * (uint32_t)K * 2u + acol;
  };
  const size_t apiece_stride = (size_t)(NW * 8) * K * 2;  // FULLM: piece j = piece 0 + j * 64 rows
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.x);
  uint32_t adst[STAGES][APIECES];
#pragma unroll
  for (int t = 0; t < STAGES; ++t)
#pragma unroll
    for (int j = 0; j < APIECES; ++j) adst[t][j] = __builtin_amdgcn_readfirstlane(lds_base + t * STAGE_BYTES + (j * NW + wave) * 1024);

  // ---- weights: wave w DMAs its own 16 packed rows (one 1 KiB piece per K-tile: lane -> row lane >> 2, 16-byte chunk (lane & 3) ^ swz_w(row),
  // qmm_mfma_large.hip's image) and reads them back as fragments: lane (fr, fg) = the 16 bytes k = 16 fg .. 16 fg + 15 of row 16 w + fr.
  // Through the LDS, not straight into registers: an asm global_load hands its result to a C++ variable long before the data arrives,
  // and hipcc is free to copy that variable (a v_mov at the loop back-edge, a spill) before the counted wait - r4, first version of this
  // kernel: one wrong operand element per few launches.  LDS-DMA writes no register; the ds_read below is compiler-visible.
  const int fr = lane & 15, fg = lane >> 4;
  uint32_t wsrc;
  {
    const int R = wave * 16 + (lane >> 2);
    int p = p0 + R;
    p = p < P ? p : P - 1;
    wsrc = (uint32_t)p * (uint32_t)K + (uint32_t)(((lane & 3) ^ swz_w(R)) * 16);  // N * K < 2^32 (launcher)
  }
  uint32_t wdst[STAGES];
#pragma unroll
  for (int t = 0; t < STAGES; ++t) wdst[t] = __builtin_amdgcn_readfirstlane(lds_base + t * STAGE_BYTES + A_BYTES + wave * 1024);
  const int woff = A_BYTES + (wave * 16 + fr) * 64 + ((fg ^ swz_w(wave * 16 + fr)) << 4);
  u32x4 raw;  // the bytes being converted: tile t's from the start of tile t - 1's second phase (.x .y = k-half 0) to the end of tile t's first (.z .w)
  auto read_w = [&](int stage) { return *reinterpret_cast<const u32x4*>(smem + stage * STAGE_BYTES + woff); };

  // ---- fragment read offsets (activations): chunk 2 fg + kk of row fr (+ 16 i), see qmm_mfma_large.hip ------------------------------
  int aoff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) aoff[kk] = fr * 128 + (((fg * 2 + kk) ^ swz_a(fr)) << 4);

  // ---- scale / shift table: slot = group % TG; thread (f = tid & 255, q = tid >> 8) fetches TH groups of table q for feature f ------
  const int tf_plane = (tid & 255) >> 7, tf_row = tid & 127, tq = tid >> 8;
  uint32_t tf_index;  // first group of this feature in the scale / shift arrays (32-bit: the loads then take an SGPR base + one VGPR offset)
  {
    int p = p0 + tf_row;
    p = p < P ? p : P - 1;
    tf_index = (uint32_t)(tf_plane * P + p) * (uint32_t)G;
  }
  uint16_t pend[TH];
  auto table_fetch = [&](int g0) {  // groups g0 .. g0 + TH - 1 (clamped) -> registers
#pragma unroll
    for (int i = 0; i < TH; ++i) {
      const uint32_t e = tf_index + (uint32_t)(g0 + i < G ? g0 + i : G - 1);
      if (tq == 0) {
        pend[i] = reinterpret_cast<const uint16_t*>(a.scale)[e];
      } else if constexpr (INT_SHIFT) {
        const T z = E::from_f32((float)(int8_t) reinterpret_cast<const uint8_t*>(a.shift)[e]);  // small integers: exact
        pend[i] = __builtin_bit_cast(uint16_t, z);
      } else {
        pend[i] = reinterpret_cast<const uint16_t*>(a.shift)[e];
      }
    }
  };
  auto table_store = [&](int g0) {  // registers -> slots (g0 + i) % TG, half-dword tq of entry [slot][plane][row]
    uint16_t* t16 = reinterpret_cast<uint16_t*>(tab);
#pragma unroll
    for (int i = 0; i < TH; ++i) t16[((((g0 + i) % TG) * 2 + tf_plane) * BP + tf_row) * 2 + tq] = pend[i];
  };
  // this lane's entries: plane j, row 16 wave + fr
  auto table_read = [&](int g, float (&s2)[2], float (&z2)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t e = tab[((g % TG) * 2 + j) * BP + wave * 16 + fr];
      T st, zt;
      st = __builtin_bit_cast(T, (uint16_t)(e & 0xFFFFu));
      zt = __builtin_bit_cast(T, (uint16_t)(e >> 16));
      float s = E::to_f32(st), z = E::to_f32(zt);
      if (j == 1) {  // plane 1 is converted from 16 q: fold the 16 into the scale (and into the zero-point), exactly
        s *= 0.0625f;
        if constexpr (INT_SHIFT) z *= 16.f;
      }
      s2[j] = s;
      z2[j] = z;
    }
  };

  f32x4 acc[2][MI];  // acc[j][i]: plane j (packed rows 16 wave + 4 fg + r), tokens 16 i + fr
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint32_t w0[2][4], w1[2][4];  // converted operands: [plane][dword], k-half 0 / 1
  V8 xf[4];
  auto as_v8 = [&](const uint32_t(&w)[4]) { return __builtin_bit_cast(V8, make_uint4(w[0], w[1], w[2], w[3])); };
  // dword c (0..7) of a k-half: plane c >> 2, operand dword d = c & 3 = byte pair d & 1 of raw dword d >> 1 of that half
  auto convert = [&](const u32x4& r, int kk, int c, const float (&s2)[2], const float (&z2)[2]) -> uint32_t {
    const int j = c >> 2, d = c & 3;
    const uint32_t word = kk == 0 ? (d < 2 ? r.x : r.y) : (d < 2 ? r.z : r.w);
    uint32_t spread = j == 0 ? (word & 0x0F0F0F0Fu) : (word & 0xF0F0F0F0u);
    asm("" : "+v"(spread));  // keeps hipcc from folding the nibble mask into per-byte extractions (v_and_b32_sdwa + v_cvt_f32_ubyte0 per weight)
    return (d & 1) ? convert_pair4<DT, INT_SHIFT, 1>(spread, s2[j], z2[j]) : convert_pair4<DT, INT_SHIFT, 0>(spread, s2[j], z2[j]);
  };

  // ---- prologue: table halves 0 and 1, tiles 0 and 1 in flight, tile 0's k-half-0 operands converted ---------------------------------
#pragma unroll 1
  for (int g0 = 0; g0 < TG; g0 += TH) {
    table_fetch(g0);
    table_store(g0);
  }
  auto issue_w = [&](int kt, int stage) { glds16(a.w + (size_t)kt * BK, wsrc, wdst[stage]); };
  auto issue_a = [&](int kt, int piece, uint32_t dst) {
    if constexpr (FULLM)
      glds16(xbase + (size_t)kt * (BK * 2) + piece * apiece_stride, asrc0, dst);
    else
      glds16(xbase + (size_t)kt * (BK * 2), a_offset(piece), dst);
  };
#pragma unroll
  for (int p = 0; p < APIECES; ++p) issue_a(0, p, adst[0][p]);
  issue_w(0, 0);
  if (nk > 1) {
#pragma unroll
    for (int p = 0; p < APIECES; ++p) issue_a(1, p, adst[1][p]);
    issue_w(1, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // group bookkeeping (wave-uniform): group of the current tile (HG: of its first 32 k), k offset of the current tile inside its group
  int g_cur = 0, k_in_g = 0;
  float sc[2], zc[2], sn[2], zn[2];  // scale / shift (per plane) of the current and of the next tile's group (HG: the group of the lane's 16 k)
  table_read(HG && fg >= 2 && BK / 2 >= C ? 1 : 0, sc, zc);
  raw = read_w(0);
#pragma unroll
  for (int c = 0; c < 8; ++c) w0[c >> 2][c & 3] = convert(raw, 0, c, sc, zc);
  xf[0] = *reinterpret_cast<const V8*>(smem + aoff[0]);
  xf[1] = *reinterpret_cast<const V8*>(smem + aoff[0] + 2048);
  xf[2] = *reinterpret_cast<const V8*>(smem + aoff[0] + 4096);

  auto tile = [&](auto p_tag, int kt, auto dma_tag, auto barrier_tag) {
    constexpr int PS = decltype(p_tag)::value, PN = (PS + 1) % STAGES, PF = (PS + 2) % STAGES;
    const bool dma = dma_tag, barrier = barrier_tag;  // integral_constants in the steady state (no branches), run-time flags in the tail
    const uint8_t* st = smem + PS * STAGE_BYTES;
    const uint8_t* sx = smem + PN * STAGE_BYTES;
    // the next tile's group (HG: the groups of its first and of its last 32 k; C is a multiple of 32)
    int g_next = g_cur, k_next = k_in_g + BK;
    int g_next1 = 0;
    if constexpr (HG) {
      k_next = k_in_g + BK / 2;
      if (k_next >= C) {
        k_next = 0;
        ++g_next;
      }
      k_next += BK / 2;
      if (k_next >= C) {
        k_next = 0;
        ++g_next;
      }
      g_next1 = k_next + BK / 2 >= C ? g_next + 1 : g_next;
    } else if (k_next >= C) {
      k_next = 0;
      ++g_next;
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int kk = s / MI, i = s % MI;
      if (s == MI) {  // before the first conversion of the second phase: the next tile's scale / shift and packed bytes
        table_read(HG && fg >= 2 ? g_next1 : g_next, sn, zn);
        raw = read_w(PN);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (kk == 0)
          acc[j][i] = Mma<DT>::run(as_v8(w0[j]), xf[s & 3], acc[j][i]);
        else
          acc[j][i] = Mma<DT>::run(as_v8(w1[j]), xf[s & 3], acc[j][i]);
        if (j == 0) {
          // one pair of weights per step: dword c = i / 2 in even steps ... split over both MFMA slots by the scheduler barrier below
          if ((i & 1) == 0) {
            const int c = i >> 1;
            if (kk == 0)
              w1[c >> 2][c & 3] = convert(raw, 1, c, sc, zc);   // this tile's k-half 1
            else
              w0[c >> 2][c & 3] = convert(raw, 0, c, sn, zn);   // next tile's k-half 0 (garbage, unused, behind the last tile)
          }
        } else {
          // activation fragment of step s + AHEAD (the next tile's first ones at the end): a step is only two MFMAs (~36 cycles of
          // matrix pipe), so the ring of four is used to its full depth - three fragments in flight behind the one in use
          constexpr int AHEAD = 3;
          xf[(s + AHEAD) & 3] = s + AHEAD < STEPS ? *reinterpret_cast<const V8*>(st + aoff[(s + AHEAD) / MI] + ((s + AHEAD) % MI) * 2048)
                                                  : *reinterpret_cast<const V8*>(sx + aoff[0] + (s + AHEAD - STEPS) * 2048);
          if (s < APIECES) {
            if (dma) issue_a(kt + 2, s, adst[PF][s]);
          } else if (s == APIECES) {
            if (dma) issue_w(kt + 2, PF);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // tile boundary: own DMA share + weight bytes of tile kt + 2 have landed -> barrier
    if (barrier) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // the next tile opens a new unit of TH groups: every group of the unit before it is dead (its last reader was the table_read of the
      // PREVIOUS tile, behind that tile's barrier) -> its slots take the groups one window further, first read TG - TH groups from now.
      // Fetched and stored right here, before the barrier (only K > 32 groups ever gets here: ~1.5 us every 8 groups)
      if (g_next != g_cur && g_next % TH == 0 && g_next >= TH && g_next - TH + TG < G) {
        table_fetch(g_next - TH + TG);
        table_store(g_next - TH + TG);
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    g_cur = g_next;
    k_in_g = k_next;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      sc[j] = sn[j];
      zc[j] = zn[j];
    }
  };
  using yes = std::integral_constant<bool, true>;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  int kt = 0;
  for (; kt + 4 < nk; kt += 3) {  // three tiles that all still have a tile kt + 2 to fetch
    tile(S0{}, kt, yes{}, yes{});
    tile(S1{}, kt + 1, yes{}, yes{});
    tile(S2{}, kt + 2, yes{}, yes{});
  }
  const int rem = nk - kt;  // 2..4 tiles left, kt % 3 == 0
  tile(S0{}, kt, rem > 2, true);
  tile(S1{}, kt + 1, rem > 3, rem > 2);
  if (rem > 2) tile(S2{}, kt + 2, false, rem > 3);
  if (rem > 3) tile(S0{}, kt + 3, false, false);

  // ---- epilogue: park the tile in LDS ([token][plane][128 packed rows] of T), store whole 256-byte rows per column block --------------
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const bool has_bias = a.bias != nullptr;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int p = p0 + wave * 16 + fg * 4 + r;
      p = p < P ? p : P - 1;
      bv[r] = has_bias ? E::to_f32(reinterpret_cast<const T*>(a.bias)[j * P + p]) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      T out[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[j][i][r];
        if (has_bias) v = E::to_f32(E::from_f32(v)) + bv[r];  // product rounded to T, then the bias, then rounded (tensor/function.py:45-46)
        out[r] = E::from_f32(v);
      }
      const int row = i * 16 + fr;
      *reinterpret_cast<uint2*>(smem + row * OUT_PITCH + (j * BP + wave * 16 + fg * 4) * 2) = *reinterpret_cast<const uint2*>(out);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  T* yg = reinterpret_cast<T*>(a.y);
  const bool full = (m0 + BM <= M) && (p0 + BP <= P) && (P % 8 == 0);
#pragma unroll 4
  for (int t = 0; t < BM * 32 / (NW * 64); ++t) {  // 32 chunks of 16 bytes per parked row, 16 rows per pass
    const int row = t * 16 + (tid >> 5), c16 = tid & 31;
    const uint4 v = *reinterpret_cast<const uint4*>(smem + row * OUT_PITCH + c16 * 16);
    const int m = m0 + row;
    const int j = c16 >> 4, pl = (c16 & 15) * 8;  // plane, first packed row of the chunk inside the tile
    const size_t n = (size_t)j * P + p0 + pl;
    if (full) {
      __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(yg + (size_t)m * N + n));
    } else if (m < M) {
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (p0 + pl + r < P) yg[(size_t)m * N + n + r] = e[r];
    }
  }
}

template <int DT, bool INT_SHIFT, bool FULLM, bool HG>
static int launch_m(const Args& a, hipStream_t stream) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N / 2 + BP - 1) / BP, tiles = tiles_m * tiles_n;
  Args b = a;
  {
    // per-XCD band of B tiles as a (g x B/g) rectangle: fetched bytes per k ~ g * BM * 2 (activations) + (B / g) * BP (packed weights)
    const int band = (tiles + 7) / 8;
    int g = 1;
    while ((g + 1) * (g + 1) * 2 * BM <= band * BP) ++g;
    const int forced = env_int("QUANTO_HIP_GROUP_M", 0);  // experiments
    if (forced > 0) g = forced;
    b.group_m = g < tiles_m ? g : tiles_m;
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbits_mfma_large_kernel<DT, INT_SHIFT, FULLM, HG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipLaunchKernelGGL((qbits_mfma_large_kernel<DT, INT_SHIFT, FULLM, HG>), dim3(tiles), dim3(NW * 64), LDS_BYTES, stream, b);
  return launch_status();
}
template <int DT, bool INT_SHIFT>
static int launch(const Args& a, hipStream_t stream) {
  if (a.C % BK != 0) return a.M % BM == 0 ? launch_m<DT, INT_SHIFT, true, true>(a, stream) : launch_m<DT, INT_SHIFT, false, true>(a, stream);
  return a.M % BM == 0 ? launch_m<DT, INT_SHIFT, true, false>(a, stream) : launch_m<DT, INT_SHIFT, false, false>(a, stream);
}

}  // namespace l4

bool qbits_mfma_large_supported(int64_t M, const PackedGeom& g, int dtype) {
  return g.bits == 4 && (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) && g.N % 2 == 0 && g.K % l4::BK == 0 && g.K >= 2 * l4::BK &&
         g.C % (l4::BK / 2) == 0 && M >= 1 && M * g.K < (1ll << 30) && g.N * g.K < (1ll << 32) && g.N * g.G < (1ll << 31) && g.N < (1 << 30) && M < (1 << 30);
}

int qbits_mm_mfma_large(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t M,
                        const PackedGeom& g, int dtype, bool int_shift, hipStream_t stream) {
  if (!qbits_mfma_large_supported(M, g, dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(y)) % 16) return QUANTO_HIP_EALIGN;
  l4::Args a{x, packed, scale, shift, bias, y, (int)M, (int)g.N, (int)g.K, (int)g.C, (int)g.G, 1};
  if (dtype == QUANTO_HIP_BF16)
    return int_shift ? l4::launch<QUANTO_HIP_BF16, true>(a, stream) : l4::launch<QUANTO_HIP_BF16, false>(a, stream);
  return int_shift ? l4::launch<QUANTO_HIP_F16, true>(a, stream) : l4::launch<QUANTO_HIP_F16, false>(a, stream);
}

}  // namespace qh
