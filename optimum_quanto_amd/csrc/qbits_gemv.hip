// qbits_mm for decode shapes (M <= 64, in passes of up to 8 rows of x): weight-streaming GEMV over the generic PackedTensor layout.
//
// HBM-bound.  With axis-0 grouping, group size 128 and N even, the packed tensor is simply
// P[N/2][K] bytes: byte (p, k) holds W[p, k] in its low nibble and W[p + N/2, k] in its high
// nibble (tensor/packed.py:24-69 + tensor/grouped.py:17-39), so every packed row is K contiguous
// bytes streamed with fully coalesced 16-byte loads.  The scale/shift of byte (p,k), plane h, is
// entry (p + h*N/2)*G + k/128.
//
// Work split (v2, latency-first: a decode call lasts a few microseconds, so everything a wave needs
// is requested in its first instructions):
//   * a wave owns K-slabs of 1024 byte-columns (slab = wave, wave+4, ...; at most ITERS of them) and
//     keeps only that slice of x in registers (8 VGPRs per slab and row of x);
//   * a block of 4 waves covers RR=4 packed rows per iteration: RR*ITERS 16-byte weight loads per lane
//     are in flight at once; per-slab partial results are combined through 64 bytes of LDS;
//   * scale/shift: lane l loads the 2-byte entries of row (l&3), group (l>>3) once per slab - four
//     tiny coalesced loads - and the four rows are fanned out inside each quad with DPP quad_perm.
//
// Arithmetic per 32-bit word of packed data (8 weights): 3 shifts + 4 v_and_or_b32 build four
// bf16x2 operands (128+q_a, 128+q_b) - 0x4300|q is exactly 128+q in bf16 - and 4 v_dot2c_f32_bf16
// accumulate them against pre-permuted x pairs; the +128 bias is cancelled by initialising each
// accumulator with -128*sum(x).  Per lane and group: y += scale * dot - shift * sum(x), fp32
// throughout, so the result is the exact-math value of the reference's integers/scales (no bf16
// rounding of W).  Wave reductions use DPP adds (no LDS traffic).
#include <cstdlib>
#include <type_traits>

#include "qh_common.h"

namespace qh {

template <int DT>
struct Dot2;
template <>
struct Dot2<QUANTO_HIP_BF16> {
  static constexpr uint32_t MAGIC = 0x43004300u;  // bf16 128.0 | q
  static constexpr float OFFSET = 128.f;
  static __device__ __forceinline__ float dot(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
  }
};
template <>
struct Dot2<QUANTO_HIP_F16> {
  static constexpr uint32_t MAGIC = 0x64006400u;  // fp16 1024.0 | q
  static constexpr float OFFSET = 1024.f;
  static __device__ __forceinline__ float dot(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
  }
};

template <int DT>
__device__ __forceinline__ float pair_lo(uint32_t p) {
  return Elem<DT>::to_f32(__builtin_bit_cast(typename Elem<DT>::T, (uint16_t)(p & 0xFFFFu)));
}
template <int DT>
__device__ __forceinline__ float pair_hi(uint32_t p) {
  return Elem<DT>::to_f32(__builtin_bit_cast(typename Elem<DT>::T, (uint16_t)(p >> 16)));
}

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// sum over the 64 lanes; only lane 63 holds the total
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v += dpp_f<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);       // row_half_mirror
  v += dpp_f<0x140>(v);       // row_mirror  -> every lane of a 16-lane row holds the row sum
  v += dpp_f<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v += dpp_f<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3 -> lane 63 holds the total
  return v;
}
template <int R>
__device__ __forceinline__ float quad_bcast(float v) {
  constexpr int ctrl = R | (R << 2) | (R << 4) | (R << 6);
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true));
}

constexpr int RR = 4;  // packed rows per wave pass

// r6: the load-order / ablation variants (QUANTO_HIP_GEMV_VARIANT = 1, 3, 4, 6) and the 8-rows-per-pass form (QUANTO_HIP_GEMV_RR=8), measured in r2 / r3
// and never the product choice, are compiled into probe builds only (-DQH_GEMV_EXPERIMENTS): 50 of this file's 258 instantiations
#ifdef QH_GEMV_EXPERIMENTS
constexpr bool QH_GEMV_EXPERIMENTS_ON = true;
#else
constexpr bool QH_GEMV_EXPERIMENTS_ON = false;
#endif

// Several Linears that share the same input (q/k/v, gate/up of a decoder layer) in ONE launch: a decode-shaped call lasts
// 4-8 us of which ~3 us are launch + first-byte latency, so every launch that disappears is worth about one small Linear.
// The weights stay separate allocations (the reference's modules are untouched): the kernel gets a table of segments and
// a block looks up which Linear it works for.
constexpr int MAX_SEGS = QUANTO_HIP_MAX_MULTI;
static_assert(MAX_SEGS == 4, "the segment lookup in the kernel compares against first_block[1..3]");
struct GemvSegs {
  const uint8_t* packed[MAX_SEGS];
  const uint16_t* scale[MAX_SEGS];
  const void* shift[MAX_SEGS];
  const uint16_t* bias[MAX_SEGS];
  uint16_t* y[MAX_SEGS];
  int N[MAX_SEGS];
  int first_block[MAX_SEGS];  // first workgroup of each segment (INT_MAX for unused slots)
};

// VARIANT (experiments, bf16 / M = 1 only; the product default is picked by the launcher):
//   bit 0: request x and the scale/shift entries BEFORE the weights.  Loads return in order: with the small L2-resident
//          loads at the tail of the queue, no row can be processed before the wave's last weight byte has landed; at the
//          head, row r is processed while rows r+1.. are still in flight.
//   bit 1: non-temporal weight loads (each weight byte is read exactly once per call).
template <int DT, int MT, int ITERS, bool INT_SHIFT, int VARIANT = 0, bool MULTI = false, int RRT = 4>
__global__ void __launch_bounds__(256)
    qbits_gemv_g128_kernel(const uint16_t* __restrict__ x, const GemvSegs segs, int K,
                           int wpr_log2 /* log2 of the waves cooperating on one row group: 0, 1 or 2 */,
                           int gshift /* VARIANT & 8 only: group of byte-column k = k >> gshift, or (k >> 5) / 3 when gshift < 0 */) {
  using E = Elem<DT>;
  using T = typename E::T;
  using D2 = Dot2<DT>;
  constexpr int RR = RRT;  // packed rows per wave pass: 4, or 8 for the long fused launches (r3, see the CLS flow below)
  // bit 4 (with bit 3): qint2 weights - four planes per byte (byte (p,k) = W[p,k] | W[p+N/4,k] << 2 | W[p+N/2,k] << 4 |
  // W[p+3N/4,k] << 6), 0x4300 | q is still exactly 128 + q: the same kernel with PL = 4 planes and a 2-bit mask
  constexpr bool INT2 = (VARIANT & 16) != 0;
  constexpr int BITS = INT2 ? 2 : 4, PL = 8 / BITS;
  static_assert(!INT2 || (VARIANT & 8) != 0, "the int2 variant uses the per-lane scale fetch");
  __shared__ float red[4][4][RR * PL * MT];  // [wave][16-lane row][value]
  constexpr bool EARLY_X = (VARIANT & 1) != 0, NT = (VARIANT & 2) != 0;
  constexpr bool ABLATE = (VARIANT & 4) != 0;  // measurement only (WRONG results): 1/4 of the arithmetic, all of the loads
  // bit 3: any group size the reference's QModuleMixin selects (nn/qmodule.py:121-129: 128, else 96 / 64 / 32 when in_features
  // is not a multiple of 128) and per-channel scales (group_size=None).  The packed layout is the same P[N/2][K] for every
  // group size (tensor/packed.py + tensor/grouped.py: grouped row n*G + kg, byte offset n*K + k); only the scale / shift index
  // changes: every lane fetches the entries of its own 16 byte-columns (16 divides every group size) for the block's RR rows,
  // instead of the quad-shared fetch that needs 64-byte-aligned groups.
  constexpr bool GEN_GS = (VARIANT & 8) != 0;

  // Prologue discipline: the call lasts a few microseconds, so nothing slow may sit in front of the first load - shifts
  // instead of divisions, and for the plain op (MULTI = false: segment 0, known at compile time) ONE round of scalar loads
  // for all kernel arguments.  The multi-Linear launch pays a second, dependent round for the selected segment's pointers
  // (cheap next to the launches it replaces).
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wpr = 1 << wpr_log2;
  const int slab0 = wave & (wpr - 1);   // first K-slab of this wave; further slabs at stride wpr
  const int rgroup = wave >> wpr_log2;  // which row group of the block
  const int bid = blockIdx.x;
  const int seg = MULTI ? (bid >= segs.first_block[1]) + (bid >= segs.first_block[2]) + (bid >= segs.first_block[3]) : 0;
  const uint8_t* __restrict__ packed = segs.packed[seg];
  const uint16_t* __restrict__ scale = segs.scale[seg];
  const void* __restrict__ shift_ = segs.shift[seg];
  const uint16_t* __restrict__ bias = segs.bias[seg];
  uint16_t* __restrict__ y = segs.y[seg];
  const int N = segs.N[seg];
  const int seg_first = MULTI ? segs.first_block[seg] : 0;
  const int P = INT2 ? N >> 2 : N >> 1;
  auto group_of = [&](int k) -> int { return gshift >= 0 ? (k >> gshift) : (int)(((uint32_t)(k >> 5) * 0xAAABu) >> 17); };
  const int G = GEN_GS ? group_of(K - 1) + 1 : K >> 7;
  const int p0 = (((bid - seg_first) << (2 - wpr_log2)) + rgroup) * RR;
  // ---- 1. request everything this wave will touch: weights, scales/shifts, x slice ---------------
  int k0[ITERS];
  bool valid[ITERS];
  uint4 W[RR][ITERS];
  constexpr int KEEP = RR * PL / 8;  // (row, plane) pairs a lane class is left with after the group-local halving (CLS flow)
  static_assert(RR * PL % 8 == 0, "the group-local halving leaves whole (row, plane) pairs");
  const int cls = ((lane & 1) << 2) | (lane & 2) | ((lane >> 2) & 1);
  float sq[ITERS][KEEP], zq[ITERS][KEEP];  // CLS flow (int4, group size 128): scale / shift of this lane class's pairs
  uint4 xa[ITERS][MT], xb[ITERS][MT];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    k0[it] = ((slab0 + it * wpr) * 64 + lane) * 16;
    valid[it] = k0[it] < K;
    k0[it] = valid[it] ? k0[it] : 0;  // out-of-range slabs read (and then ignore) the start of the row
  }
  auto load_weights = [&]() {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        // unconditional, clamped loads: rows beyond P are computed on duplicate data and never stored
        const int pr = p0 + r < P ? p0 + r : P - 1;
        const uint4* src = reinterpret_cast<const uint4*>(packed + (size_t)pr * K + k0[it]);
        if constexpr (NT) {
          typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
          const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
          W[r][it] = make_uint4(v.x, v.y, v.z, v.w);
        } else {
          W[r][it] = *src;
        }
      }
    }
  };
  float sg[GEN_GS ? ITERS : 1][RR][PL], zg[GEN_GS ? ITERS : 1][RR][PL];  // GEN_GS: this lane's own entries per row
  auto load_small = [&]() {
    if constexpr (GEN_GS) {
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int grp = group_of(k0[it]);
#pragma unroll
        for (int r = 0; r < RR; ++r) {
          const int pr = p0 + r < P ? p0 + r : P - 1;
#pragma unroll
          for (int h = 0; h < PL; ++h) {
            const size_t idx = (size_t)(pr + h * P) * G + grp;
            sg[it][r][h] = E::to_f32(__builtin_bit_cast(T, scale[idx]));
            if constexpr (INT_SHIFT)
              zg[it][r][h] = sg[it][r][h] * (float)(int8_t) reinterpret_cast<const uint8_t*>(shift_)[idx];
            else
              zg[it][r][h] = E::to_f32(__builtin_bit_cast(T, reinterpret_cast<const uint16_t*>(shift_)[idx]));
          }
        }
      }
    } else {
      // CLS flow (r3): the 8 lanes of a 128-byte group reduce their dot products among themselves BEFORE scale and shift are applied
      // (halving over lane bits 0..2, section 2), after which lane class c = b0*4 + b1*2 + b2 is left with the sums of the (row, plane)
      // pairs rp = c*KEEP + j: this lane fetches the entries of exactly those, for the group of its 16 bytes
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int j = 0; j < KEEP; ++j) {
          const int rp = cls * KEEP + j, r = rp / PL, h = rp % PL;
          const int pr = p0 + r < P ? p0 + r : P - 1;
          const size_t idx = (size_t)(pr + h * P) * G + (k0[it] >> 7);
          sq[it][j] = E::to_f32(__builtin_bit_cast(T, scale[idx]));
          if constexpr (INT_SHIFT)
            zq[it][j] = sq[it][j] * (float)(int8_t) reinterpret_cast<const uint8_t*>(shift_)[idx];
          else
            zq[it][j] = E::to_f32(__builtin_bit_cast(T, reinterpret_cast<const uint16_t*>(shift_)[idx]));
        }
      }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint4* px = reinterpret_cast<const uint4*>(x + (size_t)m * K + k0[it]);
        xa[it][m] = px[0];
        xb[it][m] = px[1];
      }
    }
  };
  if constexpr (EARLY_X) {
    load_small();
    // keep the issue order: without this the scheduler is free to hoist the (independent) weight loads back to the front
    __builtin_amdgcn_sched_barrier(0);
    load_weights();
  } else {
    load_weights();
    load_small();
  }
  uint32_t X02[ITERS][MT][4], X13[ITERS][MT][4];
  float xs[ITERS][MT];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const uint4 a = xa[it][m], b = xb[it][m];
      const uint32_t keep = valid[it] ? 0xFFFFFFFFu : 0u;  // x = 0 beyond K: such a slab contributes exactly 0
      const uint32_t pr[8] = {a.x & keep, a.y & keep, a.z & keep, a.w & keep, b.x & keep, b.y & keep, b.z & keep, b.w & keep};
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        X02[it][m][d] = __builtin_amdgcn_perm(pr[2 * d + 1], pr[2 * d], 0x05040100u);
        X13[it][m][d] = __builtin_amdgcn_perm(pr[2 * d + 1], pr[2 * d], 0x07060302u);
        s += pair_lo<DT>(pr[2 * d]) + pair_hi<DT>(pr[2 * d]) + pair_lo<DT>(pr[2 * d + 1]) + pair_hi<DT>(pr[2 * d + 1]);
      }
      xs[it][m] = s;
    }
  }

  // ---- 2. per row: dot products, scale, shift ------------------------------------------------------
  // v_and_or_b32 is VOP3: no literal operands on gfx9, at most one SGPR.  Keep the mask in an SGPR and the magic
  // exponent in a VGPR, opaque to the constant folder, so that (w & mask) | magic is ONE instruction.
  uint32_t kmask = INT2 ? 0x00030003u : 0x000F000Fu, kmagic = D2::MAGIC;
  asm volatile("" : "+s"(kmask));
  asm volatile("" : "+v"(kmagic));
  // NV = RR * PL * MT partial sums per lane, value index (r * PL + h) * MT + m.  They are reduced TOGETHER (r3): reducing each of them
  // over the 64 lanes on its own is 6 DPP steps per value (with hipcc's SLP vectorizer in the way: v_mov_b32_dpp + v_pk_add_f32 + the
  // moves that build the pairs - ~160 of the 565 instructions of an M = 1 wave, in a kernel whose fused gate+up launch is VALU-bound:
  // 14 waves per SIMD x ~450 VALU x 4.5 cycles = its 13.9 us).  Halving instead: at step j a lane keeps the half of its values
  // selected by bit j of its lane id, adds its partner's copy of that half (the lane that differs in exactly that bit) and forgets the
  // other half - NV/2 + NV/4 + ... value-steps instead of 6 NV.  After the (up to four) steps inside a 16-lane DPP row the rows' sums
  // meet in LDS, where the waves of a row group are combined anyway.
  constexpr int NV = RR * PL * MT;
  static_assert(NV >= 8 && NV <= 64, "4 or 8 rows x at least 2 planes; one lane per output of the block's row group");
  constexpr int HS = NV >= 16 ? 4 : 3;  // halving steps
  constexpr int NKEEP = NV >> HS;       // values a lane is left with
  auto halve = [&](float (&v)[NV], auto step_tag, auto n_tag) {
    constexpr int STEP = decltype(step_tag)::value, N = decltype(n_tag)::value, H = N / 2;
    // the partner differs in bit STEP of the lane id and in nothing else: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_shl:4 (for the
    // lanes whose bit is clear) / row_shr:4 (bit set), row_ror:8.  (row_half_mirror / row_mirror flip the lower bits as well: the
    // partner would hold another half of the values.)
    constexpr int CTRL_LO = STEP == 0 ? 0xB1 : (STEP == 1 ? 0x4E : (STEP == 2 ? 0x104 : 0x128));
    constexpr int CTRL_HI = STEP == 0 ? 0xB1 : (STEP == 1 ? 0x4E : (STEP == 2 ? 0x114 : 0x128));
    const bool up = (lane >> STEP) & 1;
#pragma unroll
    for (int i = 0; i < H; ++i) {
      // both candidate sums (each one v_add_f32_dpp: the DPP source is a plain register), then the lane's pick
      const float lo = v[i] + dpp_f<CTRL_LO>(v[i]), hi = v[i + H] + dpp_f<CTRL_HI>(v[i + H]);
      v[i] = up ? hi : lo;
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  auto halve_group = [&](float (&v)[NV]) {  // lane bits 0..2: the 8 lanes that hold one 128-byte group of a row
    halve(v, I0{}, std::integral_constant<int, NV>{});
    halve(v, I1{}, std::integral_constant<int, NV / 2>{});
    halve(v, I2{}, std::integral_constant<int, NV / 4>{});
  };

  float vals[NV];  // GEN_GS: scaled sums of all values; CLS: [0, NV/8) scaled sums of this lane class's values (bits 0..2 already reduced)
#pragma unroll
  for (int v = 0; v < NV; ++v) vals[v] = 0.f;

#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    float dots[NV];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      float dot[PL][MT];
#pragma unroll
      for (int h = 0; h < PL; ++h)
#pragma unroll
        for (int m = 0; m < MT; ++m) dot[h][m] = -D2::OFFSET * xs[it][m];
      const uint32_t w4[4] = {W[r][it].x, W[r][it].y, W[r][it].z, W[r][it].w};
      if constexpr (ABLATE) asm volatile("" ::"v"(w4[1]), "v"(w4[2]), "v"(w4[3]));
#pragma unroll
      for (int d = 0; d < (ABLATE ? 1 : 4); ++d) {
        const uint32_t w = w4[d];
        // plane h of bytes 0 / 2 and of bytes 1 / 3 as (128 + q_a, 128 + q_b) pairs: one shift + one v_and_or each
        uint32_t a02[PL], a13[PL];
#pragma unroll
        for (int h = 0; h < PL; ++h) {
          a02[h] = ((h == 0 ? w : w >> (BITS * h)) & kmask) | kmagic;
          a13[h] = ((w >> (8 + BITS * h)) & kmask) | kmagic;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int h = 0; h < PL; ++h) {
            dot[h][m] = D2::dot(a02[h], X02[it][m][d], dot[h][m]);
            dot[h][m] = D2::dot(a13[h], X13[it][m][d], dot[h][m]);
          }
        }
      }
#pragma unroll
      for (int h = 0; h < PL; ++h)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int v = (r * PL + h) * MT + m;
          if constexpr (GEN_GS) {  // this lane's own scale / shift entries: scale first, reduce at the end
            vals[v] = __builtin_fmaf(sg[it][r][h], dot[h][m], vals[v]);
            vals[v] = __builtin_fmaf(-zg[it][r][h], xs[it][m], vals[v]);
          } else {
            dots[v] = dot[h][m];
          }
        }
    }
    if constexpr (!GEN_GS) {
      // CLS flow: the 8 lanes of the group add their dot products and their sums of x, then ONE scale / shift application per
      // kept value instead of one per lane and value (and no fan-out of the scales through the quad)
      halve_group(dots);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float xg = xs[it][m];
        xg += dpp_f<0xB1>(xg);
        xg += dpp_f<0x4E>(xg);
        xg += dpp_f<0x141>(xg);  // row_half_mirror: a lane of the other quad, which holds that quad's sum
#pragma unroll
        for (int j = 0; j < KEEP; ++j) {
          vals[j * MT + m] = __builtin_fmaf(sq[it][j], dots[j * MT + m], vals[j * MT + m]);
          vals[j * MT + m] = __builtin_fmaf(-zq[it][j], xg, vals[j * MT + m]);
        }
      }
    }
  }

  // ---- 3. finish the reduction over lanes (DPP), over the wpr waves (LDS), store -----------------------------------
  if constexpr (GEN_GS) halve_group(vals);
  if constexpr (HS >= 4) halve(vals, I3{}, std::integral_constant<int, NV / 8>{});
  // lane l now holds, in vals[0 .. NKEEP), the sums over its HS-bit lane class of the values base(l) + i, base = sum_j bit_j(l) * (NV >> (j+1));
  // NV = 8: bit 3 of the lane id is still to be summed - one all-reduce step with the lane that differs in that bit only
  if constexpr (HS < 4) vals[0] += dpp_f<0x128>(vals[0]);
  {
    int base = 0;
#pragma unroll
    for (int j = 0; j < HS; ++j) base += ((lane >> j) & 1) * (NV >> (j + 1));
    const bool writer = HS >= 4 || (lane & 15) < (1 << HS);  // one lane per class and row
    if (writer) {
#pragma unroll
      for (int i = 0; i < NKEEP; ++i) red[wave][lane >> 4][base + i] = vals[i];
    }
  }
  __syncthreads();
  static_assert(NV <= 64, "one lane per output of the block's row group");
  if (slab0 == 0 && lane < NV) {
    const int r = lane / (PL * MT), h = (lane / MT) % PL, m = lane % MT;
    const int p = p0 + r;
    if (p < P) {
      float v = 0.f;
      for (int s = 0; s < wpr; ++s)
#pragma unroll
        for (int dr = 0; dr < 4; ++dr) v += red[wave + s][dr][lane];
      const int n = p + h * P;
      if (bias) v = E::to_f32(E::from_f32(v)) + E::to_f32(__builtin_bit_cast(T, bias[n]));
      y[(size_t)m * N + n] = __builtin_bit_cast(uint16_t, E::from_f32(v));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side: `nseg` Linears sharing x (nseg = 1 for the plain op)
struct GemvProblem {
  int nseg;
  const uint8_t* packed[MAX_SEGS];
  const void* scale[MAX_SEGS];
  const void* shift[MAX_SEGS];
  const void* bias[MAX_SEGS];
  void* y[MAX_SEGS];
  int N[MAX_SEGS];
  int gs;    // group size: 128 (quad-shared scale fetch), 32 / 64 / 96, or 0 = per-channel
  int bits;  // 4, or 2 (always on the per-lane scale fetch)
};

static int gemv_variant() {
  // experiments: QUANTO_HIP_GEMV_VARIANT = bit 0 (x / scales requested first) | bit 1 (non-temporal weight loads) |
  // bit 2 (ablation, wrong results: a quarter of the arithmetic - tells how much of a call is VALU work)
  const int v = env_int("QUANTO_HIP_GEMV_VARIANT", QUANTO_HIP_GEMV_DEFAULT_VARIANT) & 7;
  return v;
}

template <int DT, int MT, bool INT_SHIFT>
static int gemv_launch_iters(const void* x, const GemvProblem& pb, int m0, int K, hipStream_t stream) {
  const int nslab = (K + 1023) / 1024;
  const int wpr = nslab >= 3 ? 4 : nslab;  // 1, 2 or 4 waves per row group
  const int wpr_log2 = wpr == 4 ? 2 : wpr - 1;
  const int iters = (nslab + wpr - 1) / wpr;
  if (iters > 4) return QUANTO_HIP_ENOTSUP;
  // packed rows per wave pass.  8 rows cut the instructions per weight byte by a quarter (612 per 8 KiB wave against 387 per 4 KiB:
  // x slice, addresses and reduction tail are per wave) but halve the waves, and a wave is one batch of loads followed by its
  // arithmetic: r3 A/B, us with 4 / 8 rows: north-star 4.06 / 4.00, cfg3 6.48 / 6.45, q/k/v in one launch 4.99 / 5.91, gate+up in one
  // launch 12.0 / 12.9.  4 rows stay the product form; 8 is kept behind QUANTO_HIP_GEMV_RR=8 (parity-tested, same bits).
  const bool rr8_ok = pb.gs == 128 && pb.bits == 4 && MT <= 2 && iters == 1;
  const int rr = rr8_ok && env_int("QUANTO_HIP_GEMV_RR", 0) == 8 ? 8 : RR;
  const int rows_per_block = rr * (4 / wpr);
  GemvSegs segs;
  int grid = 0;
  for (int i = 0; i < MAX_SEGS; ++i) {
    const int j = i < pb.nseg ? i : 0;  // unused slots repeat segment 0 and are never selected
    segs.packed[i] = pb.packed[j];
    segs.scale[i] = reinterpret_cast<const uint16_t*>(pb.scale[j]);
    segs.shift[i] = pb.shift[j];
    segs.bias[i] = reinterpret_cast<const uint16_t*>(pb.bias[j]);
    segs.y[i] = reinterpret_cast<uint16_t*>(pb.y[j]) + (size_t)m0 * pb.N[j];
    segs.N[i] = pb.N[j];
    segs.first_block[i] = i < pb.nseg ? grid : 0x7FFFFFFF;
    if (i < pb.nseg) grid += (pb.N[i] / (8 / pb.bits) + rows_per_block - 1) / rows_per_block;
  }
  auto xs = reinterpret_cast<const uint16_t*>(x) + (size_t)m0 * K;
  // group -> shift: 32 -> 5, 64 -> 6, 128 -> 7, per-channel -> 30 (always group 0), 96 -> -1 ((k >> 5) / 3)
  const int gshift = pb.gs == 0 ? 30 : pb.gs == 96 ? -1 : pb.gs == 32 ? 5 : pb.gs == 64 ? 6 : 7;
#define QH_LAUNCH_VM(IT, V, MULTI)                                                                                                     \
  do {                                                                                                                                \
    if constexpr (QH_GEMV_EXPERIMENTS_ON && MT <= 2 && IT == 1 && ((V) == 0 || (V) == 2)) {                                            \
      if (rr == 8) {                                                                                                                  \
        hipLaunchKernelGGL((qbits_gemv_g128_kernel<DT, MT, IT, INT_SHIFT, V, MULTI, 8>), dim3(grid), dim3(256), 0, stream, xs, segs, K, \
                           wpr_log2, gshift);                                                                                         \
        break;                                                                                                                        \
      }                                                                                                                               \
    }                                                                                                                                 \
    hipLaunchKernelGGL((qbits_gemv_g128_kernel<DT, MT, IT, INT_SHIFT, V, MULTI>), dim3(grid), dim3(256), 0, stream, xs, segs, K,        \
                       wpr_log2, gshift);                                                                                             \
  } while (0)
#define QH_LAUNCH_V(IT, V)                    \
  do {                                        \
    if constexpr (MT <= 4) {                  \
      if (pb.nseg > 1)                        \
        QH_LAUNCH_VM(IT, V, true);            \
      else                                    \
        QH_LAUNCH_VM(IT, V, false);           \
    } else {                                  \
      QH_LAUNCH_VM(IT, V, false);             \
    }                                         \
  } while (0)
  // the load-order / cache-policy variants exist for the decode configuration the bench measures (bf16, float shift, M = 1)
  constexpr bool HAS_VARIANTS = DT == QUANTO_HIP_BF16 && MT == 1 && !INT_SHIFT;
#define QH_LAUNCH(IT)                           \
  do {                                          \
    if (pb.bits == 2) {                         \
      if constexpr (MT <= 4) QH_LAUNCH_VM(IT, 24, false); \
    } else if (pb.gs != 128) {                  \
      if constexpr (MT <= 4) QH_LAUNCH_VM(IT, 8, false); \
    } else if constexpr (HAS_VARIANTS && QH_GEMV_EXPERIMENTS_ON) { \
      switch (gemv_variant()) {                 \
        case 1: QH_LAUNCH_V(IT, 1); break;      \
        case 2: QH_LAUNCH_V(IT, 2); break;      \
        case 3: QH_LAUNCH_V(IT, 3); break;      \
        case 4: QH_LAUNCH_V(IT, 4); break;      \
        case 6: QH_LAUNCH_V(IT, 6); break;      \
        default: QH_LAUNCH_V(IT, 0); break;     \
      }                                         \
    } else if constexpr (HAS_VARIANTS) {        \
      QH_LAUNCH_V(IT, QUANTO_HIP_GEMV_DEFAULT_VARIANT); /* non-temporal weight loads */ \
    } else {                                    \
      QH_LAUNCH_V(IT, 0);                       \
    }                                           \
  } while (0)
  if constexpr (MT == 8) {
    switch (iters) {
      case 1: QH_LAUNCH(1); break;
      case 2: QH_LAUNCH(2); break;
      default: return QUANTO_HIP_ENOTSUP;
    }
  } else {
    switch (iters) {
      case 1: QH_LAUNCH(1); break;
      case 2: QH_LAUNCH(2); break;
      case 3: QH_LAUNCH(3); break;
      case 4: QH_LAUNCH(4); break;
    }
  }
#undef QH_LAUNCH
#undef QH_LAUNCH_V
#undef QH_LAUNCH_VM
  if ((pb.nseg > 1 || pb.gs != 128 || pb.bits != 4) && MT > 4) return QUANTO_HIP_ENOTSUP;  // not reachable: those calls are limited to M <= 4
  return launch_status();
}

template <int DT, bool INT_SHIFT>
static int gemv_launch_m(const void* x, const GemvProblem& pb, int M, int K, hipStream_t stream) {
  // Rows of x are processed in passes of at most 8 (x lives in registers: 8 VGPRs per slab and row); the weights of
  // later passes come from the Infinity Cache (a Linear's packed weight is 8-30 MB).  8 rows per pass need
  // iters <= 2 slabs per wave (K <= 8192) to stay inside the register file.
  const int nslab = (K + 1023) / 1024;
  const int iters = (nslab + (nslab >= 3 ? 4 : nslab) - 1) / (nslab >= 3 ? 4 : nslab);
  const int mt_max = (pb.gs != 128 || pb.bits != 4 || pb.nseg > 1) ? 4 : (iters <= 2 ? 8 : 4);  // the per-lane scale fetch variants exist for <= 4 rows
  int m0 = 0;
  while (m0 < M) {
    const int left = M - m0;
    const int mt = (left >= 8 && mt_max >= 8) ? 8 : (left >= 4 ? 4 : (left >= 2 ? 2 : 1));
    int st;
    if (mt == 8)
      st = gemv_launch_iters<DT, 8, INT_SHIFT>(x, pb, m0, K, stream);
    else if (mt == 4)
      st = gemv_launch_iters<DT, 4, INT_SHIFT>(x, pb, m0, K, stream);
    else if (mt == 2)
      st = gemv_launch_iters<DT, 2, INT_SHIFT>(x, pb, m0, K, stream);
    else
      st = gemv_launch_iters<DT, 1, INT_SHIFT>(x, pb, m0, K, stream);
    if (st != QUANTO_HIP_OK) return st;
    m0 += mt;
  }
  return QUANTO_HIP_OK;
}

bool qbits_gemv_supported(int64_t M, const PackedGeom& g, int dtype) {
  const bool common = (g.N % g.vpi == 0) && g.K <= 16384 && M >= 1 && (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) && g.N < (1 << 30);
  if (g.bits == 4 && g.C == 128) return common && (g.K % 128 == 0) && M <= QUANTO_HIP_GEMV_MAX_M_QBITS;
  // the other group sizes of nn/qmodule.py:121-129, per-channel scales and qint2: decode-sized calls, in passes of 4 rows up to 24
  // rows (r3; ~5-6 us per pass for a 4096^2 weight streamed from the Infinity Cache, against ~55 us for dequantize + dense GEMM,
  // which these formats otherwise take for every M > 4)
  const bool other = g.C == 32 || g.C == 64 || g.C == 96 || g.C == 128 || (g.C == g.K && g.K % 16 == 0);
  return common && other && (g.K % g.C == 0) && M <= QUANTO_HIP_GEMV_MAX_M_OTHER;
}

static int gemv_dispatch(const void* x, const GemvProblem& pb, int M, int K, int dtype, bool int_shift, hipStream_t stream) {
  if (dtype == QUANTO_HIP_BF16)
    return int_shift ? gemv_launch_m<QUANTO_HIP_BF16, true>(x, pb, M, K, stream) : gemv_launch_m<QUANTO_HIP_BF16, false>(x, pb, M, K, stream);
  return int_shift ? gemv_launch_m<QUANTO_HIP_F16, true>(x, pb, M, K, stream) : gemv_launch_m<QUANTO_HIP_F16, false>(x, pb, M, K, stream);
}

int qbits_mm_gemv(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t M,
                  const PackedGeom& g, int dtype, bool int_shift, hipStream_t stream) {
  if (!qbits_gemv_supported(M, g, dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed)) % 16) return QUANTO_HIP_EALIGN;
  GemvProblem pb{};
  pb.nseg = 1;
  pb.packed[0] = packed;
  pb.scale[0] = scale;
  pb.shift[0] = shift;
  pb.bias[0] = bias;
  pb.y[0] = y;
  pb.N[0] = (int)g.N;
  pb.gs = g.C == g.K && g.C != 128 ? 0 : (int)g.C;
  pb.bits = g.bits;
  return gemv_dispatch(x, pb, (int)M, (int)g.K, dtype, int_shift, stream);
}

// nseg (2..QUANTO_HIP_MAX_MULTI) int4 g128 Linears with the same K applied to the same x in one launch per pass of rows.
// The caller has checked qbits_gemv_supported for every segment.
int qbits_mm_gemv_multi(const void* x, int nseg, const uint8_t* const* packed, const void* const* scale, const void* const* shift,
                        const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int dtype, bool int_shift,
                        hipStream_t stream) {
  if (nseg < 1 || nseg > MAX_SEGS) return QUANTO_HIP_EINVAL;
  GemvProblem pb{};
  pb.nseg = nseg;
  pb.gs = 128;
  pb.bits = 4;
  uintptr_t align = reinterpret_cast<uintptr_t>(x);
  for (int i = 0; i < nseg; ++i) {
    pb.packed[i] = packed[i];
    pb.scale[i] = scale[i];
    pb.shift[i] = shift[i];
    pb.bias[i] = bias ? bias[i] : nullptr;
    pb.y[i] = y[i];
    pb.N[i] = (int)N[i];
    align |= reinterpret_cast<uintptr_t>(packed[i]);
  }
  if (align % 16) return QUANTO_HIP_EALIGN;
  return gemv_dispatch(x, pb, (int)M, (int)K, dtype, int_shift, stream);
}

}  // namespace qh
