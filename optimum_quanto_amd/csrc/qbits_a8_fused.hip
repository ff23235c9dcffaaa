// qbits_mm with QUANTIZED activations (W4A8, r6): int4 weights x int8 activations on v_mfma_i32_16x16x64_i8 and int4 weights x fp8-e4m3 activations on
// v_mfma_scale_f32_16x16x128_f8f6f4 - the 8-bit matrix rates (2 x bf16) for the one activation x weight combination of the reference's
// tests/tensor/ops/test_linear_dispatch.py:22-42 that still dequantized its activation (tensor/weights/awq/qbits.py:57-58 does the same on CUDA;
// BASELINE.md lists the configuration: "W int4 / A fp8").
//
//   y[m, n] = s_x * sum_g ( s[n,g] * sum_{k in g} a[m,k] q[n,k]  -  z[n,g] * sum_{k in g} a[m,k] ) (+ bias[n])
//
// a = the stored int8 / e4m3 activation values (per-tensor scale s_x, tensor/activations/qbytes.py:28-43), q = the stored nibbles 0..15,
// z = shift (float shifts) or s * zero_point.  Products of the stored values are exact (int32 accumulation for int8; an e4m3 value times an integer
// below 16 is exact in fp32), scale / shift are applied to the fp32 accumulator per group in group order, s_x once at the end - the arithmetic contract of
// the other fused kernels.  For int8 activations the result is a pure function of the integers: bit-identical to an fp32 fma chain
// over the exact group sums (oracle/quanto_oracle.py::qbits_mm_a8_chain), whichever tile or split computes it in the unsplit form.
//
// Structure = qbits_mfma_fused.hip (workgroup = 8 waves, BM tokens x 64 packed rows = 128 features, wave = all BM tokens x 16 features, K-tile = one group
// of 128, LDS-DMA ring of two stages, scale tables parked in LDS, group accumulators double-buffered so that the fold of tile kt-1 is sliced over the matrix
// steps of tile kt, split-K with a deterministic last-arriver reduce), with 1-byte activations (128-byte LDS rows) and:
//   * int8: a weight operand is the lane's nibble plane of 16 packed bytes - ((raw >> 4 plane) & 0x0F0F0F0F), TWO VALU per four weights (the bf16 kernel: one
//     per weight) -, two K = 64 MFMAs per fragment and group, int32 group accumulator -> v_cvt_f32_i32 + two FMAs in the fold;
//   * fp8: nibbles -> e4m3 codes through a 16-entry byte table (two v_perm over the low / high half of the table + one v_perm that picks by bit 3: seven
//     VALU per four weights), ONE K = 128 MX-format MFMA (unit block scales) per fragment and group;
//   * the group sums of a come from the matrix pipe as well (an all-ones weight operand, wave w for token fragment w), exact.
#include <type_traits>

#include "qh_common.h"

namespace qh {
namespace a8 {

constexpr int BK = 128, PR = 64, WAVES = 8, STAGES = 2;
constexpr int W_BYTES = PR * BK;  // 8 KiB of packed bytes per tile
template <int BM>
struct Geo {
  static constexpr int MI = BM / 16;
  static constexpr int X_BYTES = BM * BK, STAGE_BYTES = X_BYTES + W_BYTES;
  static constexpr int XP = BM / 8 / WAVES;  // activation DMA pieces (8 rows x 128 B = 1 KiB) per wave and tile
  static constexpr int OPS = XP + 1;
  static_assert(W_BYTES == WAVES * 1024 && XP >= 1 && (MI == 4 || MI == 8), "tile geometry");
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile(  // M0 is written and not restored (qmm_large_common.h: nothing else in this kernel needs it)
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

enum { A_I8 = 0, A_F8E4M3 = 1 };

struct Args {
  const uint8_t* a;      // [M, K] int8 / e4m3 activation values
  const void* a_scale;   // device scalar of the output dtype: the per-tensor activation scale
  const uint8_t* w;      // packed [N/2, K]
  const void* scale;     // [N*G]
  const void* shift;     // [N*G]
  const void* bias;      // [N] or null
  void* y;               // [M, N]
  int M, N, K, G;
  int S;                 // K split: blockIdx.z handles groups [z * G / S, (z + 1) * G / S)
  int* counters;         // [tiles] arrival counters, zero on entry and on exit (S > 1)
  float* partials;       // [tiles][S][MI][512 lanes] float4
  int ablate;            // QUANTO_HIP_A8_ABLATE (timing experiments, WRONG results): 1 no fold, 2 no matrix steps, 4 the DMA re-reads tile 0, 8 no weight unpack
};

template <int AK>
struct Acc {
  using V = f32x4;
};
template <>
struct Acc<A_I8> {
  using V = i32x4;
};

template <int DT, int AK, bool INT_SHIFT, int BM>
__global__ void __launch_bounds__(WAVES * 64, 1) qbits_a8_fused_kernel(const Args a) {
  using E = Elem<DT>;
  using T = typename E::T;
  using GV = typename Acc<AK>::V;  // group accumulator
  constexpr int MI = Geo<BM>::MI, XP = Geo<BM>::XP, X_BYTES = Geo<BM>::X_BYTES, STAGE_BYTES = Geo<BM>::STAGE_BYTES;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // layout: [STAGES x (activation tile | weight tile)] [xs: 2 x BM fp32 group sums of a] [sz: G x 2 x 128 features of T]
  float* xs_slot = reinterpret_cast<float*>(smem + STAGES * STAGE_BYTES);
  constexpr int NF = 2 * PR;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M, N = a.N, K = a.K, G = a.G;
  const int P = N >> 1;
  const int p0 = blockIdx.x * PR, m0 = blockIdx.y * BM;
  const int S = a.S, sp = blockIdx.z;
  const int nk = G / S;  // one tile per group; this workgroup's groups are kt0 .. kt0 + nk - 1
  const int kt0 = sp * nk;
  const int fi = lane & 15, fg = lane >> 4;
  T* sz = reinterpret_cast<T*>(smem + STAGES * STAGE_BYTES + 2 * BM * 4);

  // ---- DMA sources: activation pieces of 8 rows x 128 B (lane -> row lane >> 3, position lane & 7 holds chunk pos ^ (row & 7)), one weight piece per wave
  uint32_t xsrc[XP];
#pragma unroll
  for (int u = 0; u < XP; ++u) {
    const int row = 8 * (wave * XP + u) + (lane >> 3);
    const int c = (lane & 7) ^ (row & 7);
    int m = m0 + row;
    m = m < M ? m : M - 1;
    xsrc[u] = (uint32_t)((size_t)m * K + c * 16);  // M * K < 4 GiB, checked by the launcher
  }
  uint32_t wsrc;
  {
    const int r = wave * 8 + (lane >> 3), c = (lane & 7) ^ (r & 7);
    int p = p0 + r;
    p = p < P ? p : P - 1;
    wsrc = (uint32_t)((size_t)p * K + c * 16);
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  auto issue_tile = [&](int kt_tile, int stage) {
    if (a.ablate & 4) kt_tile = 0;
    const uint32_t st = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE_BYTES);
#pragma unroll
    for (int u = 0; u < XP; ++u) glds16(a.a + (size_t)(kt0 + kt_tile) * BK, xsrc[u], st + (wave * XP + u) * 1024);
    glds16(a.w + (size_t)(kt0 + kt_tile) * BK, wsrc, st + X_BYTES + wave * 1024);
  };
  const int last = nk - 1;

  // ---- prologue: tiles 0 and 1 requested, tables parked (thread -> feature tid & 127, groups tid >> 7, + 4, ...), ONE drain ----
  issue_tile(0, 0);
  issue_tile(nk > 1 ? 1 : 0, 1);
  {
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
  const float sx = E::to_f32(*reinterpret_cast<const T*>(a.a_scale));
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // hand-counted waits start from a known state

  // ---- fragment read offsets: 16-byte chunk 4 h + fg of the lane's row (h = half of the 128-byte row); (row & 7) == (fi & 7) for every fragment ----
  int xoff[2], woff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) xoff[h] = fi * 128 + (((4 * h + fg) ^ (fi & 7)) << 4);
  {
    const int r = wave * 8 + (fi & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) woff[h] = X_BYTES + r * 128 + (((4 * h + fg) ^ (r & 7)) << 4);
  }
  const uint32_t nib_shift = (fi >> 3) * 4;
  const int floc = (fg >> 1) * PR + wave * 8 + 4 * (fg & 1);  // the lane's 4 consecutive features inside the block

  f32x4 acc[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint32_t nibmask = 0x0F0F0F0Fu;
  asm volatile("" : "+s"(nibmask));
  // e4m3 codes of 0..15 (bias 7): 0, 1 = 0x38, 2 = 0x40, 3 = 0x44, 4..7 = 0x48 + 2 (q - 4), 8..15 = 0x50 + (q - 8)
  uint32_t t_lo0 = 0x44403800u, t_lo1 = 0x4E4C4A48u, t_hi0 = 0x53525150u, t_hi1 = 0x57565554u;
  asm volatile("" : "+v"(t_lo0), "+v"(t_lo1), "+v"(t_hi0), "+v"(t_hi1));

  GV accgA[MI], accgB[MI];
  float s4[4], z4[4], xsp[MI];
  const int my_xs = wave < MI ? wave : -1;
  auto load_sz = [&](int g) {
    T s4t[4], z4t[4];
    *reinterpret_cast<uint2*>(s4t) = *reinterpret_cast<const uint2*>(sz + (g * 2 + 0) * NF + floc);
    *reinterpret_cast<uint2*>(z4t) = *reinterpret_cast<const uint2*>(sz + (g * 2 + 1) * NF + floc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s4[r] = E::to_f32(s4t[r]);
      const float z = E::to_f32(z4t[r]);
      z4[r] = INT_SHIFT ? s4[r] * z : z;  // scale * (q - zp) = scale * q - (scale * zp): one fp32 rounding of the product, stated in the oracle
    }
  };
  auto load_xs = [&](int kt_prev) {
#pragma unroll
    for (int i = 0; i < MI; ++i) xsp[i] = xs_slot[(kt_prev & 1) * BM + i * 16 + fi];
  };
  // one element of the fold of a group: acc = fma(-z, A_g, fma(s, P_g, acc)) - as asm so that hipcc neither sinks the fold behind the matrix steps
  // nor packs it (qbits_mfma_fused.hip)
  auto fold_slice = [&](const GV (&pg)[MI], int q) {
    const int i = q >> 2, r = q & 3;
    float v = acc[i][r];
    if constexpr (AK == A_I8) {
      float p;
      asm volatile("v_cvt_f32_i32 %1, %2\n\tv_fmac_f32 %0, %3, %1\n\tv_fma_f32 %0, -%4, %5, %0"
                   : "+v"(v), "=&v"(p)
                   : "v"(pg[i][r]), "v"(s4[r]), "v"(z4[r]), "v"(xsp[i]));
    } else {
      asm volatile("v_fmac_f32 %0, %1, %2\n\tv_fma_f32 %0, -%3, %4, %0" : "+v"(v) : "v"(s4[r]), "v"(pg[i][r]), "v"(z4[r]), "v"(xsp[i]));
    }
    acc[i][r] = v;
  };

  constexpr int STEPS = AK == A_I8 ? 2 * MI : MI;   // matrix steps per tile
  constexpr int FPS = 4 * MI / STEPS;               // fold elements per step: 2 (int8) / 4 (fp8)
  auto tile = [&](int kt, GV (&cg)[MI], const GV (&pg)[MI], auto have_prev_tag, auto stage_tag) {
    constexpr bool have_prev = decltype(have_prev_tag)::value;
    constexpr int stage = decltype(stage_tag)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // tile kt (requested a tile ago; nothing younger is in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragment / table reads and the sum store of the previous tile
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_tile(kt + 1 < nk ? kt + 1 : last, 1 - stage);
    const uint8_t* st = smem + stage * STAGE_BYTES;
    uint4 w[2];
    w[0] = *reinterpret_cast<const uint4*>(st + woff[0]);
    w[1] = *reinterpret_cast<const uint4*>(st + woff[1]);
    // activation fragments: both 16-byte halves of every token fragment, two fragments ahead of their matrix step
    uint4 xl[MI], xh[MI];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      xl[i] = *reinterpret_cast<const uint4*>(st + xoff[0] + i * 2048);
      xh[i] = *reinterpret_cast<const uint4*>(st + xoff[1] + i * 2048);
    }
    if constexpr (have_prev) {
      load_sz(kt - 1);
      load_xs(kt - 1);
    }
    // weight operand of the tile: the lane's nibble plane of its 32 packed bytes
    uint32_t op[8];
    {
      const uint32_t raw[8] = {w[0].x, w[0].y, w[0].z, w[0].w, w[1].x, w[1].y, w[1].z, w[1].w};
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const uint32_t s = raw[d] >> nib_shift;
        if (a.ablate & 8) {
          op[d] = raw[d];
        } else if constexpr (AK == A_I8) {
          op[d] = s & nibmask;
        } else {
          const uint32_t q7 = s & 0x07070707u;
          const uint32_t lo = __builtin_amdgcn_perm(t_lo1, t_lo0, q7), hi = __builtin_amdgcn_perm(t_hi1, t_hi0, q7);
          const uint32_t sel = ((s >> 1) & 0x04040404u) | 0x03020100u;  // byte i of the result: byte i of lo, or of hi when bit 3 of the nibble is set
          op[d] = __builtin_amdgcn_perm(hi, lo, sel);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if constexpr (AK == A_I8) {
        // step order (i0,h0) (i1,h0) (i0,h1) (i1,h1) per pair of token fragments: the two dependent K = 64 products of a fragment are one independent
        // matrix instruction apart
        const int i = 2 * (s >> 2) + (s & 1), h = (s >> 1) & 1;
        const i32x4 wa = h == 0 ? i32x4{(int)op[0], (int)op[1], (int)op[2], (int)op[3]} : i32x4{(int)op[4], (int)op[5], (int)op[6], (int)op[7]};
        const uint4& xv = h == 0 ? xl[i] : xh[i];
        if (!(a.ablate & 2)) cg[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, __builtin_bit_cast(i32x4, xv), h == 0 ? i32x4{0, 0, 0, 0} : cg[i], 0, 0, 0);
      } else {
        const int i = s;
        const i32x8 wa = i32x8{(int)op[0], (int)op[1], (int)op[2], (int)op[3], (int)op[4], (int)op[5], (int)op[6], (int)op[7]};
        const i32x8 xa = i32x8{(int)xl[i].x, (int)xl[i].y, (int)xl[i].z, (int)xl[i].w, (int)xh[i].x, (int)xh[i].y, (int)xh[i].z, (int)xh[i].w};
        if (!(a.ablate & 2)) cg[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xa, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);  // e4m3 x e4m3, scales 2^0
      }
      if constexpr (have_prev) {
        if (!(a.ablate & 1)) {
#pragma unroll
          for (int q = s * FPS; q < (s + 1) * FPS; ++q) fold_slice(pg, q);
        }
      }
      // the fragment two ahead, once per token fragment (int8: fragments 2p+2, 2p+3 behind steps 0 and 2 of pair p - their registers are free: the
      // fragments of pair p+1 are not in use yet)
      const int inext = AK == A_I8 ? 2 * (s >> 2) + 2 + ((s >> 1) & 1) : s + 2;
      if ((AK != A_I8 || (s & 1) == 0) && inext < MI) {
        xl[inext] = *reinterpret_cast<const uint4*>(st + xoff[0] + inext * 2048);
        xh[inext] = *reinterpret_cast<const uint4*>(st + xoff[1] + inext * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // group sums of a for this wave's token fragment: the same matrix instruction against an all-ones weight operand (exact)
    if (my_xs >= 0) {
      const uint4 al = *reinterpret_cast<const uint4*>(st + xoff[0] + my_xs * 2048), ah = *reinterpret_cast<const uint4*>(st + xoff[1] + my_xs * 2048);
      float sum;
      if constexpr (AK == A_I8) {
        const i32x4 ones = i32x4{0x01010101, 0x01010101, 0x01010101, 0x01010101};
        i32x4 cx = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, __builtin_bit_cast(i32x4, al), i32x4{0, 0, 0, 0}, 0, 0, 0);
        cx = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, __builtin_bit_cast(i32x4, ah), cx, 0, 0, 0);
        sum = (float)cx[0];
      } else {
        const i32x8 ones = i32x8{0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};  // e4m3 1.0
        const i32x8 xa = i32x8{(int)al.x, (int)al.y, (int)al.z, (int)al.w, (int)ah.x, (int)ah.y, (int)ah.z, (int)ah.w};
        const f32x4 cx = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, xa, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        sum = cx[0];
      }
      // the matrix instruction must run with all 64 lanes: without this barrier hipcc sinks it into the lane < 16 branch below (EXEC = 0xFFFF), where the
      // K = 128 MX-format form returned wrong sums (r6 visit 3: profiles/r06_w4a8_fp8_exec_masked_mfma.md)
      asm volatile("" : "+v"(sum));
      if (lane < 16) xs_slot[(kt & 1) * BM + my_xs * 16 + lane] = sum;  // every row of the product holds the sum: row 0 leaves it for the fold one tile later
    }
  };
  auto final_fold = [&](const GV (&pg)[MI]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_sz(nk - 1);
    load_xs(nk - 1);
#pragma unroll
    for (int q = 0; q < 4 * MI; ++q) fold_slice(pg, q);
  };
  using yes = std::integral_constant<bool, true>;
  using st0 = std::integral_constant<int, 0>;
  using st1 = std::integral_constant<int, 1>;
  tile(0, accgA, accgB, std::integral_constant<bool, false>{}, st0{});
  int kt = 1;
  for (; kt + 2 <= nk; kt += 2) {
    tile(kt, accgB, accgA, yes{}, st1{});
    tile(kt + 1, accgA, accgB, yes{}, st0{});
  }
  if (kt < nk) {
    tile(kt, accgB, accgA, yes{}, st1{});
    final_fold(accgB);
  } else {
    final_fold(accgA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-requested last tile: nothing may land in LDS after the kernel moved on

  // ---- split-K: the protocol of qbits_mfma_fused.hip / qbits_skinny.hip (write-through partial tiles, arrival counter, last arriver adds in split order) ----
  if (S > 1) {
    const int tile_id = blockIdx.y * gridDim.x + blockIdx.x;
    float* mine = a.partials + ((size_t)(tile_id * S + sp) * MI * (WAVES * 64) + tid) * 4;
#pragma unroll
    for (int i = 0; i < MI; ++i)
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(mine + i * (WAVES * 64 * 4)), "v"(acc[i]) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) *flag = __hip_atomic_fetch_add(a.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (*flag != S - 1) return;
    if (tid == 0) __hip_atomic_store(a.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int QB = BM == 64 ? 4 : 2;
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
        for (int e = 0; e < MI; ++e) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[j][e])::"memory");
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

  // ---- epilogue: x activation scale, (+ bias), 4 consecutive features of one token per fragment: 8-byte stores ----
  T* yg = reinterpret_cast<T*>(a.y);
  const bool has_bias = a.bias != nullptr;
  const int pl = p0 + wave * 8 + 4 * (fg & 1);
  const int n0 = pl + (fg >> 1) * P;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (has_bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = pl + r < P ? E::to_f32(reinterpret_cast<const T*>(a.bias)[n0 + r]) : 0.f;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + i * 16 + fi;
    if (m < M) {
      T out[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][r] * sx;
        asm volatile("" : "+v"(v));  // the product is rounded to fp32 before anything else happens to it
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

inline int lds_bytes(int groups, int bm) { return STAGES * (bm * BK + W_BYTES) + 2 * bm * 4 + groups * 2 * (2 * PR) * 2; }
inline int tiles_of(int64_t M, int64_t N, int bm) { return (int)(((N / 2 + PR - 1) / PR) * ((M + bm - 1) / bm)); }

// Token tile and K split from the time model of qbits_mfma_fused.hip with this kernel's tile times (r6 sweep, profiles/r06_w4a8_*): 128-token tiles
// once they alone give every CU a workgroup, 64-token tiles (two workgroups per CU) below; K split for few tiles.  QUANTO_HIP_A8_BM / _SPLIT force.
struct Plan {
  int bm, S;
  float us;
};
inline float model_us(int tiles, int nk, int bm, int S) {
  const int wgs = tiles * S, rounds = (wgs + 255) / 256;
  const float tail = S > 1 ? 3.5f + 0.5f * (float)wgs * (float)(bm * 512) * 1e-6f : 0.f;
  return 5.8f + (float)rounds * (float)nk * (bm == 64 ? 0.45f : 0.75f) + tail;
}
inline Plan make_plan(int64_t M, int64_t N, int G) {
  const int fbm = env_int("QUANTO_HIP_A8_BM", 0), fs = env_int("QUANTO_HIP_A8_SPLIT", 0);  // experiments / tests
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
      if (best.bm == 0 || us < best.us * 0.97f) best = Plan{bm, S, us};
    }
  }
  return best;
}

template <int DT, int AK, bool INT_SHIFT, int BM>
static int launch_bm(const Args& a, hipStream_t stream) {
  const int lds = lds_bytes(a.G / a.S, BM);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbits_a8_fused_kernel<DT, AK, INT_SHIFT, BM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const dim3 grid((unsigned)((a.N / 2 + PR - 1) / PR), (unsigned)((a.M + BM - 1) / BM), (unsigned)a.S);
  hipLaunchKernelGGL((qbits_a8_fused_kernel<DT, AK, INT_SHIFT, BM>), grid, dim3(WAVES * 64), lds, stream, a);
  return launch_status();
}
template <int DT, int AK, bool INT_SHIFT>
static int launch(const Args& a, int bm, hipStream_t stream) {
  return bm == 64 ? launch_bm<DT, AK, INT_SHIFT, 64>(a, stream) : launch_bm<DT, AK, INT_SHIFT, 128>(a, stream);
}
template <int DT, int AK>
static int launch_shift(const Args& a, int bm, bool int_shift, hipStream_t stream) {
  return int_shift ? launch<DT, AK, true>(a, bm, stream) : launch<DT, AK, false>(a, bm, stream);
}

}  // namespace a8

bool qbits_a8_supported(int64_t M, const PackedGeom& g, int a_dtype, int dtype) {
  if (!(g.bits == 4 && g.C == 128 && (g.N % 8 == 0) && (g.K % 128 == 0) && M >= 1 && (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) &&
        (a_dtype == QUANTO_HIP_I8 || a_dtype == QUANTO_HIP_F8_E4M3FN) && g.N < (1 << 30) && g.K < (1 << 30) && M * g.K < (1ll << 32) &&
        g.N * g.K < (1ll << 33)))
    return false;
  return a8::make_plan(M, g.N, (int)g.G).bm != 0;
}

size_t qbits_a8_workspace(int64_t M, const PackedGeom& g) {
  const a8::Plan p = a8::make_plan(M, g.N, (int)g.G);
  if (p.bm == 0 || p.S == 1) return 0;
  return QUANTO_HIP_WS_COUNTER_BYTES + (size_t)a8::tiles_of(M, g.N, p.bm) * p.S * (a8::WAVES * 64) * ((p.bm / 16) * 16);
}

int qbits_mm_a8(const void* act, const void* act_scale, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t M,
                const PackedGeom& g, int a_dtype, int dtype, bool int_shift, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!qbits_a8_supported(M, g, a_dtype, dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(act) | reinterpret_cast<uintptr_t>(packed)) % 16) return QUANTO_HIP_EALIGN;
  a8::Plan p = a8::make_plan(M, g.N, (int)g.G);
  if (p.S > 1 && (!workspace || workspace_bytes < qbits_a8_workspace(M, g) || reinterpret_cast<uintptr_t>(workspace) % 16)) {
    p.S = 1;  // no scratch: unsplit, with whichever token tile lets the whole scale table fit
    if (a8::lds_bytes((int)g.G, p.bm) > 160 * 1024) p.bm = 64;
    if (a8::lds_bytes((int)g.G, p.bm) > 160 * 1024) return QUANTO_HIP_EINVAL;
  }
  const a8::Args a{reinterpret_cast<const uint8_t*>(act), act_scale, packed, scale, shift, bias, y, (int)M, (int)g.N, (int)g.K, (int)g.G, p.S,
                   reinterpret_cast<int*>(workspace),
                   p.S > 1 ? reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(workspace) + QUANTO_HIP_WS_COUNTER_BYTES) : nullptr,
                   env_int("QUANTO_HIP_A8_ABLATE", 0)};
  if (dtype == QUANTO_HIP_BF16)
    return a_dtype == QUANTO_HIP_I8 ? a8::launch_shift<QUANTO_HIP_BF16, a8::A_I8>(a, p.bm, int_shift, stream)
                                    : a8::launch_shift<QUANTO_HIP_BF16, a8::A_F8E4M3>(a, p.bm, int_shift, stream);
  return a_dtype == QUANTO_HIP_I8 ? a8::launch_shift<QUANTO_HIP_F16, a8::A_I8>(a, p.bm, int_shift, stream)
                                  : a8::launch_shift<QUANTO_HIP_F16, a8::A_F8E4M3>(a, p.bm, int_shift, stream);
}

}  // namespace qh
