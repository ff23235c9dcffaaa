// qbits_mm for small batches (4 < M <= 256 in passes of 64, e.g. batched decode): weight-streaming MFMA kernel over the generic
// PackedTensor layout (int4, group size 128, N even).
//
// HBM-bound like the GEMV, but the products run on the matrix cores so the cost per weight byte does not grow with M:
//   * a block of 4 waves owns 64 output features = 32 packed rows (byte (p,k) = W[p,k] | W[p+N/2,k] << 4) and streams
//     them over the whole K in tiles of 128 byte-columns (= one quantization group) through a 4..8-stage LDS-DMA pipeline
//     (`global_load_lds_dwordx4`, counted vmcnt, one s_barrier per tile); the loop contains no other memory instruction,
//     so nothing ever drains the DMA queue;
//   * wave w owns packed rows 8w..8w+7.  Its MFMA A operand is 16 FEATURES = those 8 rows x both nibble planes: lane
//     (i = lane & 15, g = lane >> 4) reads 16 bytes of row i%8 and keeps the low nibbles (i < 8) or the high nibbles
//     (i >= 8) - lanes i and i+8 read the same LDS address (broadcast).  128+q is built exactly as a bf16/fp16 number
//     (shift + mask per raw dword, one v_perm per pair of weights), 1 VALU op per weight;
//   * the activation tile (16*TF tokens x 128 k) is shared by the 4 waves; D[feature][token] accumulates one group in
//     fp32, then   acc += s[f,g]*acc_g - (z[f,g] + 128 s[f,g]) * XS[token,g],  with XS[token,g] = sum_k x from one extra
//     MFMA per k-step against an all-ones operand (no pre-kernel, no workspace).  The scales/shifts of the block's 64
//     features (16-bit) are parked in LDS before the loop (8 KB; 28 KB for K = 14336);
//   * a lane ends with 4 consecutive output features of one token: 8-byte stores.
// LDS images are linear per DMA instruction; bank-conflict swizzles are applied to the DMA source address and undone on
// the fragment reads: weights (128-byte rows, 8 chunks) chunk ^ (row & 7); activations (256-byte rows, 16 chunks)
// chunk ^ (row & 15).
#include <cstdlib>

#include "qh_common.h"

namespace qh {
namespace skinny {

constexpr int BK = 128;  // byte-columns (= k) per tile = one group

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

// s_waitcnt vmcnt(n * PER) for n = 0 .. MAXN/PER: the immediate must be a literal, hence the ladder
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
  const void* x;        // [M, K]
  const uint8_t* w;     // packed [N/2, K]
  const void* scale;    // [N*G]
  const void* shift;    // [N*G]
  const void* bias;     // [N] or null
  void* y;              // [M, N]
  int M, N, K, G;
  // split-K (S > 1): block b handles feature block b / S and K-range b % S; partial fp32 sums go to `partials`, the last block
  // of a feature block to arrive (atomic counter, zero on entry, reset to zero on exit) adds them up in split order and stores
  int S;
  int* counters;    // [N / (16*WAVES)]
  float* partials;  // [blocks][TF][WAVES*64 lanes] float4 (fragment-major: every store / load instruction covers whole lines)
  // groups per 128-k tile: 1 (group size 128), 2 (64), 4 (32); per_channel: ONE scale / shift per feature (G = 1), the tables repeat it
  int gpt, per_channel;
  int planes;       // values per packed byte: 2 (int4), 4 (int2)
  int tk;           // k per tile: 128, or 96 (group size 96)
  int nt;           // non-temporal weight DMA (single-pass calls: M <= 64)
  // QUANTO_HIP_SKINNY_ABLATE (timing experiments, WRONG results): 1 no split-K reduction, 2 no MFMA/LDS-read work,
  // 4 activation DMA re-reads tile 0 (L2 hits), 8 weight DMA re-reads tile 0 (no HBM traffic), 16 no scale/shift table;
  // r6, the skeleton itself: 32 no activation DMA instruction at all, 64 no weight DMA instruction at all, 128 return at entry (launch floor)
  int ablate;
  // QUANTO_HIP_SKINNY_TIMELINE=<device address of 32 x uint64 per block>: thread 0 of every block stamps s_memtime at the
  // phase boundaries (scripts/skinny_timeline.py); null in production
  unsigned long long* tl;
};

// Several Linears that share their input (q/k/v, gate/up) in ONE launch (MULTI): the feature blocks of all segments form one
// grid, a block looks its segment up and takes weight / scale / shift / bias / output pointers and N from it.  x, M, K, the
// split and the workspace are common; counters are indexed by the global feature block.
constexpr int MAX_SEGS = QUANTO_HIP_MAX_MULTI;
struct Segs {
  const uint8_t* w[MAX_SEGS];
  const void* scale[MAX_SEGS];
  const void* shift[MAX_SEGS];
  const void* bias[MAX_SEGS];
  void* y[MAX_SEGS];
  int N[MAX_SEGS];
  int first_fb[MAX_SEGS];  // first feature block of each segment (INT_MAX for unused slots)
};

// WAVES per block: 4 (64 features per block), 2 or 1 (16 features) - so that any N that is a multiple of 16 is served.
// SETS = 2 (64-feature blocks only): EIGHT waves, two sets of four with the same feature mapping.  The two tiles a barrier
// interval consumes go to the two sets - set 0 takes the even tile, set 1 the odd one - instead of one after the other through
// the single wave a SIMD has with SETS = 1: SQ counters of the four-wave kernel at (32,4096,4096) show waves WAITING (s_waitcnt,
// s_barrier) 59 % of their cycles, with nobody on the SIMD to use them.  The sets' sums are added through LDS at the end
// (set 0 + set 1: fixed order) and set 0 runs the split-K tail.  DMA: per pair of tiles one weight piece + TF activation pieces
// per wave.
// GPT: quantization groups per 128-k tile - 1 (group size 128 and per-channel scales), 2 (64), 4 (32); the small group sizes are
// instantiated for 64-feature blocks and the 4-stage ring only.
// PLANES: values per packed byte - 2 (int4) or 4 (int2, r4).  A wave's 16 features are then 4 packed rows x 4 planes (lane i of 16: row
// i & 3, plane i >> 2, two bits at 2 * plane) instead of 8 rows x 2 planes; its weight piece of a tile is 512 bytes (the lower 32 lanes of the
// DMA instruction).  Instantiated for 64-feature blocks, the 4-stage ring and group size 128.
// TK: k per tile - 128, or 96 (r4: group size 96, what nn/qmodule.py:121-129 picks for in_features = 96 (2j + 1), e.g. 1152, 2880, 4800).  A
// tile is then ONE group of 96 = three k-steps; the LDS image keeps its 128-byte (weights) / 256-byte (activations) row pitch and its
// swizzles, the DMA lanes whose chunk lies beyond k = 96 stay idle (6 of 8 / 12 of 16 chunks per row) and nothing reads those holes.
// 64-feature blocks, the 4-stage ring, one Linear.
template <int DT, int TF, int STAGES, bool INT_SHIFT, int WAVES, bool MULTI = false, int SETS = 1, int GPT = 1, int PLANES = 2, int TK = 128>
__global__ void __launch_bounds__(WAVES * SETS * 64) qbits_skinny_kernel(Args a, const Segs segs) {
  static_assert(SETS == 1 || WAVES == 4, "two wave sets: 64-feature blocks only");
  static_assert(PLANES == 2 || (PLANES == 4 && WAVES == 4 && !MULTI && GPT == 1), "int2: 64-feature blocks, one Linear, group size 128");
  static_assert(TK == 128 || (TK == 96 && WAVES == 4 && !MULTI && GPT == 1 && PLANES == 2), "group size 96: 64-feature blocks, one Linear, int4");
  constexpr bool K32MAP = GPT == 4 || TK == 96;  // k-step t covers k = 32 t .. 32 t + 31 (a k-step must stay inside one group / the 96 valid k)
  constexpr int RPW = 16 / PLANES;    // packed rows per wave
  constexpr int QSHIFT = DT == QUANTO_HIP_BF16 ? 5 : 6;  // int2 only: position of the two bits inside the mantissa byte
  using E = Elem<DT>;
  using T = typename E::T;
  using V8 = typename Mma<DT>::V8;
  constexpr int ROWS = RPW * WAVES;   // packed rows per block (PLANES * ROWS output features)
  constexpr int W_BYTES = ROWS * BK;  // 1 KiB per wave (int2: 512 B)
  constexpr int XP = TF * 4 / WAVES;  // 1 KiB activation DMA pieces per wave and tile
  constexpr int X_BYTES = TF * 16 * BK * 2;
  constexpr int STAGE_BYTES = W_BYTES + X_BYTES;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // layout: [STAGES x (W tile | x tile)] [sz: G x 2 x 64 elements of T]
  T* sz = reinterpret_cast<T*>(smem + STAGES * STAGE_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0 .. WAVES * SETS - 1
  const int set = SETS == 1 ? 0 : wave_id / WAVES, wave = SETS == 1 ? wave_id : wave_id % WAVES;  // wave = feature group of 16
  auto probe = [&](int slot) {
    if (a.tl && tid == 0) a.tl[blockIdx.x * 32 + slot] = __builtin_readcyclecounter();
  };
  probe(0);
  if (a.tl && tid == 0) a.tl[blockIdx.x * 32 + 30] = wall_clock64();
  if (a.ablate & 128) return;
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
    a.shift = segs.shift[seg];
    a.bias = segs.bias[seg];
    a.y = segs.y[seg];
    a.N = segs.N[seg];
  }
  const int M = a.M, N = a.N, K = a.K;
  const int P = N / PLANES;
  const int p0 = fb * ROWS;
  const int nk = K / TK / S;   // tiles (= groups) of this block's K-range
  const int kt0 = sp * nk;     // first global tile / group index
  const int G = nk * GPT;      // groups held in the LDS tables

  // ---- per-lane DMA sources ---------------------------------------------------------------------------
  // weights: this wave's 8 rows x 128 B = 1 KiB per tile; lane -> row lane>>3, position lane&7 holds chunk pos ^ (row & 7)
  const uint8_t* wsrc;
  bool w_valid;  // TK = 96: the lane's chunk lies inside the tile
  {
    // the swizzle is a function of the row INSIDE THE BLOCK (what the fragment reads undo); int2: lanes 32..63 repeat rows 0..3 and stay idle (w_lane)
    const int r = (lane >> 3) & (RPW - 1), c = (lane & 7) ^ ((wave * RPW + r) & 7);
    wsrc = a.w + (size_t)(p0 + wave * RPW + r) * K + c * 16 + (size_t)kt0 * TK;
    w_valid = c * 16 < TK;
  }
  // activations: XP KiB-instructions per wave; instruction u covers tile rows 4*(wave*XP+u) .. +3
  const uint8_t* xsrc[XP];
  bool x_valid[XP];
#pragma unroll
  for (int u = 0; u < XP; ++u) {
    const int row = 4 * (wave * XP + u) + (lane >> 4);
    const int c = (lane & 15) ^ (row & 15);
    const int m = row < M ? row : M - 1;
    xsrc[u] = reinterpret_cast<const uint8_t*>(reinterpret_cast<const T*>(a.x) + (size_t)m * K + c * 8 + (size_t)kt0 * TK);
    x_valid[u] = c * 8 < TK;
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  const bool w_lane = (PLANES == 2 || lane < 32) && (TK == 128 || w_valid);  // lanes that carry a weight DMA piece (RPW rows x 128 B = RPW * 8 lanes)
  auto issue = [&](int kt, int stage) {
    const uint32_t st = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE_BYTES);
    // ablations keep the instruction count per tile (vmcnt arithmetic) and re-read tile 0 instead: L2 hits
    const int ktw = (a.ablate & 8) ? 0 : kt, ktx = (a.ablate & 4) ? 0 : kt;
    if (w_lane && !(a.ablate & 64)) {
      if (a.nt)
        glds16_nt(wsrc + (size_t)ktw * TK, st + wave * (RPW * BK));
      else
        glds16(wsrc + (size_t)ktw * TK, st + wave * (RPW * BK));
    }
#pragma unroll
    for (int u = 0; u < XP; ++u)
      if ((TK == 128 || x_valid[u]) && !(a.ablate & 32)) glds16(xsrc[u] + (size_t)ktx * (TK * 2), st + W_BYTES + (wave * XP + u) * 1024);
  };
  // SETS = 2: the pieces of a PAIR of tiles (even tile -> stage sa, odd tile -> stage sb) dealt over the eight waves: wave (set,
  // f) carries weight piece f of the tile of its set and the activation pieces q = wave_id * TF + j of the 8 TF of the pair
  const uint8_t* xsrc2[SETS == 2 ? TF : 1];
  uint32_t xdst2[SETS == 2 ? TF : 1];
  bool x_valid2[SETS == 2 ? TF : 1];
  if constexpr (SETS == 2) {
#pragma unroll
    for (int j = 0; j < TF; ++j) {
      const int q = wave_id * TF + j, odd = q / (4 * TF), piece = q % (4 * TF);
      const int row = 4 * piece + (lane >> 4);
      const int c = (lane & 15) ^ (row & 15);
      const int m = row < M ? row : M - 1;
      xsrc2[j] = reinterpret_cast<const uint8_t*>(reinterpret_cast<const T*>(a.x) + (size_t)m * K + c * 8 + (size_t)(kt0 + odd) * TK);
      xdst2[j] = ((uint32_t)odd << 31) | (uint32_t)(W_BYTES + piece * 1024);  // bit 31: the odd tile's stage
      x_valid2[j] = c * 8 < TK;
    }
  }
  auto issue_pair = [&](int kt, int sa, int sb) {  // kt even; the odd tile exists when kt + 1 < nk
    const uint32_t sta = __builtin_amdgcn_readfirstlane(lds_base + sa * STAGE_BYTES), stb = __builtin_amdgcn_readfirstlane(lds_base + sb * STAGE_BYTES);
    const bool has_odd = kt + 1 < nk;
    const int ktw = (a.ablate & 8) ? 0 : kt + set, ktx = (a.ablate & 4) ? 0 : kt;  // ablations: re-read the first tile (pair) instead: L2 hits
    if ((set == 0 || has_odd) && w_lane && !(a.ablate & 64)) {
      const uint32_t st = set == 0 ? sta : stb;
      if (a.nt)
        glds16_nt(wsrc + (size_t)ktw * TK, st + wave * (RPW * BK));
      else
        glds16(wsrc + (size_t)ktw * TK, st + wave * (RPW * BK));
    }
#pragma unroll
    for (int j = 0; j < (SETS == 2 ? TF : 0); ++j) {
      const bool odd = xdst2[j] >> 31;
      if ((!odd || has_odd) && (TK == 128 || x_valid2[j]) && !(a.ablate & 32))
        glds16(xsrc2[j] + (size_t)ktx * (TK * 2), (odd ? stb : sta) + (xdst2[j] & 0x7FFFFFFFu));
    }
  };
  if constexpr (SETS == 2) {
#pragma unroll
    for (int t = 0; t < STAGES - 2; t += 2)
      if (t < nk) issue_pair(t, t, t + 1);
  } else {
#pragma unroll
    for (int t = 0; t < STAGES - 2; ++t)
      if (t < nk) issue(t, t);
  }

  // ---- park scale / (shift + OFFSET*scale) of the block's 64 features and the XS rows in LDS ---------------------
  // sz[g][0][f] = scale, sz[g][1][f] = shift (zero-points converted to T: small integers are exact),
  // f = plane*ROWS + local packed row; kept in the 16-bit storage type so that K = 14336 (112 groups) fits
  constexpr int NF = PLANES * ROWS;  // features per block
  // Row pitch of the table: NF + 4 entries.  With a pitch of NF (a multiple of 256 bytes for 64 features) the fill below - lanes
  // run over the groups of one feature - put all 64 lanes of a ds_write_b16 on ONE bank (SQ_LDS_BANK_CONFLICT = 3970 cycles per
  // block on the gate+up launch, 0.9 us of the (32,4096,4096) call); 8 bytes of padding spread them and keep the 8-byte reads aligned
  constexpr int NFP = NF + 4;
  for (int e = tid; e < ((a.ablate & 16) ? 0 : NF * G); e += WAVES * SETS * 64) {
    const int f = e / G, g = e - f * G;
    const size_t idx = (size_t)(p0 + (f % ROWS) + (f / ROWS) * P) * a.G + (a.per_channel ? 0 : kt0 * GPT + g);
    sz[(g * 2 + 0) * NFP + f] = reinterpret_cast<const T*>(a.scale)[idx];
    if constexpr (INT_SHIFT)
      sz[(g * 2 + 1) * NFP + f] = E::from_f32((float)(int8_t) reinterpret_cast<const uint8_t*>(a.shift)[idx]);
    else
      sz[(g * 2 + 1) * NFP + f] = reinterpret_cast<const T*>(a.shift)[idx];
  }

  // ---- fragment read offsets ----------------------------------------------------------------------------------------
  const int fi = lane & 15, fg = lane >> 4;
  const int wrow = wave * RPW + (fi & (RPW - 1));
  int woff[2];  // 16-byte chunks fg and 4+fg of the lane's row
#pragma unroll
  for (int h = 0; h < 2; ++h) woff[h] = wrow * 128 + (((4 * h + fg) ^ (wrow & 7)) << 4);
  // group size 32 (GPT = 4): a k-step must stay inside ONE group, so k-step t is k = 32 t .. 32 t + 31 and lane group fg takes its bytes
  // 8 fg .. 8 fg + 7: half fg & 1 of chunk 2 t + (fg >> 1) - four 8-byte reads instead of two 16-byte ones
  int woff4[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) woff4[t] = wrow * 128 + (((2 * t + (fg >> 1)) ^ (wrow & 7)) << 4) + 8 * (fg & 1);
  const uint32_t nib_shift = PLANES == 2 ? (fi >> 3) * 4 : (fi >> 2) * 2;  // the lane's plane: nibble 0 / 1, or bit pair 0..3
  int xoff[TF][4];
#pragma unroll
  for (int tf = 0; tf < TF; ++tf) {
    const int row = tf * 16 + fi;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      xoff[tf][t] = W_BYTES + row * 256 + (((K32MAP ? 4 * t + fg : 8 * (t >> 1) + 2 * fg + (t & 1)) ^ (row & 15)) << 4);
  }
  // this lane's 4 consecutive features inside the block: int4 plane (fg>>1), local packed rows wave*8 + 4*(fg&1) + r; int2 plane fg, rows wave*4 + r
  const int floc = PLANES == 2 ? (fg >> 1) * ROWS + wave * 8 + 4 * (fg & 1) : fg * ROWS + wave * 4;

  f32x4 acc[TF];
#pragma unroll
  for (int tf = 0; tf < TF; ++tf) acc[tf] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint32_t kmask = PLANES == 2 ? 0x0F0F0F0Fu : 0x03030303u, kmagic = Mma<DT>::MAGIC;
  asm volatile("" : "+s"(kmask));
  asm volatile("" : "+v"(kmagic));

  // all-ones operand: one extra MFMA per k-step returns sum_k x[token, k] (the XS term of the group fold) in fp32,
  // without any VALU work or a pre-kernel; the matrix pipe is idle most of the time in this HBM-bound kernel.
  const V8 ones = __builtin_bit_cast(V8, make_uint4(ONE2<DT>(), ONE2<DT>(), ONE2<DT>(), ONE2<DT>()));

  // One tile = 128 k = GPT groups: read the wave's weight bytes, 4 k-steps x (TF + TF) MFMAs, one fold into acc per group (after
  // 4 / GPT k-steps: group sizes 128, 64, 32)
  auto compute_tile = [&](const uint8_t* st, int kt) {
    constexpr int KS = TK == 96 ? 3 : 4 / GPT;
    uint4 wr[2];
    if constexpr (K32MAP) {
      const uint2 q0 = *reinterpret_cast<const uint2*>(st + woff4[0]), q1 = *reinterpret_cast<const uint2*>(st + woff4[1]);
      const uint2 q2 = *reinterpret_cast<const uint2*>(st + woff4[2]);
      const uint2 q3 = TK == 96 ? q2 : *reinterpret_cast<const uint2*>(st + woff4[3]);  // group size 96: chunks 6, 7 of a row were never written
      wr[0] = make_uint4(q0.x, q0.y, q1.x, q1.y);  // same register picture as below: k-step t = dwords 2 (t & 1), 2 (t & 1) + 1 of wr[t >> 1]
      wr[1] = make_uint4(q2.x, q2.y, q3.x, q3.y);
    } else {
      wr[0] = *reinterpret_cast<const uint4*>(st + woff[0]);
      wr[1] = *reinterpret_cast<const uint4*>(st + woff[1]);
    }
    // every activation fragment of the tile up front (r5): TF x 4 (3 for tiles of 96) independent ds_read_b128 in flight together.  hipcc,
    // left to itself, reused ONE register quad for all of them - read, wait lgkmcnt(0), two MFMAs, eight times per tile: eight exposed LDS
    // latencies made a tile ~0.6 us of a wave's time and the K loop of a (32,4096,4096) call 3.4 of its 9.8 us (in-kernel timeline,
    // profiles/r05_batched_decode_timeline.txt)
    V8 xb[TF][GPT * KS];
#pragma unroll
    for (int t = 0; t < GPT * KS; ++t)
#pragma unroll
      for (int tf = 0; tf < TF; ++tf) xb[tf][t] = *reinterpret_cast<const V8*>(st + xoff[tf][t]);
    __builtin_amdgcn_sched_barrier(0);  // (hipcc sinks the reads back next to their MFMAs otherwise: it schedules for a 64-register kernel)
#pragma unroll
    for (int gq = 0; gq < GPT; ++gq) {
      f32x4 accg[TF], accx[TF];
#pragma unroll
      for (int tf = 0; tf < TF; ++tf) {
        accg[tf] = f32x4{0.f, 0.f, 0.f, 0.f};
        accx[tf] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int t = gq * KS; t < (gq + 1) * KS; ++t) {
        // k-step t uses bytes 8*(t&1) .. +7 of chunk (t>>1): two dwords -> four operand dwords (natural k order)
        const uint32_t d0 = (t & 1) ? wr[t >> 1].z : wr[t >> 1].x, d1 = (t & 1) ? wr[t >> 1].w : wr[t >> 1].y;
        // 8 VALU per 8 weights (r3, as qbits_mfma_fused.hip): the lane's nibble plane shifted down and masked once per raw dword, then ONE
        // v_perm per pair of weights interleaves their bytes with the exponent byte of 128 (bf16 0x43) / 1024 (fp16 0x64)
        uint32_t s0 = (d0 >> nib_shift) & kmask, s1 = (d1 >> nib_shift) & kmask;
        if constexpr (PLANES == 4) {
          // int2: the two useful bits go to the TOP of the mantissa byte - operand = OFFSET + QS q with QS = 32 (bf16: 128 + 32 q < 256) or
          // 64 (fp16: 1024 + 64 q < 2048).  With OFFSET + q they ride on a constant 85 / 680 times their own size and the fp32 cancellation
          // error of the fold reaches an fp16 ulp; the fold divides by QS (exact)
          s0 <<= QSHIFT;
          s1 <<= QSHIFT;
        }
        uint32_t op[4];
        op[0] = __builtin_amdgcn_perm(kmagic, s0, 0x07010500u);  // bytes 0,1 -> (q0, exp, q1, exp)
        op[1] = __builtin_amdgcn_perm(kmagic, s0, 0x07030502u);  // bytes 2,3
        op[2] = __builtin_amdgcn_perm(kmagic, s1, 0x07010500u);
        op[3] = __builtin_amdgcn_perm(kmagic, s1, 0x07030502u);
        const V8 wa = __builtin_bit_cast(V8, make_uint4(op[0], op[1], op[2], op[3]));
#pragma unroll
        for (int tf = 0; tf < TF; ++tf) {
          accg[tf] = Mma<DT>::run(wa, xb[tf][t], accg[tf]);
          accx[tf] = Mma<DT>::run(ones, xb[tf][t], accx[tf]);
        }
      }
      // fold the group: acc += s * acc_g - zz * XS
      const int g = kt * GPT + gq;
      T s4t[4], z4t[4];
      *reinterpret_cast<uint2*>(s4t) = *reinterpret_cast<const uint2*>(sz + (g * 2 + 0) * NFP + floc);
      *reinterpret_cast<uint2*>(z4t) = *reinterpret_cast<const uint2*>(sz + (g * 2 + 1) * NFP + floc);
      float s4[4], z4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        constexpr float QS = PLANES == 4 ? (float)(1 << QSHIFT) : 1.f;  // operand = OFFSET + QS * q
        s4[r] = E::to_f32(s4t[r]) * (1.f / QS);       // exact: a power of two
        const float z = E::to_f32(z4t[r]);
        z4[r] = INT_SHIFT ? s4[r] * (QS * z + Mma<DT>::OFFSET) : z + Mma<DT>::OFFSET * s4[r];
      }
#pragma unroll
      for (int tf = 0; tf < TF; ++tf) {
        const float xs = accx[tf][0];  // every row of the ones-product holds sum_k x[token, k] of this group
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tf][r] += s4[r] * accg[tf][r] - z4[r] * xs;
      }
    }
  };
  // Tiles are consumed in PAIRS per barrier: twice the work between synchronisations and two independent MFMA/LDS
  // chains for the scheduler to interleave.  Ring of STAGES (even) stages; tiles kt.. are in flight up to kt+STAGES-1.
  int cur = 0;
  probe(1);
  for (int kt = 0; kt < nk; kt += 2) {
    const bool pair = kt + 1 < nk;
    // tiles kt (and kt+1) have landed when at most the DMA of the tiles younger than them is outstanding
    const int last = pair ? kt + 1 : kt;
    const int younger = nk - 1 - last < STAGES - 4 ? nk - 1 - last : STAGES - 4;
    if constexpr (SETS == 2)  // whole pairs only (a trailing single tile counts as none: waits for more, never for less)
      wait_vmcnt<(STAGES - 4) / 2 * (1 + TF), 1 + TF>(younger / 2);
    else
      wait_vmcnt<(STAGES - 4) * (1 + XP), 1 + XP>(younger);
    __builtin_amdgcn_s_barrier();  // both tiles visible to all; everybody is done with the previous pair (and the tables are written)
    asm volatile("" ::: "memory");
    if (kt < 32) probe(3 + (kt >> 1));
    const int nxt = cur + 1 == STAGES ? 0 : cur + 1;
    {  // refill the two stages the previous pair occupied
      const int s0 = cur >= 2 ? cur - 2 : cur + STAGES - 2, s1 = s0 + 1 == STAGES ? 0 : s0 + 1;
      if constexpr (SETS == 2) {
        if (kt + STAGES - 2 < nk) issue_pair(kt + STAGES - 2, s0, s1);
      } else {
        if (kt + STAGES - 2 < nk) issue(kt + STAGES - 2, s0);
        if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, s1);
      }
    }
    if (!(a.ablate & 2)) {
      if constexpr (SETS == 2) {
        if (set == 0)
          compute_tile(smem + cur * STAGE_BYTES, kt);
        else if (pair)
          compute_tile(smem + nxt * STAGE_BYTES, kt + 1);
      } else {
        compute_tile(smem + cur * STAGE_BYTES, kt);
        if (pair) compute_tile(smem + nxt * STAGE_BYTES, kt + 1);
      }
    }
    cur = nxt + 1 == STAGES ? 0 : nxt + 1;
  }

  // ---- split-K: park the partial sums, elect the last block of this feature block, which reduces in split order -------------
  // The blocks of one feature block may run on different XCDs, whose L2s are not coherent with each other.  An agent-scope
  // fence would be correct but writes back / invalidates a whole L2 (measured: 23 -> 57 us); instead the few KiB of partials
  // travel with system-coherent (sc0 sc1) 16-byte stores and loads and the only
  // ordering needed is "my stores are acknowledged (vmcnt(0)) before my workgroup's arrival is counted".
  if constexpr (SETS == 2) {  // set 0 + set 1 through LDS (the ring is free: every DMA was consumed)
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    if (set == 1) {
#pragma unroll
      for (int tf = 0; tf < TF; ++tf) red[tf * 256 + (tid - 256)] = acc[tf];
    }
    __syncthreads();
    if (set == 1) return;
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) {
      const f32x4 o = red[tf * 256 + tid];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[tf][r] += o[r];
    }
    __syncthreads();  // the tail re-uses smem[0] as its flag
  }
  probe(20);
  if (S > 1 && (a.ablate & 1)) {
    if (sp != 0) return;
  } else if (S > 1) {
    float* mine = a.partials + ((size_t)blockIdx.x * TF * (WAVES * 64) + tid) * 4;
#pragma unroll
    for (int tf = 0; tf < TF; ++tf)  // s_nop: gfx9 hazard "VMEM store of > 64 bits, then VALU write of its data VGPRs" - hipcc cannot see into the asm
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(mine + tf * (WAVES * 64 * 4)), "v"(acc[tf]) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    probe(21);
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) *flag = __hip_atomic_fetch_add(a.counters + fbg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    probe(22);
    if (*flag != S - 1) return;
    if (tid == 0) __hip_atomic_store(a.counters + fbg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // leave the workspace as found
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) acc[tf] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fixed order: the result does not depend on which block arrived last.  The loads of up to four splits are in flight
    // together (one fabric round trip per four splits instead of one per split)
    for (int q0 = 0; q0 < S; q0 += 4) {
      f32x4 v[4][TF];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = q0 + j < S ? q0 + j : S - 1;
        const float* theirs = a.partials + ((size_t)(fbg * S + q) * TF * (WAVES * 64) + tid) * 4;
#pragma unroll
        for (int tf = 0; tf < TF; ++tf) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[j][tf]) : "v"(theirs + tf * (WAVES * 64 * 4)) : "memory");
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tf = 0; tf < TF; ++tf) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[j][tf])::"memory");  // ties the uses below to the wait
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (q0 + j < S) {
#pragma unroll
          for (int tf = 0; tf < TF; ++tf)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tf][r] += v[j][tf][r];
        }
    }
    probe(23);
  }

  // ---- epilogue ----------------------------------------------------------------------------------------------------------
  T* yg = reinterpret_cast<T*>(a.y);
  const int n0 = PLANES == 2 ? p0 + wave * 8 + 4 * (fg & 1) + (fg >> 1) * P : p0 + wave * 4 + fg * P;  // 4 consecutive output features n0..n0+3
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  const bool has_bias = a.bias != nullptr;
  if (has_bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = E::to_f32(reinterpret_cast<const T*>(a.bias)[n0 + r]);
  }
#pragma unroll
  for (int tf = 0; tf < TF; ++tf) {
    const int m = tf * 16 + fi;
    if (m < M) {
      T out[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[tf][r];
        if (has_bias) v = E::to_f32(E::from_f32(v)) + bv[r];
        out[r] = E::from_f32(v);
      }
      *reinterpret_cast<uint2*>(yg + (size_t)m * N + n0) = *reinterpret_cast<const uint2*>(out);
    }
  }
  probe(24);
  if (a.tl && tid == 0) a.tl[blockIdx.x * 32 + 31] = wall_clock64();
}

constexpr int lds_bytes(int tf, int stages, int G, int waves, int planes = 2) {
  return stages * (waves * (16 / planes) * BK + tf * 16 * BK * 2) + G * 2 * (16 * waves + 4) * 2;
}

// `segs` (with the total number of feature blocks) selects the multi-Linear launch; 64-feature blocks only, like the two-set form
template <int DT, int TF, int STAGES, bool INT_SHIFT, int WAVES, bool MULTI, int SETS, int GPT = 1, int PLANES = 2, int TK = 128>
static int launch_k(const Args& a, hipStream_t stream, const Segs& segs, int grid, int lds) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbits_skinny_kernel<DT, TF, STAGES, INT_SHIFT, WAVES, MULTI, SETS, GPT, PLANES, TK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((qbits_skinny_kernel<DT, TF, STAGES, INT_SHIFT, WAVES, MULTI, SETS, GPT, PLANES, TK>), dim3(grid), dim3(WAVES * SETS * 64), lds, stream, a, segs);
  return launch_status();
}

template <int DT, int TF, int STAGES, bool INT_SHIFT, int WAVES>
static int launch_s(const Args& a, hipStream_t stream, const Segs* segs = nullptr, int total_fb = 0) {
  const int lds = lds_bytes(TF, STAGES, a.K / BK / a.S, WAVES);
  if constexpr (WAVES == 4) {
    // eight waves per block from two token fragments on (us, four -> eight waves: (32,4096,4096) 10.74 -> 10.21, (64,4096,4096)
    // 17.6 -> 15.7, (32,14336,4096) 18.5 -> 17.7, gate+up M = 32 in one launch 29.5 -> 26.6; but one fragment, q/k/v M = 8:
    // 9.87 -> 10.52: too little work per tile to share; the 8-bit kernel of qbytes_skinny.hip gains nothing: int8 (32,4096,4096)
    // 10.95 -> 10.96, gate+up in one launch 26.5 -> 26.7 - four times the weight bytes per tile, no group fold)
    // r6: the choice is a compile-time one (the other wave-set count of each TF was reachable through QUANTO_HIP_SKINNY_SETS only and doubled
    // the 64-feature-block instantiations of this file)
    constexpr int SETS = TF >= 2 ? 2 : 1;
    if (segs) return launch_k<DT, TF, STAGES, INT_SHIFT, 4, true, SETS>(a, stream, *segs, total_fb * a.S, lds);
    if constexpr (SETS == 2) return launch_k<DT, TF, STAGES, INT_SHIFT, 4, false, 2>(a, stream, Segs{}, a.N / 64 * a.S, lds);
  }
  return launch_k<DT, TF, STAGES, INT_SHIFT, WAVES, false, 1>(a, stream, Segs{}, a.N / (16 * WAVES) * a.S, lds);
}

// Deepest DMA pipeline that fits: the kernel is latency-bound per block (bytes in flight = stages x tile bytes).  Narrow
// blocks budget for two (or three) blocks per CU.
template <int DT, int TF, bool INT_SHIFT, int WAVES>
static int launch(const Args& a, hipStream_t stream, const Segs* segs = nullptr, int total_fb = 0) {
  // LDS budget per block: 50 KiB = three blocks per CU.  A deeper ring with the CU to itself is slower: (32,4096,4096) 11.5 us
  // at 150 KiB, 11.0 at 76, 10.6 at 50; (32,4096,14336) split 2: 28.3 / 17.8 / 17.3 us
  const int budget = env_int("QUANTO_HIP_SKINNY_LDS_KB", 50) * 1024;
  constexpr int per_tile = 1 + TF * 4 / WAVES;  // DMA instructions per wave and tile; vmcnt counts at most 63 of them
  if constexpr ((8 - 4) * per_tile <= 60)
    if (lds_bytes(TF, 8, a.K / BK / a.S, WAVES) <= budget) return launch_s<DT, TF, 8, INT_SHIFT, WAVES>(a, stream, segs, total_fb);
  if constexpr ((6 - 4) * per_tile <= 60)
    if (lds_bytes(TF, 6, a.K / BK / a.S, WAVES) <= budget) return launch_s<DT, TF, 6, INT_SHIFT, WAVES>(a, stream, segs, total_fb);
  return launch_s<DT, TF, 4, INT_SHIFT, WAVES>(a, stream, segs, total_fb);
}

// waves per block: the widest that divides N.  Narrower blocks do not help small N (measured, N = 4096, M = 32: 4 waves
// 21.9 us, 2 waves 21.8 us, 1 wave 26.1 us): the kernel is bound by the instruction stream of the single wave each SIMD gets,
// not by the number of occupied CUs - what it lacks for N <= 4096 is K-parallelism.
inline int pick_waves(int N, int tf = 1) {
  const int forced = env_int("QUANTO_HIP_SKINNY_WAVES", 0);  // experiments
  // (r5 experiment, withdrawn in r6: forced == 8 -> 128 features per block - half the activation traffic per weight byte, twice the partial
  //  sums: (32,4096,4096) 11.0 vs 9.5 us, profiles/r05_batched_decode_128_feature_blocks_ab.jsonl; patch: scripts/archive/experiments_r6/)
  if (forced == 1 || forced == 2 || forced == 4) return (N % (16 * forced)) == 0 ? forced : 1;
  return N % 64 == 0 ? 4 : (N % 32 == 0 ? 2 : 1);
}

template <int DT, bool INT_SHIFT, int TF>
static int launch_waves(const Args& a, hipStream_t stream, const Segs* segs = nullptr, int total_fb = 0) {
  if (segs) return launch<DT, TF, INT_SHIFT, 4>(a, stream, segs, total_fb);
  const int w = pick_waves(a.N, TF);
  if (w == 4) return launch<DT, TF, INT_SHIFT, 4>(a, stream);
  if (w == 2) return launch<DT, TF, INT_SHIFT, 2>(a, stream);
  return launch<DT, TF, INT_SHIFT, 1>(a, stream);
}

// group sizes 64 / 32: 64-feature blocks, 4-stage ring, two wave sets from two token fragments on (as the group-size-128 form)
template <int DT, bool INT_SHIFT, int TF, int GPT>
static int launch_small_groups(const Args& a, hipStream_t stream) {
  const int lds = lds_bytes(TF, 4, a.K / BK / a.S * GPT, 4);
  return launch_k<DT, TF, 4, INT_SHIFT, 4, false, (TF >= 2 ? 2 : 1), GPT>(a, stream, Segs{}, a.N / 64 * a.S, lds);
}
template <int DT, bool INT_SHIFT, int GPT>
static int launch_small_groups_tf(const Args& a, hipStream_t stream) {
  if (a.M <= 16) return launch_small_groups<DT, INT_SHIFT, 1, GPT>(a, stream);
  if (a.M <= 32) return launch_small_groups<DT, INT_SHIFT, 2, GPT>(a, stream);
  return launch_small_groups<DT, INT_SHIFT, 4, GPT>(a, stream);
}

// qint2 (four planes per byte): 64-feature blocks, 4-stage ring, two wave sets from two token fragments on
template <int DT, bool INT_SHIFT, int TF>
static int launch_int2(const Args& a, hipStream_t stream) {
  const int lds = lds_bytes(TF, 4, a.K / BK / a.S, 4, 4);
  return launch_k<DT, TF, 4, INT_SHIFT, 4, false, (TF >= 2 ? 2 : 1), 1, 4>(a, stream, Segs{}, a.N / 64 * a.S, lds);
}

// group size 96 (r4): tiles of 96 k, 64-feature blocks, 4-stage ring, two wave sets from two token fragments on
template <int DT, bool INT_SHIFT, int TF>
static int launch_g96(const Args& a, hipStream_t stream) {
  const int lds = lds_bytes(TF, 4, a.K / 96 / a.S, 4);
  return launch_k<DT, TF, 4, INT_SHIFT, 4, false, (TF >= 2 ? 2 : 1), 1, 2, 96>(a, stream, Segs{}, a.N / 64 * a.S, lds);
}

template <int DT, bool INT_SHIFT>
static int launch_tf(const Args& a, hipStream_t stream, const Segs* segs = nullptr, int total_fb = 0) {
  if (a.tk == 96) {
    if (a.M <= 16) return launch_g96<DT, INT_SHIFT, 1>(a, stream);
    if (a.M <= 32) return launch_g96<DT, INT_SHIFT, 2>(a, stream);
    return launch_g96<DT, INT_SHIFT, 4>(a, stream);
  }
  if (a.planes == 4) {
    if (a.M <= 16) return launch_int2<DT, INT_SHIFT, 1>(a, stream);
    if (a.M <= 32) return launch_int2<DT, INT_SHIFT, 2>(a, stream);
    return launch_int2<DT, INT_SHIFT, 4>(a, stream);
  }
  if (a.gpt == 2) return launch_small_groups_tf<DT, INT_SHIFT, 2>(a, stream);
  if (a.gpt == 4) return launch_small_groups_tf<DT, INT_SHIFT, 4>(a, stream);
  if (a.M <= 16) return launch_waves<DT, INT_SHIFT, 1>(a, stream, segs, total_fb);
  if (a.M <= 32) return launch_waves<DT, INT_SHIFT, 2>(a, stream, segs, total_fb);
  return launch_waves<DT, INT_SHIFT, 4>(a, stream, segs, total_fb);
}

}  // namespace skinny

// Split factor.  r2 measurements (M = 32, K = 4096, us per launch, S = 1 / 2 / 4 / 8 with three blocks per CU): N = 4096
// - / 13.6 / 10.6 / 12.3, N = 14336 22.4 (one block per CU) / 17.3 / 21.4 / 28.1: the best split gives the chip 250-500 blocks of
// four waves (two to three co-resident blocks per CU hide each other's barrier and reduction stalls), never leaves a block
// fewer than 8 groups (its DMA ring would barely fill) and must divide the group count.
static int skinny_split(const PackedGeom& g, int64_t M) {
  const int forced = env_int("QUANTO_HIP_SKINNY_SPLIT", 0);  // experiments
  const int tf = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
  // group sizes 64 / 32 always launch 64-feature blocks (launch_small_groups), whatever the wave knob says
  const bool g96 = g.C == 96 && g.K != 96;
  const int blocks = (g.C == 64 || g.C == 32 || g96 || g.bits == 2) ? (int)(g.N / 64) : (int)(g.N / (16 * skinny::pick_waves((int)g.N, tf)));
  int s = 1;
  const int tiles = (int)(g.K / (g96 ? 96 : 128));  // 128-k tiles (= groups of 128), or groups of 96
  while (s < 8 && blocks * s * 2 <= 512 && tiles % (s * 2) == 0 && tiles / (s * 2) >= 8) s *= 2;
  // K = 96 j with j not a multiple of 4: the largest divisor of the tile count under the same bounds, not only powers of two, while the partial
  // sums are small (us, bf16, N = 4096, K = 4800 = 50 tiles, split 2 -> 5: M = 32 17.0 -> 13.5, but M = 64 20.3 -> 22.1, M = 128 39.3 -> 41.8)
  if (g96 && tf <= 2)
    for (int d = 8; d > s; --d)
      if (tiles % d == 0 && blocks * d <= 512 && tiles / d >= 8) {
        s = d;
        break;
      }
  if (forced > 0 && tiles % forced == 0) s = forced;
  if ((size_t)blocks * 4 > QUANTO_HIP_WS_COUNTER_BYTES) s = 1;  // one counter per feature block
  return s;
}
// The arrival counters of every split-K kernel of the library live in the same fixed-size region at the start of the
// workspace and the partial sums always start behind it: a buffer that served one problem can serve any other without a
// later call reading an earlier call's partial sums as counters (the partials are never reset, the counters always are).
static size_t skinny_counter_bytes(const PackedGeom&) { return QUANTO_HIP_WS_COUNTER_BYTES; }

bool qbits_skinny_supported(int64_t M, const PackedGeom& g, int dtype) {
  const int64_t Mp = M > 64 ? 64 : M;  // rows per pass
  const int tf = Mp <= 16 ? 1 : (Mp <= 32 ? 2 : 4);
  // group sizes 128, 64, 32 (1, 2, 4 groups per 128-k tile) and per-channel scales (r3: the formats nn/qmodule.py:121-129 selects
  // when in_features is not a multiple of 128 or the caller asks for them no longer leave the streaming kernel)
  const bool per_channel = g.C == g.K && g.C != 128;
  const bool grouped = g.C == 128 || ((g.C == 64 || g.C == 32) && g.N % 64 == 0);
  const int groups = per_channel ? (int)(g.K / 128) : (int)g.G;
  if (g.bits == 4 && g.C == 96 && !per_channel)  // group size 96 (r4): tiles of 96 k, 64-feature blocks
    return g.N % 64 == 0 && g.K % 96 == 0 && g.K >= 192 && M >= 1 && M <= QUANTO_HIP_SKINNY_MAX_M &&
           (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) && g.N < (1 << 30) && g.K < (1 << 30) &&
           skinny::lds_bytes(tf, 4, (int)g.G, 4) <= 160 * 1024;
  if (g.bits == 2)  // qint2 (r4): group size 128, 64-feature blocks (16 packed rows x 4 planes)
    return g.C == 128 && g.N % 64 == 0 && g.K % 128 == 0 && M >= 1 && M <= QUANTO_HIP_SKINNY_MAX_M &&
           (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) && g.N < (1 << 30) && g.K < (1 << 30) &&
           skinny::lds_bytes(tf, 4, (int)g.G, 4, 4) <= 160 * 1024;
  return g.bits == 4 && (grouped || per_channel) && (g.N % 16 == 0) && (g.K % 128 == 0) && M >= 1 && M <= QUANTO_HIP_SKINNY_MAX_M &&
         (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) && g.N < (1 << 30) && g.K < (1 << 30) &&
         skinny::lds_bytes(tf, 4, groups, skinny::pick_waves((int)g.N)) <= 160 * 1024;
}

// [counters (zero on entry, zero on exit) | fp32 partial sums]; 0 when the problem is not split
size_t qbits_skinny_workspace(int64_t M, const PackedGeom& g) {
  const int S = skinny_split(g, M);
  if (S == 1) return 0;
  const int tf = M <= 16 ? 1 : (M <= 32 ? 2 : 4);  // M > 64 runs in passes of 64 rows, which reuse the workspace
  return skinny_counter_bytes(g) + (size_t)(g.N / 16) * S * 64 * tf * 16;
}

int qbits_mm_skinny(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t M,
                    const PackedGeom& g, int dtype, bool int_shift, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!qbits_skinny_supported(M, g, dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed)) % 16) return QUANTO_HIP_EALIGN;
  // split-K only with a workspace (whose counter words the caller guarantees to be zero); without one: one block per feature block
  int S = skinny_split(g, M > 64 ? 64 : M);
  if (S > 1 && (!workspace || workspace_bytes < qbits_skinny_workspace(M, g) || reinterpret_cast<uintptr_t>(workspace) % 16)) S = 1;
  const bool per_channel = g.C == g.K && g.C != 128;
  const size_t esize = 2;  // bf16 / fp16
  for (int64_t m0 = 0; m0 < M; m0 += 64) {  // passes of up to 64 rows (stream-ordered: each pass leaves the counters zero)
    const int64_t rows = M - m0 < 64 ? M - m0 : 64;
    skinny::Args a{reinterpret_cast<const uint8_t*>(x) + (size_t)m0 * g.K * esize, packed, scale, shift, bias,
                   reinterpret_cast<uint8_t*>(y) + (size_t)m0 * g.N * esize, (int)rows, (int)g.N, (int)g.K, (int)g.G, S,
                   reinterpret_cast<int*>(workspace),
                   S > 1 ? reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(workspace) + skinny_counter_bytes(g)) : nullptr,
                   per_channel ? 1 : (int)(128 / g.C), per_channel ? 1 : 0, g.bits == 2 ? 4 : 2, (g.C == 96 && !per_channel) ? 96 : 128,
                   // later passes of a multi-pass call re-read the weights from the Infinity Cache: keep them cacheable there
                   env_int("QUANTO_HIP_SKINNY_NT", M <= 64 ? 1 : 0), env_int("QUANTO_HIP_SKINNY_ABLATE", 0),
                   reinterpret_cast<unsigned long long*>(env_ptr("QUANTO_HIP_SKINNY_TIMELINE"))};
    int r;
    if (dtype == QUANTO_HIP_BF16)
      r = int_shift ? skinny::launch_tf<QUANTO_HIP_BF16, true>(a, stream) : skinny::launch_tf<QUANTO_HIP_BF16, false>(a, stream);
    else
      r = int_shift ? skinny::launch_tf<QUANTO_HIP_F16, true>(a, stream) : skinny::launch_tf<QUANTO_HIP_F16, false>(a, stream);
    if (r != QUANTO_HIP_OK) return r;
  }
  return QUANTO_HIP_OK;
}

// ---- several Linears with a shared input in one launch (5 <= M <= 64) --------------------------------------------------------
// Per-call fixed costs of the streaming kernel (launch, first-byte latency, split-K tail: ~7.7 of the 10.6 us of a
// (32,4096,4096) call) are paid once for q/k/v or gate/up, and the wider grid needs a smaller split (gate+up of
// Llama-3-8B: 448 feature blocks -> no split at all).
static PackedGeom multi_geom(int nseg, const int64_t* N, int64_t K) {
  int64_t total = 0;
  for (int i = 0; i < nseg; ++i) total += N[i];
  return make_geom(total, K, 4, 128);
}

bool qbits_skinny_multi_supported(int nseg, const int64_t* N, int64_t M, int64_t K, int dtype) {
  if (nseg < 1 || nseg > skinny::MAX_SEGS || M < 1 || M > 64) return false;
  for (int i = 0; i < nseg; ++i)
    if (N[i] <= 0 || N[i] % 64) return false;  // 4-wave blocks only
  return qbits_skinny_supported(M, multi_geom(nseg, N, K), dtype);
}

size_t qbits_skinny_multi_workspace(int nseg, const int64_t* N, int64_t M, int64_t K) { return qbits_skinny_workspace(M, multi_geom(nseg, N, K)); }

int qbits_mm_skinny_multi(const void* x, int nseg, const uint8_t* const* packed, const void* const* scale, const void* const* shift,
                          const void* const* bias, void* const* y, const int64_t* N, int64_t M, int64_t K, int dtype, bool int_shift,
                          void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!qbits_skinny_multi_supported(nseg, N, M, K, dtype)) return QUANTO_HIP_ENOTSUP;
  const PackedGeom g = multi_geom(nseg, N, K);
  uintptr_t align = reinterpret_cast<uintptr_t>(x);
  skinny::Segs segs;
  int fb = 0;
  for (int i = 0; i < skinny::MAX_SEGS; ++i) {
    const int j = i < nseg ? i : 0;  // unused slots repeat segment 0 and are never selected
    segs.w[i] = packed[j];
    segs.scale[i] = scale[j];
    segs.shift[i] = shift[j];
    segs.bias[i] = bias ? bias[j] : nullptr;
    segs.y[i] = y[j];
    segs.N[i] = (int)N[j];
    segs.first_fb[i] = i < nseg ? fb : 0x7FFFFFFF;
    if (i < nseg) {
      fb += (int)(N[i] / 64);
      align |= reinterpret_cast<uintptr_t>(packed[i]);
    }
  }
  if (align % 16) return QUANTO_HIP_EALIGN;
  int S = skinny_split(g, M);
  if (S > 1 && (!workspace || workspace_bytes < qbits_skinny_workspace(M, g) || reinterpret_cast<uintptr_t>(workspace) % 16)) S = 1;
  skinny::Args a{x, packed[0], scale[0], shift[0], bias ? bias[0] : nullptr, y[0], (int)M, (int)N[0], (int)K, (int)g.G, S,
                 reinterpret_cast<int*>(workspace),
                 S > 1 ? reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(workspace) + skinny_counter_bytes(g)) : nullptr, 1, 0, 2, 128,
                 env_int("QUANTO_HIP_SKINNY_NT", 1), 0, nullptr};
  if (dtype == QUANTO_HIP_BF16)
    return int_shift ? skinny::launch_tf<QUANTO_HIP_BF16, true>(a, stream, &segs, fb) : skinny::launch_tf<QUANTO_HIP_BF16, false>(a, stream, &segs, fb);
  return int_shift ? skinny::launch_tf<QUANTO_HIP_F16, true>(a, stream, &segs, fb) : skinny::launch_tf<QUANTO_HIP_F16, false>(a, stream, &segs, fb);
}

}  // namespace qh
