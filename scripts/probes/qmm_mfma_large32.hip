// NOT part of libquanto_hip.so since r5 (the r2 experiment behind DESIGN 4.3a: same tile on v_mfma_f32_32x32x16, 7.5 % slower at the part's power limit).
// Kept as a probe: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I optimum_quanto_amd/csrc -c scripts/probes/qmm_mfma_large32.hip
//
// qbytes_mm MFMA GEMM on v_mfma_f32_32x32x16_{bf16,f16}: 256x256x64 tile, eight waves as 2 (tokens) x 4 (features), each
// owning 128 tokens x 64 features = 2 x 4 accumulator blocks of 32 x 32 (128 accumulator registers).
//
// y[M,N] = (x[M,K] @ q[N,K]^T) * scale[N]   with x bf16/fp16 and q int8 / fp8 (1 byte per weight).
//
// Why a second large-tile kernel next to qmm_mfma_large.hip (16x16x32 MFMAs): the weight conversion costs the same VALU work
// per flop in both shapes (it only depends on the 128 tokens a wave covers: 3 VALU per pair of weights, each converted
// fragment feeding 4 token blocks), but a 32x32x16 MFMA keeps the matrix pipe busy for 32 cycles instead of 16 - the same
// conversion, LDS-read and DMA instructions have twice as many cycles of cover per issue slot they take (8 issue slots per
// MFMA instead of 4; the hardware hides about 5 fillers behind a 32-cycle MFMA), and the back-to-back issue ceiling of the
// shape is higher (MI355X_MICROARCH.md: 32x32x16 2.38-2.49 PFLOP/s vs 2.08 for 16x16x32).  r01 counters of the 16x16x32 kernel at
// 4096^3: MFMA pipe 63 % busy, 40 % of wave cycles issue stalls, 1.8 VALU + 0.24 SALU per MFMA.
//
// Per K-tile (64 k) and wave: 32 MFMAs (4 k-steps of 16 x 2 feature blocks x 4 token blocks), 96 conversion VALU,
// 16 + 4 ds_read_b128, 12 LDS-DMA pieces.  One software-pipelined instruction stream, one workgroup barrier per K-tile:
//   step s = (k-step t, token block i), 16 per K-tile: 2 MFMAs acc[j][i] += W_t[j] * x(i, t), j = 0, 1;
//   behind each MFMA one converted dword (3 VALU) of the NEXT k-step's weight operands (double-buffered wop[2]), behind the
//   first MFMA of a step the activation fragment of step s+2, behind the second one DMA piece of tile kt+2 (steps 0-11) or a
//   16-byte raw weight fragment of tile kt+1 (as soon as its register is dead).
// k inside a K-tile is permuted (the same way for both operands, so the contraction is unchanged): k-step t, lane half g
// (lane >> 5) use the 8-element chunk c(t, g) = 4 (t >> 1) + 2 g + (t & 1) - a lane's weight bytes for k-steps 2u and 2u+1
// are then 16 contiguous bytes (one ds_read_b128 per feature block and pair of k-steps).
// LDS images are linear per DMA instruction; bank-conflict swizzles (derived for the 4 x 16 lane groups of ds_read_b128
// on gfx950) sit on the DMA source address and are undone on the read: activations (128-byte rows, 8 chunks)
// chunk ^ ((row >> 1) & 7), weights (64-byte rows, 4 chunks) chunk ^ ((row >> 2) & 3).
#include <type_traits>

#include "qmm_large_common.h"

namespace qh {
namespace lt32 {

using lt::Args;
using lt::BK;
using lt::convert_pair;
using lt::glds16;
using lt::lds_ptr_t;
using lt::STAGES;
using lt::tile_coords;

typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int DT>
struct Mma32;
template <>
struct Mma32<QUANTO_HIP_BF16> {
  using V8 = bf16x8;
  static __device__ __forceinline__ f32x16 run(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <>
struct Mma32<QUANTO_HIP_F16> {
  using V8 = f16x8;
  static __device__ __forceinline__ f32x16 run(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ int swz_a32(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_w32(int row) { return (row >> 2) & 3; }

constexpr int BM = 256, BN = 256, WN = 4, NWAVES = 8;  // waves as 2 (token halves) x WN
constexpr int MI = 4;   // 32-token blocks per wave
constexpr int NJ = 2;   // 32-feature blocks per wave
constexpr int KS = 4;   // k-steps of 16 per K-tile
constexpr int STEPS = KS * MI;
constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK, STAGE_BYTES = A_BYTES + W_BYTES;
constexpr int APIECES = BM / 8 / NWAVES;   // 4: activation DMA pieces (8 rows x 128 B) per wave and K-tile
constexpr int WPIECES = BN / 16 / NWAVES;  // 2: weight DMA pieces (16 rows x 64 B)
constexpr int NPIECES = APIECES + WPIECES;
static_assert(NPIECES <= STEPS, "one DMA piece per step");

template <int DT, int FMT>
__global__ void __launch_bounds__(NWAVES * 64, 1) qbytes_mfma_large32_kernel(const Args a) {
  using E = Elem<DT>;
  using T = typename E::T;
  using V8 = typename Mma32<DT>::V8;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int M = a.M, N = a.N, K = a.K;
  const int nk = K / BK;
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  int tm, tn;
  tile_coords(blockIdx.x, tiles_m, tiles_n, a.group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- DMA sources: per K-tile 32 activation pieces + 16 weight pieces of 1 KiB; 4 + 2 per wave -----------------------------
  uint32_t asrc[APIECES], wsrc[WPIECES];
#pragma unroll
  for (int j = 0; j < APIECES; ++j) {
    const int R = (j * NWAVES + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz_a32(R);
    int m = m0 + R;
    m = m < M ? m : M - 1;
    asrc[j] = (uint32_t)(((size_t)m * K + c * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < WPIECES; ++j) {
    const int R = (j * NWAVES + wave) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz_w32(R);
    int n = n0 + R;
    n = n < N ? n : N - 1;
    wsrc[j] = (uint32_t)((size_t)n * K + c * 16);
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.x);

  // ---- fragment read offsets --------------------------------------------------------------------------------------------------
  const int li = lane & 31, lg = lane >> 5;
  const int ra = wm * (MI * 32) + li, rw = wn * (NJ * 32) + li;
  int aoff[KS];  // activation chunk of k-step t: c(t, g) = 4 (t >> 1) + 2 g + (t & 1); block i adds i * 32 rows = i * 4096 bytes
#pragma unroll
  for (int t = 0; t < KS; ++t) aoff[t] = ra * 128 + (((4 * (t >> 1) + 2 * lg + (t & 1)) ^ swz_a32(ra)) << 4);
  int woff[2];   // raw weight bytes of k-steps 2u, 2u+1: 16-byte chunk 2u + g of the 64-byte row; block j adds j * 2048 bytes
#pragma unroll
  for (int u = 0; u < 2; ++u) woff[u] = A_BYTES + rw * 64 + (((2 * u + lg) ^ swz_w32(rw)) << 4);

  f32x16 acc[NJ][MI];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  uint4 raw[NJ][2];          // raw[j][u]: .xy = k-step 2u, .zw = k-step 2u+1 (8 weight bytes each)
  uint32_t wop[2][NJ][4];    // converted operands of k-step t in wop[t & 1]
  V8 xf[4];                  // activation fragments, two steps ahead
  auto as_v8 = [&](const uint32_t(&w)[4]) { return __builtin_bit_cast(V8, make_uint4(w[0], w[1], w[2], w[3])); };
  auto rawword = [&](int j, int t, int d) -> uint32_t {  // dword d (0..3) of the 4-dword operand <- bytes 2d, 2d+1 of a raw dword
    const uint4& r = raw[j][t >> 1];
    const uint32_t lo = (t & 1) ? r.z : r.x, hi = (t & 1) ? r.w : r.y;
    return d < 2 ? lo : hi;
  };

  // stage-dependent addresses as loop constants (the K loop is unrolled over the three stages: no address arithmetic inside)
  const uint8_t* xb[STAGES][KS];
  const uint8_t* wb[STAGES][2];
  uint32_t mdst[STAGES][NPIECES];
#pragma unroll
  for (int st = 0; st < STAGES; ++st) {
#pragma unroll
    for (int t = 0; t < KS; ++t) xb[st][t] = smem + st * STAGE_BYTES + aoff[t];
#pragma unroll
    for (int u = 0; u < 2; ++u) wb[st][u] = smem + st * STAGE_BYTES + woff[u];
#pragma unroll
    for (int p = 0; p < NPIECES; ++p)
      mdst[st][p] = __builtin_amdgcn_readfirstlane(
          lds_base + st * STAGE_BYTES + (p < APIECES ? (p * NWAVES + wave) * 1024 : A_BYTES + ((p - APIECES) * NWAVES + wave) * 1024));
  }
  auto issue_piece = [&](int kt, int stage, int piece) {
    if (piece < APIECES)
      glds16(xbase + (size_t)kt * (BK * 2), asrc[piece], mdst[stage][piece]);
    else
      glds16(a.w + (size_t)kt * BK, wsrc[piece - APIECES], mdst[stage][piece]);
  };

  // ---- prologue: tiles 0 and 1 in flight, both complete, raw(0) loaded, W_0 of tile 0 converted, x(0..1, t0) in registers ----
#pragma unroll
  for (int p = 0; p < NPIECES; ++p) issue_piece(0, 0, p);
  if (nk > 1) {
#pragma unroll
    for (int p = 0; p < NPIECES; ++p) issue_piece(1, 1, p);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int u = 0; u < 2; ++u) raw[j][u] = *reinterpret_cast<const uint4*>(wb[0][u] + j * 2048);
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int d = 0; d < 4; ++d) wop[0][j][d] = convert_pair<DT, FMT>(rawword(j, 0, d), d & 1);
  xf[0] = *reinterpret_cast<const V8*>(xb[0][0]);
  xf[1] = *reinterpret_cast<const V8*>(xb[0][0] + 4096);

  // One K-tile.  The source order IS the schedule: a sched_barrier after every MFMA slot keeps hipcc from clustering the
  // conversions in front of the MFMAs (see qmm_mfma_large.hip).
  auto tile = [&](auto p_tag, int kt, auto dma_tag, auto barrier_tag) {
    constexpr int P = decltype(p_tag)::value, PN = (P + 1) % STAGES, PF = (P + 2) % STAGES;
    const bool DMA = dma_tag, BARRIER = barrier_tag;  // integral_constants in the steady state, run-time flags in the tail
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int t = s / MI, i = s % MI;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[j][i] = Mma32<DT>::run(as_v8(wop[t & 1][j]), xf[s & 3], acc[j][i]);
        {
          // conversion slot: dword c of the 8 operand dwords of the next k-step (this tile's t+1, or the next tile's t = 0 - its
          // raw[.][0] was reloaded during k-step 1)
          const int c = i * NJ + j, f = c >> 2, d = c & 3, tn = (t + 1) % KS;
          wop[(t + 1) & 1][f][d] = convert_pair<DT, FMT>(rawword(f, tn, d), d & 1);
        }
        if (j == 0) {
          // activation fragment of step s+2 (the first two of the next tile at the end; garbage, unused, on the last tile)
          const int s2 = s + 2;
          xf[s2 & 3] = s2 < STEPS ? *reinterpret_cast<const V8*>(xb[P][s2 / MI] + (s2 % MI) * 4096)
                                  : *reinterpret_cast<const V8*>(xb[PN][0] + (s2 - STEPS) * 4096);
        } else {
          if (s < NPIECES) {
            if (DMA) {
              if (s < APIECES)
                glds16(xbase + (size_t)(kt + 2) * (BK * 2), asrc[s], mdst[PF][s]);
              else
                glds16(a.w + (size_t)(kt + 2) * BK, wsrc[s - APIECES], mdst[PF][s]);
            }
          }
          // next tile's raw weight bytes: raw[.][0] (k-steps 0, 1) is dead once k-step 1's operands are converted, i.e. after
          // k-step 0; raw[.][1] (k-steps 2, 3) after k-step 2
          if (s == MI + 2) raw[0][0] = *reinterpret_cast<const uint4*>(wb[PN][0]);
          if (s == MI + 3) raw[1][0] = *reinterpret_cast<const uint4*>(wb[PN][0] + 2048);
          if (s == 3 * MI) raw[0][1] = *reinterpret_cast<const uint4*>(wb[PN][1]);
          if (s == 3 * MI + 1) raw[1][1] = *reinterpret_cast<const uint4*>(wb[PN][1] + 2048);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (BARRIER) {
      // tile boundary: the own DMA share of tile kt+2 has landed -> barrier -> everybody's share visible and every wave done
      // with tile kt, whose stage the next tile refills (same discipline as qmm_mfma_large.hip: tile kt+1 reads its
      // successor's raw bytes and first activation fragments while it runs, so tile kt+2 must be complete before it starts)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using yes = std::integral_constant<bool, true>;
  static_assert(STAGES == 3, "the loop below is unrolled over three stages");
  int kt = 0;
  for (; kt + 4 < nk; kt += 3) {
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

  // ---- epilogue: scale (+bias) on the fp32 accumulator, park the wave's 128 tokens x 64 features in LDS (128-byte rows), store
  // whole lines.  D layout of the 32x32 MFMA: lane (col = lane & 31 -> token, g = lane >> 5), register r -> feature
  // (r & 3) + 8 (r >> 2) + 4 g: four runs of 4 consecutive features per block.
  T* yg = reinterpret_cast<T*>(a.y);
  const bool has_bias = a.bias != nullptr, has_scale = a.scale != nullptr;
  const bool full = (m0 + BM <= M) && (n0 + BN <= N) && (N % 8 == 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  uint8_t* park = smem + wave * (MI * 32 * 128);  // 16 KiB per wave
  const int nw0 = n0 + wn * (NJ * 32);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nb = nw0 + j * 32 + 8 * q + 4 * lg;
      float sc[4], bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb + r < N ? nb + r : N - 1;
        sc[r] = has_scale ? E::to_f32(reinterpret_cast<const T*>(a.scale)[n]) : 1.f;
        bv[r] = has_bias ? E::to_f32(reinterpret_cast<const T*>(a.bias)[n]) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        T out[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[j][i][q * 4 + r] * sc[r];
          asm volatile("" : "+v"(v));  // product rounded to fp32 first, with and without bias (no single-rounding v_fma_mixlo_f16)
          if (has_bias) v = E::to_f32(E::from_f32(v)) + bv[r];
          out[r] = E::from_f32(v);
        }
        // 8 bytes = features 8q + 4g .. +3 of block j: 16-byte chunk j*4 + q of the row, half g.  chunk ^ (row & 7) and
        // half ^ bit 3 of the row: the 16 lanes of a ds_write_b64 group then cover 16 distinct 8-byte slots (no conflict)
        const int row = i * 32 + li;
        const int pos = (((j * 4 + q) ^ (row & 7)) << 4) + ((lg ^ ((row >> 3) & 1)) << 3);
        *reinterpret_cast<uint2*>(park + row * 128 + pos) = *reinterpret_cast<const uint2*>(out);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private region: no barrier needed
#pragma unroll
  for (int it = 0; it < MI * 32 / 8; ++it) {
    const int row = it * 8 + (lane >> 3);
    const int p = lane & 7, c = p ^ (row & 7);  // position p of the row holds chunk c = features (c >> 2) * 32 + (c & 3) * 8 .. +7
    uint4 v = *reinterpret_cast<const uint4*>(park + row * 128 + (p << 4));
    if ((row >> 3) & 1) v = make_uint4(v.z, v.w, v.x, v.y);  // halves were swapped on the way in
    const int m = m0 + wm * (MI * 32) + row;
    const int n = nw0 + (c >> 2) * 32 + (c & 3) * 8;
    if (full) {
      typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
      __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(yg + (size_t)m * N + n));
    } else if (m < M) {
      const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (n + r < N) yg[(size_t)m * N + n + r] = e[r];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int DT, int FMT>
static int launch(const Args& a, hipStream_t stream) {
  constexpr int lds = STAGES * STAGE_BYTES;
  static_assert(lds >= NWAVES * MI * 32 * 128, "the epilogue parks the output tile in the stage memory");
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN, tiles = tiles_m * tiles_n;
  Args b = a;
  {
    const int band = (tiles + 7) / 8;
    int g = 1;
    while ((g + 1) * (g + 1) * 2 * BM <= band * BN) ++g;
    const int forced = env_int("QUANTO_HIP_GROUP_M", 0);  // experiments
    if (forced > 0) g = forced;
    b.group_m = g < tiles_m ? g : tiles_m;
  }
  b.S = 1;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbytes_mfma_large32_kernel<DT, FMT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((qbytes_mfma_large32_kernel<DT, FMT>), dim3(tiles), dim3(NWAVES * 64), lds, stream, b);
  return launch_status();
}

}  // namespace lt32

// same argument contract as qbytes_mm_mfma_large (256-tile configuration, no split-K)
int qbytes_mm_mfma_large32(const void* x, const void* w, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int b_dtype,
                           int out_dtype, hipStream_t stream) {
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16) return QUANTO_HIP_EALIGN;
  lt::Args a{x, reinterpret_cast<const uint8_t*>(w), s, bias, y, (int)M, (int)N, (int)K, 1, 1, nullptr, nullptr};
#define QH_CASE(DT, FMT) return lt32::launch<DT, FMT>(a, stream)
  if (out_dtype == QUANTO_HIP_BF16) {
    if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_BF16, lt::W_I8);
    if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_BF16, lt::W_F8E4M3);
    QH_CASE(QUANTO_HIP_BF16, lt::W_F8E5M2);
  }
  if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_F16, lt::W_I8);
  if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_F16, lt::W_F8E4M3);
  QH_CASE(QUANTO_HIP_F16, lt::W_F8E5M2);
#undef QH_CASE
}

}  // namespace qh
