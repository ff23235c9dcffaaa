// qbytes_mm with QUANTIZED activations: int8 x int8 on v_mfma_i32_16x16x64_i8 and fp8 x fp8 on gfx950's native
// v_mfma_f32_16x16x32_fp8_fp8 - no conversion instruction anywhere, both operands go HBM -> LDS -> MFMA as stored.
//
// y[M,N] = (A[M,K] @ B[N,K]^T) * scales[N]      (tensor/weights/qbytes.py:72-73 -> library/qbytes_mm.py:36-50)
//   int8 x int8: exact int32 accumulation, one fp32 multiply by the (activation scale x weight scale) product, one
//                rounding to the output dtype -> bit-identical to the reference's _int_mm path;
//   fp8  x fp8 : every product of two e4m3/e5m2 values is exact in fp32; fp32 accumulation.
//
// Same structure as qmm_mfma_large.hip (256x256 tile, 8 waves as 2x4, LDS-DMA with counted vmcnt, swizzled 64-byte rows,
// one software-pipelined instruction stream per wave, one barrier per K-tile, LDS-transposed full-line epilogue), with a
// K-tile of 64 bytes per row for BOTH operands (4 stages of 32 KiB): 12 ds_read_b128, 4 DMA issues and 32 (int8) or
// 64 (fp8) MFMAs per wave and K-tile.
#include <cstdlib>
#include <type_traits>

#include "qh_common.h"

#ifndef QH_N8_ABLATE
#define QH_N8_ABLATE 0  // timing experiments only: 1 = no DMA inside the K loop
#endif

namespace qh {
namespace n8 {

#ifdef QH_N8_STAMPS  // scripts/probes/native8_timing.hip: s_memrealtime (100 MHz) per workgroup at entry / loop start / loop end / stores issued / exit
__device__ unsigned long long g_stamps[4096 * 8];
#define QH_N8_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define QH_N8_STAMP(i) do { } while (0)
#endif

constexpr int BK = 64;  // bytes per row and K-tile, both operands
constexpr int STAGES = 4;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) int i32x4;

// M0 is written and not restored (see qmm_mfma_large.hip: nothing else in these kernels reads it, and every scalar
// instruction in the K loop takes an MFMA issue gap).
#ifndef QH_GLDS_POLICY
#define QH_GLDS_POLICY ""  // cache policy bits of the operand DMA (probes: " sc1", " nt", " sc0 sc1": profiles/r06_glds_cache_policy_ab.jsonl)
#endif
__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1" QH_GLDS_POLICY
      :
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

__device__ __forceinline__ int swz64(int row) { return (-(row >> 2)) & 3; }  // 64-byte rows, lanes read chunk lane>>4

enum { K_I8 = 0, K_F8E4M3 = 1, K_F8E5M2 = 2, K_BF16 = 3, K_F16 = 4 };  // K_BF16 / K_F16: dense 16-bit operands, 32 elements per K-tile

template <int KIND>
struct Acc {
  using V = f32x4;
};
template <>
struct Acc<K_I8> {
  using V = i32x4;
};

struct Args {
  const uint8_t* a;   // [M, K] 1 byte per element (2 for the dense 16-bit kinds)
  const uint8_t* w;   // [N, K]
  const void* scale;  // [N] output dtype, or null (= 1)
  const void* bias;   // [N] or null
  void* y;            // [M, N]
  int M, N, K;
  int gm;             // tile raster: consecutive workgroup ids walk down gm tile rows before moving to the next tile column (1 = row-major)
  // split-K (r6, 128-byte-row kernel): S workgroups per tile, workgroup (tile, sp) multiplies the K range sp of S; the accumulators (int32 / fp32) travel
  // fragment-major through `partials`, an arrival counter per tile elects the last workgroup, which adds in split order and runs the epilogue
  // (protocol and workspace contract of qmm_mfma_large.hip / qbits_skinny.hip: counters zero on entry and on exit)
  int S;
  int* counters;      // [tiles]
  void* partials;     // [tiles * S][NJ * 8][threads] 16-byte accumulator quads
  int poll_ticks;     // how long a workgroup waits for its partners (s_memrealtime ticks of 10 ns) before it leaves its slice to the last arriver
};

// Tile raster.  The XCD remap in the kernels hands every XCD (its own 4 MiB L2, 32 CUs) one contiguous range of tile indices; all of an
// XCD's workgroups walk K in step, so its L2 fetches every operand line once per distinct tile row / tile column in that range.  Row-major
// indices make the range 2 tile rows x 16 tile columns at 4096^3 (18 row-panels of traffic per XCD, 72 % L2 hits), and ONE row x 32 columns for
// the 128-tiles of (512,8192,8192) (33 panels); walking `gm` tile rows first turns it into a gm x (32 / gm) block: 4 x 8 or 8 x 4 = 12 panels (measured, r5: L2 misses -27 %, (512,8192,8192)
// int8 / fp8 46.1 / 46.5 -> 42.8 / 43.4 us, int4 prefill 4096^3 104.3 -> 102.4, the square 4096^3 products unchanged).
// The vector L1 keeps ~57 of its 64 miss slots busy in these kernels (requests x latency / cycles, profiles/r05_native8_row128.md), so the
// fill rate is slots x 128 B / latency and the L2 hit rate sets the latency.
__device__ __forceinline__ void tile_of(int bid, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
  const int per_group = gm * tiles_n;
  const int grp = bid / per_group, first = grp * gm;
  const int rows = tiles_m - first < gm ? tiles_m - first : gm;
  const int in = bid - grp * per_group;
  tn = in / rows;
  tm = first + (in - tn * rows);
}

// ---- r6: per-feature scale / bias of the tile, parked in LDS behind the operand ring by the prologue ------------------------------------
// The epilogue used to fetch them from global memory after the K loop: a round trip in front of the first output byte of every tile, and all
// tiles of these grids end together.  One load per thread (threads = 2 * BN), issued in FRONT of the prologue's DMA - the oldest entry of the
// in-order vector-memory queue, so the prologue's counted wait covers it - and stored behind the ring before the prologue's barrier.  As asm:
// a load hipcc can see makes it drain the DMA queue (vmcnt(0)) at the store.  (Same change as qmm_mfma_large.hip: cfg2 -1.4 us, cfg4 -1.1 us.)
template <int ODT, int BN>
struct FeatureTable {
  using T = typename Elem<ODT>::T;
  static constexpr int BYTES = 2 * BN * (int)sizeof(T);  // [scale x BN | bias x BN]
  uint32_t v;
  bool have;
  __device__ __forceinline__ void fetch(const Args& a, int n0, int tid) {
    v = 0;
    have = tid < BN ? a.scale != nullptr : a.bias != nullptr;
    if (tid < 2 * BN && have) {
      int n = n0 + (tid < BN ? tid : tid - BN);
      n = n < a.N ? n : a.N - 1;
      const T* src = reinterpret_cast<const T*>(tid < BN ? a.scale : a.bias) + n;
      if constexpr (sizeof(T) == 2)
        asm volatile("global_load_ushort %0, %1, off" : "=v"(v) : "v"(src) : "memory");
      else
        asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(src) : "memory");
    }
  }
  // after the prologue's vmcnt wait, before its barrier
  __device__ __forceinline__ void park(uint8_t* tab, int tid) {
    asm volatile("" : "+v"(v));
    if (tid < 2 * BN) {
      if constexpr (sizeof(T) == 2) {
        const uint16_t one = ODT == QUANTO_HIP_BF16 ? 0x3F80 : 0x3C00;
        reinterpret_cast<uint16_t*>(tab)[tid] = have ? (uint16_t)v : (tid < BN ? one : (uint16_t)0);
      } else {
        reinterpret_cast<uint32_t*>(tab)[tid] = have ? v : (tid < BN ? 0x3F800000u : 0u);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
};

// ---- epilogue: (int32 | fp32) accumulator * scale[n] (+ bias), parked per wave in LDS, stored as full 128-byte lines ----
// (shared by the 64-byte-row and the 128-byte-row kernels; every wave must be done with the operand stages: the barrier below)
// NI / i0: the token fragments acc[.][0 .. NI-1] are fragments i0 .. i0 + NI - 1 of the wave's 128 rows (the K split hands every workgroup 8 / S of them)
template <int ODT, int KIND, int NJ, int BM, int BN, int NI = 8>
__device__ __forceinline__ void epilogue(const Args& a, typename Acc<KIND>::V (&acc)[NJ][NI], uint8_t* smem, const uint8_t* tabp, int m0, int n0, int wm,
                                         int wn, int wave, int lane, int i0 = 0) {
  using E = Elem<ODT>;
  using T = typename E::T;
  const int M = a.M, N = a.N;
  T* yg = reinterpret_cast<T*>(a.y);
  const bool has_bias = a.bias != nullptr;
  const bool full = (m0 + BM <= M) && (n0 + BN <= N) && (N % 8 == 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  constexpr int FP = NJ * 16 < 128 / (int)sizeof(T) ? NJ * 16 : 128 / (int)sizeof(T);  // features per pass (rows of <= 128 bytes)
  constexpr int JP = FP / 16, PASSES = NJ / JP;  // feature fragments per pass
  constexpr int ROWB = FP * (int)sizeof(T), LPR = ROWB / 16;  // bytes per parked row, lanes per row on the read side
  uint8_t* park = smem + wave * (128 * ROWB);     // <= 16 KiB per wave
  const T* tab = reinterpret_cast<const T*>(tabp);
  // r6: the wave's rows leave in RH halves - the stores of the first half drain while the second half is scaled, converted and parked (as
  // qmm_mfma_large.hip: one pass over all rows ran the phases of all waves in step, first all on the VALU / LDS, then all on the store path)
  constexpr int RH = NI >= 8 ? 2 : 1, NIH = NI / RH;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    float sc[JP][4], bv[JP][4];
#pragma unroll
    for (int jj = 0; jj < JP; ++jj) {
      const int nl = wn * (NJ * 16) + (p * JP + jj) * 16 + (lane >> 4) * 4;  // the lane's four consecutive features inside the tile
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sc[jj][r] = E::to_f32(tab[nl + r]);       // 1.0 without a scale
        bv[jj][r] = E::to_f32(tab[BN + nl + r]);  // 0.0 without a bias (not added below)
      }
    }
#pragma unroll
    for (int h = 0; h < RH; ++h) {
#pragma unroll
      for (int jj = 0; jj < JP; ++jj) {
        const int j = p * JP + jj;
#pragma unroll
        for (int i = h * NIH; i < (h + 1) * NIH; ++i) {
          T out[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = (float)acc[j][i][r] * sc[jj][r];  // library/qbytes_mm.py:47-49: fp32(int32) * fp32(scale), rounded to fp32 ...
            asm volatile("" : "+v"(v));                 // ... and only then to the output dtype (no single-rounding v_fma_mixlo_f16)
            if (has_bias) v = E::to_f32(E::from_f32(v)) + bv[jj][r];
            out[r] = E::from_f32(v);
          }
          const int row = i * 16 + (lane & 15);
          if constexpr (sizeof(T) == 2) {
            const int chunk = (jj * 4 + (lane >> 4)) ^ ((row & (LPR - 1)) << 1);  // 8-byte chunks
            *reinterpret_cast<uint2*>(park + row * ROWB + chunk * 8) = *reinterpret_cast<const uint2*>(out);
          } else {
            const int chunk = (jj * 4 + (lane >> 4)) ^ (row & (LPR - 1));  // 16-byte chunks
            *reinterpret_cast<uint4*>(park + row * ROWB + chunk * 16) = *reinterpret_cast<const uint4*>(out);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      constexpr int TH = 2 * LPR * NIH / 8;  // read / store iterations per half: 64 / LPR rows each
#pragma unroll
      for (int t = h * TH; t < (h + 1) * TH; ++t) {
        const int row = t * (64 / LPR) + lane / LPR;
        const int c16 = lane % LPR;
        uint4 v;
        if constexpr (sizeof(T) == 2)
          v = *reinterpret_cast<const uint4*>(park + row * ROWB + (((c16 * 2) ^ ((row & (LPR - 1)) << 1)) * 8));
        else
          v = *reinterpret_cast<const uint4*>(park + row * ROWB + ((c16 ^ (row & (LPR - 1))) * 16));
        const int m = m0 + wm * 128 + i0 * 16 + row;
        const int n = n0 + wn * (NJ * 16) + p * FP + c16 * (16 / (int)sizeof(T));
        if (full) {
          typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;  // non-temporal: see qmm_mfma_large.hip
          __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(yg + (size_t)m * N + n));
        } else if (m < M) {
          const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
          for (int r = 0; r < 16 / (int)sizeof(T); ++r)
            if (n + r < N) yg[(size_t)m * N + n + r] = e[r];
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next pass parks into the rows this one has just read
  }
}

// ---- split-K tail (r6) ---------------------------------------------------------------------------------------------------------------
// The S workgroups of a tile each hold a full partial accumulator tile (int32 / fp32).  Reducing it in ONE of them (the last arriver reads S x 256 KiB through
// one CU: 1 MB at ~70 GB/s per workgroup, MI355X_MICROARCH.md "handoff-payload") cost more than the split saved (measured: (512,8192,8192) fp8 52 us against 44
// unsplit).  Here every workgroup reduces and stores ONE slice of the tile - token fragments sp * 8 / S .. of every wave:
//   1. store the partial tile write-through (sc0 sc1), drain, arrive on the tile's counter;
//   2. poll (one lane, relaxed system-scope loads, s_sleep) until all S have arrived - the usual case, the S workgroups of a tile are dispatched together
//      (launch plan: tiles * S <= the workgroups the chip holds at once);
//   3. add the S partial slices in split order - the same sum whoever computes it: run-to-run identical bits - and run the epilogue on the slice.
// Nothing depends on co-residency for CORRECTNESS: a workgroup whose poll times out (another stream holds the CUs its partners need) marks its slice
// abandoned and leaves; the LAST arriver - for which every partial is in memory by construction - finishes the abandoned slices.  State words per tile
// (zero on entry and on exit): [arrivals, completions, state of slice 0 .. S-1 (0 = owner still waiting, 1 = owner reduces it, 2 = abandoned)].
// part 1, common to every S: the partial tile goes out.  (Kept out of the per-S code below on purpose: with the accumulators live into three inlined reductions hipcc
// spilled 130 - 470 registers per lane; after this function they are dead.)
template <int KIND, int NJ, int NWAVES>
__device__ __forceinline__ void splitk_store(const Args& a, typename Acc<KIND>::V (&acc)[NJ][8], int S, int tile_lin, int sp, int tid) {
  using AV = typename Acc<KIND>::V;
  constexpr int NT = NWAVES * 64, NF = NJ * 8;
  const uint8_t* part = reinterpret_cast<const uint8_t*>(a.partials) + ((size_t)tile_lin * S + sp) * NF * NT * 16;
  const uint32_t lane_off = (uint32_t)tid * 16;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // every wave is done with the operand stages: the flag words of part 2 live there
  asm volatile("" ::: "memory");
  // the whole partial tile, the own slice included: whoever reduces a slice runs the same code on the same bytes, and no second copy of the own slice has to stay in
  // registers next to the 32 quads a slice reduction has in flight
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint32_t t;  // (the add sits inside the asm: hipcc otherwise computes all 32 offsets ahead of the stores)
      // s_nop: a store of more than 8 bytes reads its data registers for a few cycles after issue; they are dead here and the next asm's temporary may be one of them
      // (the hazard recognizer does not look into inline asm: without the wait states the 128-tile kernel stored wrong first / last dwords, r6 visit 5)
      asm volatile("v_add_u32_e32 %0, %3, %1\n\tglobal_store_dwordx4 %0, %2, %4 sc0 sc1\n\ts_nop 1" : "=&v"(t) : "v"(lane_off), "v"(acc[j][i]), "s"((j * 8 + i) * NT * 16), "s"(part) : "memory");
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

template <int ODT, int KIND, int NJ, int BM, int BN, int NWAVES, int SS>
__device__ __forceinline__ void splitk_reduce(const Args& a, uint8_t* smem, const uint8_t* tabp, int tile_lin, int sp, int m0, int n0, int wm, int wn, int wave,
                                              int lane, int tid) {
  using AV = typename Acc<KIND>::V;
  constexpr int NI = 8 / SS, NT = NWAVES * 64, NF = NJ * 8, CW = 2 + SS;
  const unsigned long long POLL_LIMIT = (unsigned long long)a.poll_ticks;  // s_memrealtime ticks (100 MHz); 20000 = 200 us
  int* cw = a.counters + tile_lin * CW;
  // quad of (split q, fragment f) of this thread: ONE scalar base per tile + a 32-bit lane offset (tid + (q * NF + f) * NT quads): 64-bit lane addresses for the 32
  // quads in flight would cost 64 registers next to the 128 they load into, a scalar base per quad runs the kernel out of SGPRs
  const uint8_t* part = reinterpret_cast<const uint8_t*>(a.partials) + (size_t)tile_lin * SS * NF * NT * 16;
  const uint32_t lane_off = (uint32_t)tid * 16;
  auto ld = [&](AV& v, uint32_t base_off, int quads) __attribute__((always_inline)) {
    uint32_t t;
    asm volatile("v_add_u32_e32 %1, %3, %2\n\tglobal_load_dwordx4 %0, %1, %4 sc0 sc1" : "=&v"(v), "=&v"(t) : "v"(base_off), "s"(quads * NT * 16), "s"(part) : "memory");
  };
  volatile int* flag = reinterpret_cast<volatile int*>(smem);
  if (tid == 0) flag[0] = __hip_atomic_fetch_add(cw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  const bool last = flag[0] == SS - 1;
  if (!last) {
    if (tid == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      int ok = 0;
      for (;;) {
        if (__hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == SS) {
          ok = 1;
          break;
        }
        if (__builtin_amdgcn_s_memrealtime() - t0 > POLL_LIMIT) break;
        __builtin_amdgcn_s_sleep(8);
      }
      // 1 = this workgroup reduces its slice, 2 = abandoned (partners not on the chip): the last arriver takes it
      __hip_atomic_store(cw + 2 + sp, ok ? 1 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      flag[1] = ok;
    }
    __syncthreads();
    if (flag[1] == 0) return;
  }
  // slice s of the tile: the S partial slices added in split order
  auto do_slice = [&](int s) __attribute__((always_inline)) {
    AV L[SS][NJ][NI];
    const uint32_t slice_off = lane_off + (uint32_t)s * (NI * NT * 16);
#pragma unroll
    for (int q = 0; q < SS; ++q)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) ld(L[q][j][ii], slice_off, q * NF + j * 8 + ii);
#pragma unroll
    for (int q = 0; q < SS; ++q)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) asm volatile("s_waitcnt vmcnt(0)" : "+v"(L[q][j][ii])::"memory");  // ties the uses below to the wait
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int ii = 0; ii < NI; ++ii)
#pragma unroll
        for (int q = 1; q < SS; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) L[0][j][ii][r] += L[q][j][ii][r];
    // opaque zero: the epilogue's per-feature addresses and scale / bias values are common to the three split instantiations, the unsplit epilogue and the sweep
    // loop below; hipcc otherwise computes them once ahead of all of them and carries ~150 registers through the reduction (measured: 130 - 470 spills)
    int zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    epilogue<ODT, KIND, NJ, BM, BN, NI>(a, L[0], smem, tabp, m0, n0 + zero, wm, wn, wave, lane, s * NI);
  };
  // completion: S slices + the last arriver's sweep; whoever counts the last one leaves the state words as found
  auto complete = [&]() {
    __syncthreads();
    if (tid == 0 && __hip_atomic_fetch_add(cw + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == SS) {
#pragma unroll
      for (int w = 0; w < CW; ++w) __hip_atomic_store(cw + w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  };
  do_slice(sp);
  complete();
  if (!last) return;
#pragma unroll 1
  for (int s = 0; s < SS; ++s) {
    if (s == sp) continue;
    __syncthreads();  // the flag words sit in the parking area of the epilogue
    if (tid == 0) {
      int v;
      while ((v = __hip_atomic_load(cw + 2 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) == 0) __builtin_amdgcn_s_sleep(2);  // its owner is running: 1 or 2 follows
      flag[2] = v;
    }
    __syncthreads();
    if (flag[2] == 2) {
      do_slice(s);
      complete();
    }
  }
  complete();
}

// PAIRED (fp8 kinds, K % 128 == 0): K-tiles are consumed two at a time by the K = 128 MX-format MFMA (unit scales), which runs
// at twice the rate of the 16x16x32 fp8 MFMA - see the paired loop below.
// SMALL: 128x128 tile with four waves of 128 x 32 (16 KiB stages: two workgroups share a CU) for shapes whose 256-tiles cannot
// occupy the chip; otherwise 256x256 with eight waves of 128 x 64.
template <int ODT, int KIND, bool PAIRED = false, bool SMALL = false>
__global__ void __launch_bounds__(SMALL ? 256 : 512, 1) qbytes_native8_kernel(const Args a) {
  using AV = typename Acc<KIND>::V;
  constexpr int BM = SMALL ? 128 : 256, BN = BM;
  constexpr int NWAVES = SMALL ? 4 : 8;
  constexpr int NJ = SMALL ? 2 : 4;  // 16-feature fragments per wave; 8 token fragments per wave in both layouts
  constexpr int A_BYTES = BM * BK, W_BYTES = BN * BK, STAGE_BYTES = A_BYTES + W_BYTES;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = SMALL ? 0 : wave >> 2, wn = wave & 3;
  constexpr int ES = (KIND == K_BF16 || KIND == K_F16) ? 2 : 1;  // operand element size; a K-tile is always 64 bytes per row
  const int M = a.M, N = a.N, K = a.K;
  const int nk = K * ES / BK;

  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int nwg = tiles_n * tiles_m;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  tile_of(bid, tiles_m, tiles_n, a.gm, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  constexpr int RING_BYTES = STAGES * STAGE_BYTES;
  FeatureTable<ODT, BN> ftab;  // scale / bias of the tile: fetched ahead of the DMA, parked behind the ring before the prologue's barrier
  static_assert(NWAVES * 64 == 2 * BN, "one table entry per thread");
  ftab.fetch(a, n0, tid);

  // ---- DMA: 2 + 2 pieces of 1 KiB per wave and K-tile; piece j of an operand covers tile rows (j*8+wave)*16 .. +15 -------
  uint32_t asrc[2], wsrc[2];
  int adst[2], wdst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int R = (j * NWAVES + wave) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz64(R);
    int m = m0 + R, n = n0 + R;
    m = m < M ? m : M - 1;
    n = n < N ? n : N - 1;
    asrc[j] = (uint32_t)((size_t)m * K * ES + c * 16);
    wsrc[j] = (uint32_t)((size_t)n * K * ES + c * 16);
    adst[j] = (j * NWAVES + wave) * 1024;
    wdst[j] = A_BYTES + (j * NWAVES + wave) * 1024;
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  // DMA destinations of the four pieces (2 activation + 2 weight) in each stage: loop constants, kept in SGPRs
  uint32_t mdst[STAGES][4];
#pragma unroll
  for (int t = 0; t < STAGES; ++t)
#pragma unroll
    for (int p = 0; p < 4; ++p)
      mdst[t][p] = __builtin_amdgcn_readfirstlane(lds_base + t * STAGE_BYTES + (p < 2 ? adst[p] : wdst[p - 2]));
  auto issue_piece_to = [&](int kt, const uint32_t (&dst)[4], int piece) {
    if (piece < 2)
      glds16(a.a + (size_t)kt * BK, asrc[piece], dst[piece]);
    else
      glds16(a.w + (size_t)kt * BK, wsrc[piece - 2], dst[piece]);
  };
  auto issue = [&](int kt, int stage) {  // prologue only (run-time stage)
    const uint32_t st = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE_BYTES);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(a.a + (size_t)kt * BK, asrc[j], st + adst[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(a.w + (size_t)kt * BK, wsrc[j], st + wdst[j]);
  };

  // ---- fragment reads: ONE ds_read_b128 per 16-row fragment and K-tile (bytes k = 16g .. 16g+15, g = lane >> 4); the
  // swizzle only depends on (row & 15) >> 2, so fragment i / j adds a compile-time multiple of 1024 bytes
  const int ra = wm * 128 + (lane & 15), rw = wn * (NJ * 16) + (lane & 15);
  const int aoff0 = ra * 64 + (((lane >> 4) ^ swz64(ra)) << 4);
  const int boff0 = A_BYTES + rw * 64 + (((lane >> 4) ^ swz64(rw)) << 4);
  // The K loops are unrolled over the four stages (tile kt lives in stage kt & 3): fragment bases per stage are loop
  // constants in registers, a K-tile carries no address arithmetic (run-time stage bookkeeping was ~40 of the ~190
  // instructions per two tiles and wave)
  // (two base registers per operand: the 16-bit offset field of ds_read reaches stage t from the base of stage t & ~1)
  const uint8_t* xb2[2] = {smem + aoff0, smem + 2 * STAGE_BYTES + aoff0};
  const uint8_t* wb2[2] = {smem + boff0, smem + 2 * STAGE_BYTES + boff0};
  auto xbs = [&](int t) -> const uint8_t* { return xb2[t >> 1] + (t & 1) * STAGE_BYTES; };
  auto wbs = [&](int t) -> const uint8_t* { return wb2[t >> 1] + (t & 1) * STAGE_BYTES; };

  // acc[j][i]: the weight fragment is the MFMA A operand (rows = output features), the activation fragment the B operand
  AV acc[NJ][8];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = AV{0, 0, 0, 0};

  // Single software-pipelined stream per wave (same scheme as qmm_mfma_large.hip): per K-tile 8 steps, step i = the
  // MFMAs of token fragment i against the four resident weight fragments, each MFMA followed by at most one LDS read or
  // one DMA issue.  Fragments are fetched a full K-tile ahead (with 32 MFMAs per tile a two-step distance is shorter
  // than the LDS latency under load and the stream stalls on every fragment): the activation fragment of step i is
  // replaced by the next tile's as soon as step i has issued its MFMAs; the weight fragments, live for the whole tile,
  // ping-pong between two register sets.  One barrier per K-tile.
  uint4 xf[8], wq[2][NJ];
  auto read_x = [&](const uint8_t* st, int i) -> uint4 { return *reinterpret_cast<const uint4*>(st + aoff0 + i * 1024); };
  auto read_w = [&](const uint8_t* st, int j) -> uint4 { return *reinterpret_cast<const uint4*>(st + boff0 + j * 1024); };
  auto mma = [&](AV& c, const uint4& w, const uint4& x) {
    if constexpr (KIND == K_I8) {
      c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, w), __builtin_bit_cast(i32x4, x), c, 0, 0, 0);
    } else if constexpr (KIND == K_BF16) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
    } else if constexpr (KIND == K_F16) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
    } else {
      const long wlo = (long)(((unsigned long)w.y << 32) | w.x), whi = (long)(((unsigned long)w.w << 32) | w.z);
      const long xlo = (long)(((unsigned long)x.y << 32) | x.x), xhi = (long)(((unsigned long)x.w << 32) | x.z);
      if constexpr (KIND == K_F8E4M3) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wlo, xlo, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(whi, xhi, c, 0, 0, 0);
      } else {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(wlo, xlo, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(whi, xhi, c, 0, 0, 0);
      }
    }
  };

  if constexpr (PAIRED) {
    // ---- fp8 x fp8 on v_mfma_scale_f32_16x16x128_f8f6f4 --------------------------------------------------------------------
    // An operand is 32 bytes per lane: the lane's 16 bytes of tile 2p (k = 16g..16g+15 of that tile) followed by its 16 bytes
    // of tile 2p+1 - the same k assignment in both operands.  Pair p = 32 MFMAs per wave in two halves:
    //   half A: token fragments 0..3 x weight fragments 0..3, while X[4..7] of THIS pair are fetched;
    //   (own DMA share of tiles 2p+2, 2p+3 landed -> barrier -> refill the stages of tiles 2p, 2p+1 with tiles 2p+4, 2p+5)
    //   half B: weight-fragment-major over token fragments 4..7; X[0..3] and each W[j], dead after its last MFMA, are
    //           refetched for pair p+1.
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    constexpr int FMTSEL = KIND == K_F8E5M2 ? 1 : 0;  // cbsz / blgp: 0 = fp8 (e4m3), 1 = bf8 (e5m2)
    i32x8 X[8], W[NJ];
    auto load_pair = [&](i32x8& dst, const uint8_t* s0, const uint8_t* s1, int off) {
      const uint4 lo = *reinterpret_cast<const uint4*>(s0 + off), hi = *reinterpret_cast<const uint4*>(s1 + off);
      dst = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    };
    auto mma128 = [&](AV& c, const i32x8& w, const i32x8& x) {
      if constexpr (KIND == K_F8E4M3 || KIND == K_F8E5M2) {
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, x, c, FMTSEL, FMTSEL, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);  // scales 2^0
      } else {  // other kinds: the two tiles of the pair as two MFMAs (same schedule, half as many barriers as the per-tile loop)
        const uint4 wl = make_uint4(w[0], w[1], w[2], w[3]), wh = make_uint4(w[4], w[5], w[6], w[7]);
        const uint4 xl = make_uint4(x[0], x[1], x[2], x[3]), xh = make_uint4(x[4], x[5], x[6], x[7]);
        mma(c, wl, xl);
        mma(c, wh, xh);
      }
    };
    const int np = nk >> 1;
    // prologue: tiles 0..3 in flight, pair 0 visible, its W and X[0..3] in registers
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < nk) issue(t, t);
    if (nk > 2)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ftab.park(smem + RING_BYTES, tid);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < NJ; ++j) load_pair(W[j], smem, smem + STAGE_BYTES, boff0 + j * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_pair(X[i], smem, smem + STAGE_BYTES, aoff0 + i * 1024);
    // (this loop keeps run-time stage pointers: unrolled over the pair parity it spills - 8-register operand tuples - and a
    // pair of 32 K = 128 MFMAs amortises the bookkeeping)
    for (int p = 0; p < np; ++p) {
      const uint8_t* c0 = smem + ((2 * p) & 3) * STAGE_BYTES;
      const uint8_t* c1 = smem + ((2 * p + 1) & 3) * STAGE_BYTES;
      const uint8_t* n0s = smem + ((2 * p + 2) & 3) * STAGE_BYTES;
      const uint8_t* n1s = smem + ((2 * p + 3) & 3) * STAGE_BYTES;
      // ---- half A ----
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          mma128(acc[j][i], W[j], X[i]);
          if (j == 1) load_pair(X[4 + i], c0, c1, aoff0 + (4 + i) * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (p + 1 < np) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tiles 2p+2, 2p+3 (issued one pair ago; nothing younger is in flight)
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // X[4..7] returned: this wave is done reading tiles 2p, 2p+1
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (2 * p + 4 < nk) issue(2 * p + 4, (2 * p) & 3);
      if (2 * p + 5 < nk) issue(2 * p + 5, (2 * p + 1) & 3);
      // ---- half B ----
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int i = 4; i < 8; ++i) {
          mma128(acc[j][i], W[j], X[i]);
          // X[0..3] of the next pair (dead since half A): 4 / NJ of them behind each weight fragment's MFMAs
          if (i - 4 < 4 / NJ) load_pair(X[j * (4 / NJ) + (i - 4)], n0s, n1s, aoff0 + (j * (4 / NJ) + (i - 4)) * 1024);
          if (i == 7) load_pair(W[j], n0s, n1s, boff0 + j * 1024);            // W[j] of the next pair: its last MFMA was just issued
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
  // prologue: tiles 0, 1, 2 in flight; tiles 0 and 1 visible (tile 1 feeds the prefetches issued during tile 0)
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 2) {
      issue(2, 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    ftab.park(smem + RING_BYTES, tid);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < NJ; ++j) wq[0][j] = read_w(smem, j);
#pragma unroll
    for (int i = 0; i < 8; ++i) xf[i] = read_x(smem, i);

    // tile kt (stage kt & 3): refill the stage of tile kt-1 with tile kt+3, prefetch from the stage of tile kt+1, and before
    // the closing barrier wait for the own DMA share of tile kt+2 (tile kt+3 may stay in flight)
    auto tile = [&](int kt, auto stage_tag, bool dma, int wait_mode /* 2: vmcnt(4), 1: vmcnt(0), 0: none */, bool barrier) {
      constexpr int ST = decltype(stage_tag)::value, P = ST & 1, SN = (ST + 1) & 3, SF = (ST + 3) & 3;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          mma(acc[j][i], wq[P][j], xf[i]);
          if (j == 1 % NJ) {
            if (i < NJ) wq[P ^ 1][i] = *reinterpret_cast<const uint4*>(wbs(SN) + i * 1024);
          }
          if (j == 2 % NJ) {
#if QH_N8_ABLATE != 1
            if (i >= 4 && dma) issue_piece_to(kt + 3, mdst[SF], i - 4);
#endif
          }
          if (j == NJ - 1) xf[i] = *reinterpret_cast<const uint4*>(xbs(SN) + i * 1024);  // same fragment of the next tile
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (wait_mode == 2)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (wait_mode == 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (barrier) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    };
    using st0 = std::integral_constant<int, 0>;
    using st1 = std::integral_constant<int, 1>;
    using st2 = std::integral_constant<int, 2>;
    using st3 = std::integral_constant<int, 3>;
    int kt = 0;
    for (; kt + 6 < nk; kt += 4) {  // steady state: every one of the four tiles still has a tile kt+3 to fetch
      tile(kt, st0{}, true, 2, true);
      tile(kt + 1, st1{}, true, 2, true);
      tile(kt + 2, st2{}, true, 2, true);
      tile(kt + 3, st3{}, true, 2, true);
    }
    // at most six tiles left (kt is a multiple of 4): nothing, or less, to prefetch - straight-line so the stage stays static
#define QH_TAIL_TILE(J, STAGE)                                                                                           \
    if (kt + (J) < nk)                                                                                                     \
      tile(kt + (J), STAGE{}, kt + (J) + 3 < nk, kt + (J) + 3 < nk ? 2 : (kt + (J) + 2 < nk ? 1 : 0), kt + (J) + 1 < nk)
    QH_TAIL_TILE(0, st0);
    QH_TAIL_TILE(1, st1);
    QH_TAIL_TILE(2, st2);
    QH_TAIL_TILE(3, st3);
    QH_TAIL_TILE(4, st0);
    QH_TAIL_TILE(5, st1);
#undef QH_TAIL_TILE

  }

  epilogue<ODT, KIND, NJ, BM, BN>(a, acc, smem, smem + RING_BYTES, m0, n0, wm, wn, wave, lane);
}

// =============================================================================================================================
// 128-byte rows (r5).  The kernel above stages 64 bytes per row and K-tile: every 128-byte line of an operand row is fetched by
// two instructions a K-tile apart, i.e. as two half-line vector-L1 fills (16.7 M requests per dense bf16 4096^3 launch against
// the vendor kernel's 8.4 M full-line ones, profiles/r04_l1_fill_counters_vendor_vs_library.json).  Here a DMA piece is 8 rows x
// 128 bytes - eight adjacent lanes fetch one whole line - and the LDS holds TWO buffers of 128-byte rows (a "pair" of the 64-byte
// k-steps the MFMA stream still works in):
//   * image: [row][128 B], lane l of a piece = (row l >> 3, slot l & 7); the lane fetches chunk slot ^ h(row), h = (row >> 1) & 7, and a
//     fragment lane reads chunk c = 4 * half + (lane >> 4) at slot c ^ h(row) - conflict-free for the ds_read_b128 lane groups of
//     gfx950 and byte-exact, both checked on the CPU by scripts/models/row128_lds_model.py;
//   * int8 / 16-bit stream: the k-step pipeline of the kernel above (all twelve fragments of k-step s+1 fetched during k-step s, in
//     place), with the buffer bookkeeping per pair: even step = second half of the current buffer, odd step = first half of the
//     other buffer + the 8 DMA pieces of pair p+2 into the buffer that was just read out.  ONE barrier per pair (before the odd
//     step: it publishes pair p+1 and retires the reads of pair p), half as many as before;
//   * fp8: one v_mfma_scale_f32_16x16x128_f8f6f4 per fragment pair and 128-byte tile, as in the paired loop above.
// Needs K * element size % 128 == 0; everything else stays on the 64-byte-row kernel (QUANTO_HIP_NATIVE8_ROW128=0 forces that one).
// =============================================================================================================================
__device__ __forceinline__ int swz128(int row) { return (row >> 1) & 7; }

// Tried and dropped (r5, profiles/r05_native8_row128.md): the eight DMA pieces of a pair behind the first four token fragments of the odd
// step instead of one behind each of the eight (w8a8 4096^3 53.0 -> 54.6 us: two back-to-back DMA issues stall the MFMA stream for longer than
// the earlier landing saves), and no lgkmcnt(0) in front of the barrier (52.9 vs 53.0: the wait is free, so the rigorous form stays).
template <int ODT, int KIND, bool SMALL = false>
__global__ void __launch_bounds__(SMALL ? 256 : 512, 1) qbytes_native8_r128_kernel(const Args a) {
  using AV = typename Acc<KIND>::V;
  constexpr bool MX = KIND == K_F8E4M3 || KIND == K_F8E5M2;
  constexpr int BM = SMALL ? 128 : 256, BN = BM;
  constexpr int NWAVES = SMALL ? 4 : 8;
  constexpr int NJ = SMALL ? 2 : 4;
  constexpr int RB = 128;                                   // bytes per row and pair
  // LDS: [activations buffer 0 | activations buffer 1 | weights buffer 0 | weights buffer 1], 32 KiB each (16 KiB for the 128-tile): both
  // buffers of an operand are within the 16-bit offset field of ds_read from ONE base register
  constexpr int OP_BYTES = BM * RB, W_BASE = 2 * OP_BYTES;
  constexpr int PPW = BM / 8 / NWAVES;                      // pieces per wave, operand and pair: 4 in both layouts
  static_assert(PPW == 4, "piece bookkeeping below assumes four pieces per wave and operand");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  QH_N8_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = SMALL ? 0 : wave >> 2, wn = wave & 3;
  constexpr int ES = (KIND == K_BF16 || KIND == K_F16) ? 2 : 1;
  const int M = a.M, N = a.N, K = a.K;
  const int S = a.S;
  const int np = K * ES / RB / S;  // pairs (128-byte K-tiles) of this workgroup's K range

  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int nwg = tiles_n * tiles_m * S;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // the S workgroups of a tile are neighbours in the remapped order: same XCD (they share the tile's operand panels' neighbours in L2)
  const int tile_lin = S > 1 ? bid / S : bid, sp = S > 1 ? bid - tile_lin * S : 0;
  const size_t kbase = (size_t)sp * np * RB;  // byte offset of this K range inside an operand row
  int tm, tn;
  tile_of(tile_lin, tiles_m, tiles_n, a.gm, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  constexpr int RING_BYTES = 4 * OP_BYTES;
  FeatureTable<ODT, BN> ftab;  // scale / bias of the tile: fetched ahead of the DMA, parked behind the ring before the prologue's barrier
  static_assert(NWAVES * 64 == 2 * BN, "one table entry per thread");
  ftab.fetch(a, n0, tid);

  // ---- DMA: 4 + 4 pieces of 1 KiB per wave and pair; piece j of an operand covers tile rows (j * NWAVES + wave) * 8 .. + 7 ----
  uint32_t asrc[PPW], wsrc[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int R = (j * NWAVES + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz128(R);
    int m = m0 + R, n = n0 + R;
    m = m < M ? m : M - 1;
    n = n < N ? n : N - 1;
    asrc[j] = (uint32_t)((size_t)m * K * ES + c * 16);
    wsrc[j] = (uint32_t)((size_t)n * K * ES + c * 16);
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  uint32_t mdst[2][2 * PPW];  // DMA destinations per buffer: loop constants, kept in SGPRs
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int q = 0; q < 2 * PPW; ++q)
      mdst[b][q] = __builtin_amdgcn_readfirstlane(lds_base + (q < PPW ? 0 : W_BASE) + b * OP_BYTES + ((q % PPW) * NWAVES + wave) * 1024);
  auto issue_piece = [&](int p, const uint32_t (&dst)[2 * PPW], int q) {  // piece q of pair p: q < 4 activations, else weights
    if (q < PPW)
      glds16(a.a + kbase + (size_t)p * RB, asrc[q], dst[q]);
    else
      glds16(a.w + kbase + (size_t)p * RB, wsrc[q - PPW], dst[q]);
  };

  // ---- fragment reads: one ds_read_b128 per 16-row fragment and k-step; chunk 4 * half + (lane >> 4) of the lane's row ----
  const int ra = wm * 128 + (lane & 15), rw = wn * (NJ * 16) + (lane & 15);
  const int aoff = ra * RB + ((((lane >> 4) ^ swz128(ra)) & 7) << 4);            // half 0; half 1 = ^ 64
  const int boff = W_BASE + rw * RB + ((((lane >> 4) ^ swz128(rw)) & 7) << 4);
  constexpr int FR = 16 * RB;  // bytes between consecutive 16-row fragments

  AV acc[NJ][8];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = AV{0, 0, 0, 0};

  // prologue (both loops): pairs 0 and 1 in flight, pair 0 visible
#pragma unroll
  for (int q = 0; q < 2 * PPW; ++q) issue_piece(0, mdst[0], q);
  if (np > 1) {
#pragma unroll
    for (int q = 0; q < 2 * PPW; ++q) issue_piece(1, mdst[1], q);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  ftab.park(smem + RING_BYTES, tid);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  QH_N8_STAMP(1);
  if constexpr (MX) {
    // ---- fp8 x fp8: an operand is the lane's 16 bytes of the first half followed by its 16 bytes of the second half of the row ----
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    constexpr int FMTSEL = KIND == K_F8E5M2 ? 1 : 0;  // cbsz / blgp: 0 = fp8 (e4m3), 1 = bf8 (e5m2)
    i32x8 X[8], W[NJ];
    auto load_pair = [&](i32x8& dst, const uint8_t* buf, int off) {
      const uint4 lo = *reinterpret_cast<const uint4*>(buf + off), hi = *reinterpret_cast<const uint4*>(buf + (off ^ 64));
      dst = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    };
    auto mma128 = [&](AV& c, const i32x8& w, const i32x8& x) {
      c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, x, c, FMTSEL, FMTSEL, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);  // scales 2^0
    };
#pragma unroll
    for (int j = 0; j < NJ; ++j) load_pair(W[j], smem, boff + j * FR);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_pair(X[i], smem, aoff + i * FR);
    for (int p = 0; p < np; ++p) {
      const uint8_t* cur = smem + (p & 1) * OP_BYTES;
      const uint8_t* nxt = smem + ((p + 1) & 1) * OP_BYTES;
      // ---- half A: token fragments 0..3, while X[4..7] of this pair are fetched ----
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          mma128(acc[j][i], W[j], X[i]);
          if (j == 1) load_pair(X[4 + i], cur, aoff + (4 + i) * FR);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (p + 1 < np) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own share of pair p+1 (issued one pair ago; nothing younger in flight)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // X[4..7] returned: this wave is done reading the current buffer
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (p + 2 < np) {
        if (p & 1) {
#pragma unroll
          for (int q = 0; q < 2 * PPW; ++q) issue_piece(p + 2, mdst[1], q);
        } else {
#pragma unroll
          for (int q = 0; q < 2 * PPW; ++q) issue_piece(p + 2, mdst[0], q);
        }
      }
      // ---- half B: weight-fragment-major over token fragments 4..7; X[0..3] and each W[j] refetched for pair p+1 ----
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int i = 4; i < 8; ++i) {
          mma128(acc[j][i], W[j], X[i]);
          if (i - 4 < 4 / NJ) load_pair(X[j * (4 / NJ) + (i - 4)], nxt, aoff + (j * (4 / NJ) + (i - 4)) * FR);
          if (i == 7) load_pair(W[j], nxt, boff + j * FR);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
    // ---- int8 / 16-bit: k-steps of 64 bytes; xf is refilled in place, the weight fragments ping-pong between two register sets ----
    uint4 xf[8], wq[2][NJ];
    auto rd = [&](const uint8_t* p) -> uint4 { return *reinterpret_cast<const uint4*>(p); };
    auto mma = [&](AV& c, const uint4& w, const uint4& x) {
      if constexpr (KIND == K_I8) {
        c = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, w), __builtin_bit_cast(i32x4, x), c, 0, 0, 0);
      } else if constexpr (KIND == K_BF16) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
      } else {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
      }
    };
    // fragment bases per half: loop constants in registers (the 16-bit offset field of ds_read carries buffer and fragment index)
    const uint8_t* xh[2] = {smem + aoff, smem + (aoff ^ 64)};
    const uint8_t* wh[2] = {smem + boff, smem + (boff ^ 64)};
    auto xp = [&](int b, int h) -> const uint8_t* { return xh[h] + b * OP_BYTES; };
    auto wp = [&](int b, int h) -> const uint8_t* { return wh[h] + b * OP_BYTES; };
#pragma unroll
    for (int j = 0; j < NJ; ++j) wq[0][j] = rd(wp(0, 0) + j * FR);
#pragma unroll
    for (int i = 0; i < 8; ++i) xf[i] = rd(xp(0, 0) + i * FR);

    // pair p, resident in buffer B: [even step: k-step 2p, prefetching k-step 2p+1 from the same buffer] -> own DMA share of pair p+1
    // landed, own reads of buffer B returned -> barrier -> [odd step: k-step 2p+1, prefetching k-step 2p+2 from the other buffer and
    // refilling buffer B with pair p+2, one DMA piece behind each token fragment's MFMAs]
    auto pair = [&](int p, auto btag, bool has_next, bool has_dma) {
      constexpr int B = decltype(btag)::value;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          mma(acc[j][i], wq[0][j], xf[i]);
          if (j == 1 % NJ) {
            if (i < NJ) wq[1][i] = rd(wp(B, 1) + i * FR);
          }
          if (j == NJ - 1) xf[i] = rd(xp(B, 1) + i * FR);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (has_next) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          mma(acc[j][i], wq[1][j], xf[i]);
          if (j == 1 % NJ) {
            if (i < NJ && has_next) wq[0][i] = rd(wp(B ^ 1, 0) + i * FR);
          }
          if (j == 2 % NJ) {
            if (has_dma) issue_piece(p + 2, mdst[B], i);
          }
          if (j == NJ - 1 && has_next) xf[i] = rd(xp(B ^ 1, 0) + i * FR);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    using b0 = std::integral_constant<int, 0>;
    using b1 = std::integral_constant<int, 1>;
    int p = 0;
    for (; p + 3 < np; p += 2) {  // steady state: both pairs have a successor and a pair p+2 to fetch
      pair(p, b0{}, true, true);
      pair(p + 1, b1{}, true, true);
    }
    // at most three pairs left (p is even): straight-line, so that the buffer index stays static
    if (p < np) pair(p, b0{}, p + 1 < np, p + 2 < np);
    if (p + 1 < np) pair(p + 1, b1{}, p + 2 < np, false);
    if (p + 2 < np) pair(p + 2, b0{}, false, false);
  }

  QH_N8_STAMP(2);
  if (S > 1) {
    splitk_store<KIND, NJ, NWAVES>(a, acc, S, tile_lin, sp, tid);
    if (S == 2)
      splitk_reduce<ODT, KIND, NJ, BM, BN, NWAVES, 2>(a, smem, smem + RING_BYTES, tile_lin, sp, m0, n0, wm, wn, wave, lane, tid);
    else if (S == 4)
      splitk_reduce<ODT, KIND, NJ, BM, BN, NWAVES, 4>(a, smem, smem + RING_BYTES, tile_lin, sp, m0, n0, wm, wn, wave, lane, tid);
    else
      splitk_reduce<ODT, KIND, NJ, BM, BN, NWAVES, 8>(a, smem, smem + RING_BYTES, tile_lin, sp, m0, n0, wm, wn, wave, lane, tid);
    return;
  }
  epilogue<ODT, KIND, NJ, BM, BN>(a, acc, smem, smem + RING_BYTES, m0, n0, wm, wn, wave, lane);
#ifdef QH_N8_STAMPS
  QH_N8_STAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  QH_N8_STAMP(4);
#endif
}

template <int ODT, int KIND, bool SMALL>
static int launch_r128(const Args& a, hipStream_t stream) {
  constexpr int T = SMALL ? 128 : 256;
  constexpr int need = 2 * 2 * T * 128 + FeatureTable<ODT, T>::BYTES;  // two buffers of 128-byte rows: 128 KiB (64 KiB for the 128-tile; the epilogue parks in it) + scale / bias table
  const int tiles = ((a.N + T - 1) / T) * ((a.M + T - 1) / T);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbytes_native8_r128_kernel<ODT, KIND, SMALL>), hipFuncAttributeMaxDynamicSharedMemorySize, need);
  hipLaunchKernelGGL((qbytes_native8_r128_kernel<ODT, KIND, SMALL>), dim3(tiles * a.S), dim3(SMALL ? 256 : 512), need, stream, a);
  return launch_status();
}

static int raster_group() {
  const int g = env_int("QUANTO_HIP_NATIVE8_GROUP_M", 8);  // experiments: 1 = row-major
  return g < 1 ? 1 : g;
}

template <int ODT, int KIND, bool PAIRED, bool SMALL>
static int launch_cfg(const Args& a, hipStream_t stream) {
  constexpr int T = SMALL ? 128 : 256;
  constexpr int need = STAGES * 2 * T * BK + FeatureTable<ODT, T>::BYTES;  // 128 KiB (64 KiB for the 128-tile: two workgroups per CU; the epilogue parks in it) + scale / bias table
  const int tiles = ((a.N + T - 1) / T) * ((a.M + T - 1) / T);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbytes_native8_kernel<ODT, KIND, PAIRED, SMALL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, need);
  hipLaunchKernelGGL((qbytes_native8_kernel<ODT, KIND, PAIRED, SMALL>), dim3(tiles), dim3(SMALL ? 256 : 512), need, stream, a);
  return launch_status();
}

// ---- launch plan: tile size and K split ------------------------------------------------------------------------------------------------
// Measured on the 128-byte-row kernels (r6, profiles/r06_native8_split_k.md; us, int8 / fp8, hipGraph replay):
//   * tile: 128-tiles win whenever all of them are resident at once (two per CU), for int8 as for fp8 - (2048,4096,4096) 42.6 -> 35.3 / 32.1,
//     (512,14336,4096) 41.5 -> 32.2 / 30.3, (768,8192,4096) 41.4 -> 30.3 / 28.2, (1024,8192,8192) 75.2 -> 66.4 / 60.6 (r5's opposite finding for int8 was taken
//     on the 64-byte-row kernel);
//   * K split: every split workgroup sends its 64 KiB (256 KiB for a 256-tile) accumulator tile through memory and back - S x M x N x 8 bytes of fabric traffic
//     next to (M + N) x K operand bytes.  It pays where K is long for the output it feeds and the unsplit grid leaves CUs or residency slots idle:
//     (512,4096,14336) 68.5 / 70.2 -> 48.0 / 44.5, (256,8192,8192) 41 -> 32, (128,4096,4096) 18.1 / 21.6 -> 13.4 / 14.3, (32,4096,14336) 67 -> 28; it loses on
//     everything squarer ((1024,4096,4096) 25 -> 38 with four-way split 256-tiles: 64 MB of partials for a 38 MB problem), and 256-tiles split S ways never beat
//     128-tiles split S / 2 ways (their partial tile is four times the size).  QUANTO_HIP_NATIVE8_SMALL / _SPLIT force a configuration (tests, sweeps).
struct Plan {
  bool small;
  int S;
};
template <int KIND>
static Plan make_plan(const Args& a, bool have_ws) {
  constexpr int ES = (KIND == K_BF16 || KIND == K_F16) ? 2 : 1;
  constexpr bool BYTE = ES == 1;
  const int small_env = env_int("QUANTO_HIP_NATIVE8_SMALL", -1), split_env = env_int("QUANTO_HIP_NATIVE8_SPLIT", 0);  // experiments / tests
  const int64_t tiles256 = (int64_t)((a.N + 255) / 256) * ((a.M + 255) / 256), tiles128 = (int64_t)((a.N + 127) / 128) * ((a.M + 127) / 128);
  const bool row128 = (a.K * ES) % 128 == 0 && env_int("QUANTO_HIP_NATIVE8_ROW128", 1) != 0;
  // 8-bit operands from 128-byte rows: 128-tiles as long as all of them are resident at once; 16-bit operands (the dense GEMM behind int4 prefill) and the
  // 64-byte-row kernel: 256-tiles when they give every CU at least ~3/8 of a tile
  Plan p{small_env >= 0 ? small_env != 0 : (BYTE && row128 ? tiles128 <= 512 : tiles256 < 96), 1};
  if (!row128 || !have_ws || split_env == 1) return p;
  const int np = a.K * ES / 128;
  // S in {2, 4, 8} (a slice = 8 / S token fragments of every wave), 2 + S state words per tile in the counter region, and all tiles * S workgroups on the chip at
  // once (256 CUs x one 256-tile / two 128-tile workgroups): the tail does not need that to be correct, only to be fast
  auto fits = [&](int64_t tiles, int S, bool small_tile) {
    return (S == 2 || S == 4 || S == 8) && np % S == 0 && np / S >= 2 && tiles * (2 + S) * 4 <= (int64_t)QUANTO_HIP_WS_COUNTER_BYTES &&
           tiles * S <= (small_tile ? 512 : 256);
  };
  if (split_env > 1) {  // forced: with the tile QUANTO_HIP_NATIVE8_SMALL asks for (default: 256-tiles)
    p.small = small_env > 0;
    p.S = fits(p.small ? tiles128 : tiles256, split_env, p.small) ? split_env : 1;
    return p;
  }
  if (!BYTE || !p.small) return p;
  // AUTO (128-tiles): four ways for a handful of tiles (K >= 4096) or a long K (>= 3072 per range), two ways while a range keeps K >= 4096
  if (tiles128 <= 64 && np >= 32 && fits(tiles128, 4, true)) return Plan{true, 4};
  if (np >= 96 && fits(tiles128, 4, true)) return Plan{true, 4};
  if (np >= 64 && tiles128 <= 128 && fits(tiles128, 2, true)) return Plan{true, 2};  // (beyond one split workgroup per CU the partial traffic eats the gain: (512,8192,8192) 44.1 -> 46.0)
  return p;
}
template <int KIND>
static size_t plan_workspace(const Plan& p, int64_t M, int64_t N) {
  if (p.S <= 1) return 0;
  const int T = p.small ? 128 : 256;
  const int64_t tiles = ((N + T - 1) / T) * ((M + T - 1) / T);
  return (size_t)QUANTO_HIP_WS_COUNTER_BYTES + (size_t)tiles * p.S * ((size_t)T * T * 4);  // one 4-byte accumulator per tile element and split
}

template <int ODT, int KIND>
static int launch(Args a, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  constexpr int ES = (KIND == K_BF16 || KIND == K_F16) ? 2 : 1;
  const bool ws_ok = workspace && reinterpret_cast<uintptr_t>(workspace) % 16 == 0;
  Plan p = make_plan<KIND>(a, ws_ok);
  if (p.S > 1 && workspace_bytes < plan_workspace<KIND>(p, a.M, a.N)) p = make_plan<KIND>(a, false);
  a.S = p.S;
  a.counters = p.S > 1 ? reinterpret_cast<int*>(workspace) : nullptr;
  a.partials = p.S > 1 ? reinterpret_cast<uint8_t*>(workspace) + QUANTO_HIP_WS_COUNTER_BYTES : nullptr;
  a.poll_ticks = env_int("QUANTO_HIP_NATIVE8_POLL_TICKS", 20000);  // tests: 0 = nobody waits, the last arriver reduces every slice it finds abandoned
  const bool small = p.small;
  // 128-byte rows (full-line vector-L1 fills, one barrier per 128 bytes of K) whenever K allows
  if ((a.K * ES) % 128 == 0 && env_int("QUANTO_HIP_NATIVE8_ROW128", 1) != 0)
    return small ? launch_r128<ODT, KIND, true>(a, stream) : launch_r128<ODT, KIND, false>(a, stream);
  // 64-byte rows: K * element size = 64 (mod 128).  (r6: the PAIRED instantiations of this kernel - two 64-byte tiles per K = 128 MX-format MFMA - were
  // reachable only where the 128-byte-row kernel applies as well, i.e. through QUANTO_HIP_NATIVE8_ROW128=0, and left the product library; the
  // loop stays in the source for probes built with -DQH_N8_EXPERIMENTS)
#ifdef QH_N8_EXPERIMENTS
  constexpr bool FP8 = KIND == K_F8E4M3 || KIND == K_F8E5M2;
  if (FP8 && (a.K * ES) % 128 == 0 && env_int("QUANTO_HIP_PAIRED", 1) != 0)
    return small ? launch_cfg<ODT, KIND, true, true>(a, stream) : launch_cfg<ODT, KIND, true, false>(a, stream);
#endif
  return small ? launch_cfg<ODT, KIND, false, true>(a, stream) : launch_cfg<ODT, KIND, false, false>(a, stream);
}

}  // namespace n8

// Dense 16-bit GEMM y = x @ w^T (+ bias) on the same pipeline; used by qbits_mm for prefill-sized M after the packed
// weight has been dequantized (bit-identically to the reference's dequantize()) into the caller's workspace.
bool dense_mm_large_supported(int64_t M, int64_t N, int64_t K, int dtype) {
  return (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) && K % 32 == 0 && K >= 32 && M >= 1 && M * K < (1ll << 30) &&
         N * K < (1ll << 30) && M < (1 << 30) && N < (1 << 30);
}

int dense_mm_large(const void* x, const void* w, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int dtype, hipStream_t stream) {
  if (!dense_mm_large_supported(M, N, K, dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16) return QUANTO_HIP_EALIGN;
  n8::Args args{reinterpret_cast<const uint8_t*>(x), reinterpret_cast<const uint8_t*>(w), nullptr, bias, y, (int)M, (int)N, (int)K, n8::raster_group(), 1, nullptr, nullptr, 0};
  if (dtype == QUANTO_HIP_BF16) return n8::launch<QUANTO_HIP_BF16, n8::K_BF16>(args, nullptr, 0, stream);
  return n8::launch<QUANTO_HIP_F16, n8::K_F16>(args, nullptr, 0, stream);
}

bool qbytes_native8_supported(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  const bool pair = (a_dtype == QUANTO_HIP_I8 && b_dtype == QUANTO_HIP_I8) ||
                    (a_dtype == QUANTO_HIP_F8_E4M3FN && b_dtype == QUANTO_HIP_F8_E4M3FN) ||
                    (a_dtype == QUANTO_HIP_F8_E5M2 && b_dtype == QUANTO_HIP_F8_E5M2);
  const bool od = out_dtype == QUANTO_HIP_BF16 || out_dtype == QUANTO_HIP_F16 || out_dtype == QUANTO_HIP_F32;
  return pair && od && K % n8::BK == 0 && K >= n8::BK && M >= 1 && M * K < (1ll << 31) && N * K < (1ll << 31) && M < (1 << 30) &&
         N < (1 << 30);
}

static int native8_kind(int a_dtype) { return a_dtype == QUANTO_HIP_I8 ? n8::K_I8 : a_dtype == QUANTO_HIP_F8_E4M3FN ? n8::K_F8E4M3 : n8::K_F8E5M2; }

// split-K scratch for this problem: [QUANTO_HIP_WS_COUNTER_BYTES of arrival counters, zero on entry and on exit | partial accumulator tiles]; 0 = the plan does not split
size_t qbytes_native8_workspace(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  if (!qbytes_native8_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return 0;
  n8::Args args{nullptr, nullptr, nullptr, nullptr, nullptr, (int)M, (int)N, (int)K, 1, 1, nullptr, nullptr, 0};
  switch (native8_kind(a_dtype)) {
    case n8::K_I8: return n8::plan_workspace<n8::K_I8>(n8::make_plan<n8::K_I8>(args, true), M, N);
    case n8::K_F8E4M3: return n8::plan_workspace<n8::K_F8E4M3>(n8::make_plan<n8::K_F8E4M3>(args, true), M, N);
    default: return n8::plan_workspace<n8::K_F8E5M2>(n8::make_plan<n8::K_F8E5M2>(args, true), M, N);
  }
}

int qbytes_mm_native8(const void* a, const void* b, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int a_dtype,
                      int b_dtype, int out_dtype, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!qbytes_native8_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16) return QUANTO_HIP_EALIGN;
  n8::Args args{reinterpret_cast<const uint8_t*>(a), reinterpret_cast<const uint8_t*>(b), s, bias, y, (int)M, (int)N, (int)K, n8::raster_group(), 1, nullptr, nullptr, 0};
#define QH_KIND(ODT)                                                                  \
  if (a_dtype == QUANTO_HIP_I8) return n8::launch<ODT, n8::K_I8>(args, workspace, workspace_bytes, stream);       \
  if (a_dtype == QUANTO_HIP_F8_E4M3FN) return n8::launch<ODT, n8::K_F8E4M3>(args, workspace, workspace_bytes, stream); \
  return n8::launch<ODT, n8::K_F8E5M2>(args, workspace, workspace_bytes, stream)
  if (out_dtype == QUANTO_HIP_BF16) { QH_KIND(QUANTO_HIP_BF16); }
  if (out_dtype == QUANTO_HIP_F16) { QH_KIND(QUANTO_HIP_F16); }
  QH_KIND(QUANTO_HIP_F32);
#undef QH_KIND
}

}  // namespace qh
