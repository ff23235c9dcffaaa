// qbytes_mm with QUANTIZED activations: int8 x int8 on v_mfma_i32_16x16x64_i8 and fp8 x fp8 on gfx950's native
// v_mfma_f32_16x16x32_fp8_fp8 - no conversion instruction anywhere, both operands go HBM -> LDS -> MFMA as stored.
//
// y[M,N] = (A[M,K] @ B[N,K]^T) * scales[N]      (tensor/weights/qbytes.py:72-73 -> library/qbytes_mm.py:36-50)
//   int8 x int8: exact int32 accumulation, one fp32 multiply by the (activation scale x weight scale) product, one
//                rounding to the output dtype -> bit-identical to the reference's _int_mm path;
//   fp8  x fp8 : every product of two e4m3/e5m2 values is exact in fp32; fp32 accumulation.
//
// Same LDS image as qmm_mfma_large.hip (256x256 tile, 8 waves as 2x4, LDS-DMA with counted vmcnt, swizzled 64-byte rows,
// alternating load / compute phases with waves 4-7 one phase behind waves 0-3, LDS-transposed full-line epilogue), but a
// K-tile of 64 bytes per row for BOTH operands (4 stages of 32 KiB) and one load + one compute phase per K-tile:
// 12 ds_read_b128 and 32 (int8) or 64 (fp8) MFMAs per wave.
#include "qh_common.h"

#ifndef QH_N8_ABLATE
#define QH_N8_ABLATE 0  // experiments only: 1 = no DMA in the steady loop, 2 = no MFMA, 3 = no fragment reads
#endif

namespace qh {
namespace n8 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int A_BYTES = BM * BK;  // 16 KiB
constexpr int W_BYTES = BN * BK;  // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
constexpr int STAGES = 4;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) int i32x4;

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

__device__ __forceinline__ int swz64(int row) { return (-(row >> 2)) & 3; }  // 64-byte rows, lanes read chunk lane>>4

enum { K_I8 = 0, K_F8E4M3 = 1, K_F8E5M2 = 2 };

template <int KIND>
struct Acc {
  using V = f32x4;
};
template <>
struct Acc<K_I8> {
  using V = i32x4;
};

struct Args {
  const uint8_t* a;   // [M, K] 1 byte per element
  const uint8_t* w;   // [N, K]
  const void* scale;  // [N] output dtype
  const void* bias;   // [N] or null
  void* y;            // [M, N]
  int M, N, K;
};

template <int ODT, int KIND>
__global__ void __launch_bounds__(512, 1) qbytes_native8_kernel(const Args a) {
  using E = Elem<ODT>;
  using T = typename E::T;
  using AV = typename Acc<KIND>::V;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, grp = wave >> 2;
  const int M = a.M, N = a.N, K = a.K;
  const int nk = K / BK;

  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int nwg = tiles_n * tiles_m;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- DMA: 2 + 2 pieces of 1 KiB per wave and K-tile; piece j of an operand covers tile rows (j*8+wave)*16 .. +15 -------
  uint32_t asrc[2], wsrc[2];
  int adst[2], wdst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int R = (j * 8 + wave) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz64(R);
    int m = m0 + R, n = n0 + R;
    m = m < M ? m : M - 1;
    n = n < N ? n : N - 1;
    asrc[j] = (uint32_t)((size_t)m * K + c * 16);
    wsrc[j] = (uint32_t)((size_t)n * K + c * 16);
    adst[j] = (j * 8 + wave) * 1024;
    wdst[j] = A_BYTES + (j * 8 + wave) * 1024;
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  auto issue = [&](int kt, int stage) {
    const uint32_t st = __builtin_amdgcn_readfirstlane(lds_base + stage * STAGE_BYTES);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(a.a + (size_t)kt * BK, asrc[j], st + adst[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(a.w + (size_t)kt * BK, wsrc[j], st + wdst[j]);
  };

  // ---- fragment reads: ONE ds_read_b128 per 16-row fragment and K-tile (bytes k = 16g .. 16g+15, g = lane >> 4) --------
  int aoff[8], boff[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int R = wm * 128 + i * 16 + (lane & 15);
    aoff[i] = R * 64 + (((lane >> 4) ^ swz64(R)) << 4);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int R = wn * 64 + j * 16 + (lane & 15);
    boff[j] = A_BYTES + R * 64 + (((lane >> 4) ^ swz64(R)) << 4);
  }

  // acc[j][i]: the weight fragment is the MFMA A operand (rows = output features), the activation fragment the B operand
  AV acc[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = AV{0, 0, 0, 0};

  uint4 xa[8], wq[4];
  auto read_frags = [&](const uint8_t* st) {
#pragma unroll
    for (int j = 0; j < 4; ++j) wq[j] = *reinterpret_cast<const uint4*>(st + boff[j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) xa[i] = *reinterpret_cast<const uint4*>(st + aoff[i]);
  };
  auto end_load_phase = [&]() {
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

  // prologue: tiles 0, 1, 2 in flight; tile 0 must be visible before the first load phase
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  if (nk > 2) issue(2, 2);
  if (nk > 2)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nk > 1)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp == 1) end_compute_phase();  // G1 runs one phase behind G0

  // Per K-tile and wave: L (DMA of tile kt+3, fragment reads of tile kt, wait for the own share of tile kt+1) then
  // C (32 / 64 MFMAs).  Tile kt+1 is visible to everybody once all waves passed the barrier that ends their L(kt);
  // the stage of tile kt-1 is free once all waves passed the barrier that ends their L(kt-1)... + one more slot for G1,
  // which is why the refill (tile kt+3 -> stage of tile kt-1) is issued in L(kt), after G1's L(kt-1) completed.
  int cur = 0;
  auto compute = [&]() {
#if QH_N8_ABLATE == 2
    return;
#endif
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (KIND == K_I8) {
          acc[j][i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, wq[j]), __builtin_bit_cast(i32x4, xa[i]),
                                                             acc[j][i], 0, 0, 0);
        } else {
          const long wlo = (long)(((unsigned long)wq[j].y << 32) | wq[j].x), whi = (long)(((unsigned long)wq[j].w << 32) | wq[j].z);
          const long xlo = (long)(((unsigned long)xa[i].y << 32) | xa[i].x), xhi = (long)(((unsigned long)xa[i].w << 32) | xa[i].z);
          if constexpr (KIND == K_F8E4M3) {
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wlo, xlo, acc[j][i], 0, 0, 0);
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(whi, xhi, acc[j][i], 0, 0, 0);
          } else {
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(wlo, xlo, acc[j][i], 0, 0, 0);
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(whi, xhi, acc[j][i], 0, 0, 0);
          }
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // steady state: tiles kt+1 .. kt+3 exist; the own share of tile kt+1 has landed once at most 8 DMAs are in flight
  int kt = 0;
  for (; kt + 3 < nk; ++kt) {
#if QH_N8_ABLATE != 3
    read_frags(smem + cur * STAGE_BYTES);
#endif
#if QH_N8_ABLATE != 1
    issue(kt + 3, (cur + 3) & 3);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
    end_load_phase();
    compute();
    end_compute_phase();
    cur = (cur + 1) & 3;
  }
  // drain: 2, 1, 0 younger tiles in flight
  for (; kt < nk; ++kt) {
    read_frags(smem + cur * STAGE_BYTES);
    const int younger = nk - 2 - kt;
    if (younger >= 1)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    end_load_phase();
    compute();
    end_compute_phase();
    cur = (cur + 1) & 3;
  }
  if (grp == 0) end_compute_phase();

  // ---- epilogue: (int32 | fp32) accumulator * scale[n] (+ bias), parked per wave in LDS, stored as full 128-byte lines ----
  T* yg = reinterpret_cast<T*>(a.y);
  const bool has_bias = a.bias != nullptr;
  const bool full = (m0 + BM <= M) && (n0 + BN <= N) && (N % 8 == 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  constexpr int ROWB = 128;                    // bytes per parked row: 64 features (16-bit) or 32 features (fp32, two passes)
  constexpr int PASSES = sizeof(T) / 2;        // 1 or 2
  constexpr int JP = 4 / PASSES;               // feature fragments per pass
  uint8_t* park = smem + wave * (128 * ROWB);  // 16 KiB per wave
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
#pragma unroll
    for (int jj = 0; jj < JP; ++jj) {
      const int j = p * JP + jj;
      const int nb = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
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
          float v = (float)acc[j][i][r] * sc[r];  // library/qbytes_mm.py:47-49: fp32(int32) * fp32(scale), rounded to fp32 ...
          asm volatile("" : "+v"(v));             // ... and only then to the output dtype (no single-rounding v_fma_mixlo_f16)
          if (has_bias) v = E::to_f32(E::from_f32(v)) + bv[r];
          out[r] = E::from_f32(v);
        }
        const int row = i * 16 + (lane & 15);
        if constexpr (sizeof(T) == 2) {
          const int chunk = (jj * 4 + (lane >> 4)) ^ ((row & 7) << 1);  // 8-byte chunks
          *reinterpret_cast<uint2*>(park + row * ROWB + chunk * 8) = *reinterpret_cast<const uint2*>(out);
        } else {
          const int chunk = (jj * 4 + (lane >> 4)) ^ (row & 7);  // 16-byte chunks
          *reinterpret_cast<uint4*>(park + row * ROWB + chunk * 16) = *reinterpret_cast<const uint4*>(out);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int row = t * 8 + (lane >> 3);
      const int c16 = lane & 7;
      uint4 v;
      if constexpr (sizeof(T) == 2)
        v = *reinterpret_cast<const uint4*>(park + row * ROWB + (((c16 * 2) ^ ((row & 7) << 1)) * 8));
      else
        v = *reinterpret_cast<const uint4*>(park + row * ROWB + ((c16 ^ (row & 7)) * 16));
      const int m = m0 + wm * 128 + row;
      const int n = n0 + wn * 64 + p * (64 / PASSES) + c16 * (16 / (int)sizeof(T));
      if (full) {
        *reinterpret_cast<uint4*>(yg + (size_t)m * N + n) = v;
      } else if (m < M) {
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int r = 0; r < 16 / (int)sizeof(T); ++r)
          if (n + r < N) yg[(size_t)m * N + n + r] = e[r];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <int ODT, int KIND>
static int launch(const Args& a, hipStream_t stream) {
  constexpr int need = STAGES * STAGE_BYTES;  // 128 KiB; the epilogue parks 8 x 16 KiB in the same space
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbytes_native8_kernel<ODT, KIND>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, need);
  const int tiles = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM);
  hipLaunchKernelGGL((qbytes_native8_kernel<ODT, KIND>), dim3(tiles), dim3(512), need, stream, a);
  return launch_status();
}

}  // namespace n8

bool qbytes_native8_supported(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  const bool pair = (a_dtype == QUANTO_HIP_I8 && b_dtype == QUANTO_HIP_I8) ||
                    (a_dtype == QUANTO_HIP_F8_E4M3FN && b_dtype == QUANTO_HIP_F8_E4M3FN) ||
                    (a_dtype == QUANTO_HIP_F8_E5M2 && b_dtype == QUANTO_HIP_F8_E5M2);
  const bool od = out_dtype == QUANTO_HIP_BF16 || out_dtype == QUANTO_HIP_F16 || out_dtype == QUANTO_HIP_F32;
  return pair && od && K % n8::BK == 0 && K >= n8::BK && M >= 1 && M * K < (1ll << 31) && N * K < (1ll << 31) && M < (1 << 30) &&
         N < (1 << 30);
}

int qbytes_mm_native8(const void* a, const void* b, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int a_dtype,
                      int b_dtype, int out_dtype, hipStream_t stream) {
  if (!qbytes_native8_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16) return QUANTO_HIP_EALIGN;
  n8::Args args{reinterpret_cast<const uint8_t*>(a), reinterpret_cast<const uint8_t*>(b), s, bias, y, (int)M, (int)N, (int)K};
#define QH_KIND(ODT)                                                                  \
  if (a_dtype == QUANTO_HIP_I8) return n8::launch<ODT, n8::K_I8>(args, stream);       \
  if (a_dtype == QUANTO_HIP_F8_E4M3FN) return n8::launch<ODT, n8::K_F8E4M3>(args, stream); \
  return n8::launch<ODT, n8::K_F8E5M2>(args, stream)
  if (out_dtype == QUANTO_HIP_BF16) { QH_KIND(QUANTO_HIP_BF16); }
  if (out_dtype == QUANTO_HIP_F16) { QH_KIND(QUANTO_HIP_F16); }
  QH_KIND(QUANTO_HIP_F32);
#undef QH_KIND
}

}  // namespace qh
