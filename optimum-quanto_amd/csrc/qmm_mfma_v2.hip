// qbytes_mm MFMA GEMM, v2: 256x256x64 tile, 8 waves, LDS-DMA staging, in-register weight conversion.
//
// y[M,N] = (x[M,K] @ q[N,K]^T) * scale[N]   with x bf16/fp16 and q int8 / fp8 (1 byte per weight).
//
// Why this shape (MI355X: 256 CUs, 160 KiB LDS, MFMA 16x16x32 = 16 cycles per SIMD):
//   * both operands are K-contiguous, so both are staged by `global_load_lds` (16 B per lane, no VGPR round trip, no
//     ds_write pass).  The weight tile stays in its 1-byte storage format in LDS: half the LDS footprint and half the
//     ds_read traffic of a bf16 image, and the stage is a pure DMA pipeline (3 stages of 48 KiB in flight, counted
//     vmcnt, one raw s_barrier per K-tile, loads always one to two tiles ahead, never drained inside the loop);
//   * each wave converts the weight fragments it is about to use in registers (int8: v_cvt_f32_i32 with SDWA byte
//     select + v_cvt_pk_bf16_f32; fp8: v_cvt_pk_f32_fp8 + pack) - 1.5 VALU ops per weight next to 64 MFMAs per
//     K-tile and wave, exact because every int8 / fp8 value is representable in bf16 and fp16;
//   * the LDS image is linear per DMA instruction (wave-uniform base + lane*16), so the bank-conflict swizzle is
//     applied to the per-lane GLOBAL source address and undone on the fragment read (both are the same involution):
//       activations: 128-byte rows, 16-byte chunk c of row r lives at chunk position c ^ (r & 7);
//       weights:      64-byte rows, chunk c of row r lives at position c ^ ((-(r >> 2)) & 3);
//   * a weight lane reads ONE ds_read_b128 per fragment and K-tile (bytes k = 16g .. 16g+15 of its row, g = lane>>4)
//     and uses the low half for the first MFMA k-step and the high half for the second; the activation fragments are
//     read with the matching k assignment (chunk 2g + kk) - MFMA results are invariant under a k permutation applied
//     to both operands.
#include "qh_common.h"

namespace qh {

namespace v2 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int A_BYTES = BM * BK * 2;   // 32 KiB
constexpr int W_BYTES = BN * BK;       // 16 KiB
constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
constexpr int STAGES = 3;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// LDS-DMA of 16 bytes per lane: LDS[m0 + lane*16 .. +16) <- *gsrc.  Issued through inline asm on purpose: hipcc
// (ROCm 7.2) tracks the builtin form as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front of the next ds_read,
// which drains the whole prefetch pipeline every K-tile.  The asm form is invisible to that bookkeeping, so the
// counted `s_waitcnt vmcnt(N)` + `s_barrier` below are the ONLY ordering between the DMA and the fragment reads.
// `lds_dst` must be wave-uniform (it is moved to M0; M0 is saved/restored because the compiler owns it).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

template <int DT>
struct Mma;
template <>
struct Mma<QUANTO_HIP_BF16> {
  using V8 = bf16x8;
  static __device__ __forceinline__ f32x4 run(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    bf16x2 r;
    r.x = (__bf16)a;
    r.y = (__bf16)b;
    return __builtin_bit_cast(uint32_t, r);
  }
};
template <>
struct Mma<QUANTO_HIP_F16> {
  using V8 = f16x8;
  static __device__ __forceinline__ f32x4 run(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));  // exact for int8 / fp8 values
  }
};

enum { W_I8 = 0, W_F8E4M3 = 1, W_F8E5M2 = 2 };

// 8 one-byte weights (two dwords) -> one MFMA operand (8 x 16-bit)
template <int DT, int FMT>
__device__ __forceinline__ typename Mma<DT>::V8 convert8(uint32_t w0, uint32_t w1) {
  uint32_t out[4];
  const uint32_t in[2] = {w0, w1};
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    float f0, f1, f2, f3;
    if constexpr (FMT == W_I8) {
      f0 = (float)(int8_t)(in[d] & 0xFFu);
      f1 = (float)(int8_t)((in[d] >> 8) & 0xFFu);
      f2 = (float)(int8_t)((in[d] >> 16) & 0xFFu);
      f3 = (float)(int8_t)(in[d] >> 24);
    } else if constexpr (FMT == W_F8E4M3) {
      const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)in[d], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)in[d], true);
      f0 = lo.x; f1 = lo.y; f2 = hi.x; f3 = hi.y;
    } else {
      const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_bf8((int)in[d], false), hi = __builtin_amdgcn_cvt_pk_f32_bf8((int)in[d], true);
      f0 = lo.x; f1 = lo.y; f2 = hi.x; f3 = hi.y;
    }
    out[2 * d] = Mma<DT>::pack(f0, f1);
    out[2 * d + 1] = Mma<DT>::pack(f2, f3);
  }
  const uint4 v = make_uint4(out[0], out[1], out[2], out[3]);
  return __builtin_bit_cast(typename Mma<DT>::V8, v);
}

struct Args {
  const void* x;
  const uint8_t* w;
  const void* scale;
  const void* bias;
  void* y;
  int M, N, K;
};

template <int DT, int FMT>
__global__ void __launch_bounds__(512, 1) qbytes_mfma_v2_kernel(const Args a) {
  using E = Elem<DT>;
  using T = typename E::T;
  using V8 = typename Mma<DT>::V8;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int M = a.M, N = a.N, K = a.K;
  const int nk = K / BK;

  // XCD-aware tile order: consecutive block ids land on different XCDs (block b -> XCD b % 8); give each XCD a
  // contiguous band of tiles so that neighbouring tiles (which share an activation or weight panel) share an L2.
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  const int nwg = tiles_n * tiles_m;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane DMA source pointers (advance by BK per tile) ----------------------------------------
  const T* xg = reinterpret_cast<const T*>(a.x);
  const uint8_t* asrc[4];
  const uint8_t* wsrc[2];
  int adst[4], wdst[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int R = (j * 8 + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (R & 7);
    int m = m0 + R;
    m = m < M ? m : M - 1;
    asrc[j] = reinterpret_cast<const uint8_t*>(xg + (size_t)m * K + c * 8);
    adst[j] = (j * 8 + wave) * 1024;  // wave-uniform: the DMA adds lane*16
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int R = (j * 8 + wave) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((-(R >> 2)) & 3);
    int n = n0 + R;
    n = n < N ? n : N - 1;
    wsrc[j] = a.w + (size_t)n * K + c * 16;
    wdst[j] = A_BYTES + (j * 8 + wave) * 1024;
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  auto issue = [&](int kt) {
    const uint32_t st = __builtin_amdgcn_readfirstlane(lds_base + (kt % STAGES) * STAGE_BYTES);
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(asrc[j] + (size_t)kt * (BK * 2), st + adst[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(wsrc[j] + (size_t)kt * BK, st + wdst[j]);
  };

  // ---- fragment read offsets (constant per lane) ------------------------------------------------------
  int aoff[8][2], boff[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int R = wm * 128 + i * 16 + (lane & 15);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) aoff[i][kk] = R * 128 + ((((lane >> 4) * 2 + kk) ^ (R & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int R = wn * 64 + j * 16 + (lane & 15);
    boff[j] = A_BYTES + R * 64 + (((lane >> 4) ^ ((-(R >> 2)) & 3)) << 4);
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0);
  if (nk > 1) issue(1);

  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most the 6 DMA instructions of tile kt+1 are still outstanding
    if (kt + 1 < nk)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's part of tile kt is visible; every wave is done reading stage (kt+2)%3
    asm volatile("" ::: "memory");
    if (kt + 2 < nk) issue(kt + 2);

    const uint8_t* st = smem + (kt % STAGES) * STAGE_BYTES;
    uint4 braw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) braw[j] = *reinterpret_cast<const uint4*>(st + boff[j]);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      V8 fa[8], fb[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const V8*>(st + aoff[i][kk]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        fb[j] = kk == 0 ? convert8<DT, FMT>(braw[j].x, braw[j].y) : convert8<DT, FMT>(braw[j].z, braw[j].w);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Mma<DT>::run(fa[i], fb[j], acc[i][j]);
    }
  }

  // ---- epilogue: per-channel scale on the fp32 accumulator, optional bias, store -------------------------
  T* yg = reinterpret_cast<T*>(a.y);
  const bool has_bias = a.bias != nullptr;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + (lane & 15);
    if (n >= N) continue;
    const float sc = E::to_f32(reinterpret_cast<const T*>(a.scale)[n]);
    const float bv = has_bias ? E::to_f32(reinterpret_cast<const T*>(a.bias)[n]) : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 128 + i * 16 + (lane >> 4) * 4 + r;
        if (m < M) {
          float v = acc[i][j][r] * sc;
          if (has_bias) v = E::to_f32(E::from_f32(v)) + bv;
          yg[(size_t)m * N + n] = E::from_f32(v);
        }
      }
    }
  }
}

template <int DT, int FMT>
static int launch(const Args& a, hipStream_t stream) {
  static bool attr_done = false;
  constexpr int lds = STAGES * STAGE_BYTES;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbytes_mfma_v2_kernel<DT, FMT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done = true;
  }
  const int tiles = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM);
  hipLaunchKernelGGL((qbytes_mfma_v2_kernel<DT, FMT>), dim3(tiles), dim3(512), lds, stream, a);
  return launch_status();
}

}  // namespace v2

// Large-tile kernel: worth it when the grid fills the chip with 256x256 tiles.
bool qbytes_mfma_v2_supported(int64_t M, int64_t N, int64_t K, int a_dtype, int b_dtype, int out_dtype) {
  const bool bd = b_dtype == QUANTO_HIP_I8 || b_dtype == QUANTO_HIP_F8_E4M3FN || b_dtype == QUANTO_HIP_F8_E5M2;
  return bd && a_dtype == out_dtype && (out_dtype == QUANTO_HIP_BF16 || out_dtype == QUANTO_HIP_F16) && K % v2::BK == 0 &&
         K >= 2 * v2::BK && M >= 1 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30);
}

int qbytes_mm_mfma_v2(const void* x, const void* w, const void* s, const void* bias, void* y, int64_t M, int64_t N, int64_t K, int a_dtype,
                      int b_dtype, int out_dtype, hipStream_t stream) {
  if (!qbytes_mfma_v2_supported(M, N, K, a_dtype, b_dtype, out_dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16) return QUANTO_HIP_EALIGN;
  v2::Args a{x, reinterpret_cast<const uint8_t*>(w), s, bias, y, (int)M, (int)N, (int)K};
#define QH_CASE(DT, FMT) return v2::launch<DT, FMT>(a, stream)
  if (out_dtype == QUANTO_HIP_BF16) {
    if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_BF16, v2::W_I8);
    if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_BF16, v2::W_F8E4M3);
    QH_CASE(QUANTO_HIP_BF16, v2::W_F8E5M2);
  }
  if (b_dtype == QUANTO_HIP_I8) QH_CASE(QUANTO_HIP_F16, v2::W_I8);
  if (b_dtype == QUANTO_HIP_F8_E4M3FN) QH_CASE(QUANTO_HIP_F16, v2::W_F8E4M3);
  QH_CASE(QUANTO_HIP_F16, v2::W_F8E5M2);
#undef QH_CASE
}

}  // namespace qh
