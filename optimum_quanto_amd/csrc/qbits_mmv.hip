// qbits_mm for small decode batches (4 < M <= 32), int4, group size 128: the GEMV's structure with the products on the
// matrix cores.
//
// Why (r2, scripts/skinny_timeline.py on the streaming kernel of qbits_skinny.hip at (32,4096,4096), 10.9 us): 1.9 us until the
// first bytes arrive, 8 k-tile steps of 0.5 us (LDS-DMA issue, barrier, LDS fragment reads, MFMAs and fold, one after the
// other in the single wave each SIMD has), and 3.3 us of split-K tail (partials written through to memory, an arrival
// counter, the last block's reads): three fabric round trips that the M = 1 GEMV (4.4 us for the same weights) does not
// have, because it splits K over the waves of a block, not over blocks.  This kernel does the same for M up to 32:
//   * a block of 4 waves owns 16 FG output features (FG x 8 packed rows, both nibble planes) over the WHOLE K; wave w takes
//     the k-tiles (128 k = one group) kt = w, w + 4, ...; nothing is shared between the waves but the scale/shift table, so
//     the main loop has no barrier and no LDS traffic except two table reads per tile;
//   * weights AND activations go from global memory straight into the MFMA operand registers (the B fragment of
//     v_mfma_f32_16x16x32 is 16 contiguous bytes of one token's row): a ring of D tiles of asm loads per wave, in-order
//     vmcnt arithmetic (one wait per tile), wave-uniform SGPR base + 32-bit lane offset addressing;
//   * 128+q operands, one accumulator per group, fp32 fold acc += s acc_g - (z + 128 s) XS with XS from the ones-MFMA:
//     the arithmetic of the streaming kernel, exact-math oracle;
//   * the four waves' sums are added through LDS in wave order; no workspace, deterministic.
// Cost: every block reads all of x (M x K x 2 bytes, L2 hits) for 16 FG features - the reason this is for M <= 32 only and
// why FG grows with N.
#include "qh_common.h"

namespace qh {
namespace mmv {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <int DT>
struct Mma;
template <>
struct Mma<QUANTO_HIP_BF16> {
  using V8 = bf16x8;
  static __device__ __forceinline__ f32x4 run(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static constexpr uint32_t MAGIC = 0x43004300u, ONE2 = 0x3F803F80u;
  static constexpr float OFFSET = 128.f;
};
template <>
struct Mma<QUANTO_HIP_F16> {
  using V8 = f16x8;
  static __device__ __forceinline__ f32x4 run(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static constexpr uint32_t MAGIC = 0x64006400u, ONE2 = 0x3C003C00u;
  static constexpr float OFFSET = 1024.f;
};

// s_waitcnt vmcnt(n * PER), n = 0 .. MAXN / PER (the immediate must be a literal)
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

// Asm loads: the result register is written when the data ARRIVES - every target is "touched" behind the covering
// s_waitcnt before its first use and no load is issued whose result is not consumed (qbits_mfma_fused.hip, lesson 1).
template <int OFF, bool NT, typename R>
__device__ __forceinline__ void gload16(R& dst, uint32_t voff, const void* sbase) {
  if constexpr (NT)
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
  else
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}

struct Args {
  const void* x;      // [M, K]
  const uint8_t* w;   // packed [N/2, K]
  const void* scale;  // [N, G]
  const void* shift;  // [N, G]
  const void* bias;   // [N] or null
  void* y;            // [M, N]
  int M, N, K, G;
};

template <int DT, int TF, int FG, int D, bool INT_SHIFT, int NW>
__global__ void __launch_bounds__(NW * 64) qbits_mmv_kernel(const Args a) {
  using E = Elem<DT>;
  using T = typename E::T;
  using V8 = typename Mma<DT>::V8;
  constexpr int L = FG * 2 + TF * 4;  // loads per tile and lane
  constexpr int NF = 16 * FG;         // features per block
  constexpr int NFP = NF + 4;         // table row pitch: 8 bytes of padding against bank conflicts in the fill (qbits_skinny.hip)
  static_assert((D - 1) * L <= 63, "vmcnt immediate");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // [table: G x {scale, shift} x NF of T] then the cross-wave reduction buffer (re-uses the table's space)
  T* sz = reinterpret_cast<T*>(smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M, N = a.N, K = a.K, G = a.G;
  const int P = N >> 1;
  const int p0 = blockIdx.x * (8 * FG);  // first packed row of the block
  const int fi = lane & 15, fg = lane >> 4;
  const int J = (G - wave + NW - 1) / NW;  // k-tiles of this wave: kt = wave + NW j

  // ---- lane offsets (bytes) from the wave-uniform bases ---------------------------------------------------------------
  // weights: feature group q, MFMA row fi -> packed row fi & 7; chunk fg (k-steps 0,1) and 4 + fg (k-steps 2,3) of the tile
  uint32_t wlane[FG];
#pragma unroll
  for (int q = 0; q < FG; ++q) wlane[q] = (uint32_t)(p0 + q * 8 + (fi & 7)) * (uint32_t)K + fg * 16;
  const uint32_t nib_shift = (fi >> 3) * 4;
  // activations: token tf*16 + fi (clamped); k-step t of lane group fg multiplies the weight bytes 8 (t & 1) .. +7 of chunk
  // 4 (t >> 1) + fg, i.e. k = 64 (t >> 1) + 16 fg + 8 (t & 1): 16 bytes at 128 (t >> 1) + 32 fg + 16 (t & 1) of the token's tile
  uint32_t xlane[TF];
#pragma unroll
  for (int tf = 0; tf < TF; ++tf) {
    const int m = tf * 16 + fi;
    xlane[tf] = (uint32_t)(m < M ? m : M - 1) * (uint32_t)K * 2 + fg * 32;
  }
  const uint8_t* wbase = a.w + (size_t)wave * 128;  // (wave < NW)
  const uint8_t* xbase = reinterpret_cast<const uint8_t*>(a.x) + (size_t)wave * 256;

  u32x4_t wr[D][FG][2];
  V8 xr[D][TF][4];
  auto issue_tile = [&](int j, u32x4_t (&w)[FG][2], V8 (&x)[TF][4]) {
    const uint8_t* wb = wbase + (size_t)j * (128 * NW);  // tile kt = wave + NW j starts at byte 128 kt of a packed row
    const uint8_t* xb = xbase + (size_t)j * (256 * NW);  // and at byte 256 kt of an activation row
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) {
      gload16<0, false>(x[tf][0], xlane[tf], xb);
      gload16<16, false>(x[tf][1], xlane[tf], xb);
      gload16<128, false>(x[tf][2], xlane[tf], xb);
      gload16<144, false>(x[tf][3], xlane[tf], xb);
    }
#pragma unroll
    for (int q = 0; q < FG; ++q) {
      gload16<0, true>(w[q][0], wlane[q], wb);
      gload16<64, true>(w[q][1], wlane[q], wb);
    }
  };
  auto touch_tile = [&](u32x4_t (&w)[FG][2], V8 (&x)[TF][4]) {
#pragma unroll
    for (int tf = 0; tf < TF; ++tf)
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(x[tf][t]));
#pragma unroll
    for (int q = 0; q < FG; ++q) {
      asm volatile("" : "+v"(w[q][0]));
      asm volatile("" : "+v"(w[q][1]));
    }
  };

  // ---- fill the ring, then the scale / shift table -------------------------------------------------------------------------
  // sz[(g * 2 + which) * NFP + f], f = q*16 + plane*8 + row: the lane's four consecutive features are 8 contiguous bytes.
  // The table loads are hipcc's: it waits for them with vmcnt(0), i.e. for the whole ring issued before them as well - one
  // first-byte latency for both (the other order would pay it twice).
#pragma unroll
  for (int u = 0; u < D; ++u)
    if (u < J) issue_tile(u, wr[u], xr[u]);

  {
    const int total = NF * G;
    for (int e = tid; e < total; e += NW * 64) {
      const int f = e / G, g = e - f * G;
      const int q = f >> 4, plane = (f >> 3) & 1, row = f & 7;
      const size_t idx = (size_t)(p0 + q * 8 + row + plane * P) * G + g;
      sz[(g * 2 + 0) * NFP + f] = reinterpret_cast<const T*>(a.scale)[idx];
      if constexpr (INT_SHIFT)
        sz[(g * 2 + 1) * NFP + f] = E::from_f32((float)(int8_t) reinterpret_cast<const uint8_t*>(a.shift)[idx]);
      else
        sz[(g * 2 + 1) * NFP + f] = reinterpret_cast<const T*>(a.shift)[idx];
    }
  }
  __syncthreads();

  f32x4 acc[FG][TF];
#pragma unroll
  for (int q = 0; q < FG; ++q)
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) acc[q][tf] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint32_t kmask = 0x000F000Fu, kmagic = Mma<DT>::MAGIC;
  asm volatile("" : "+s"(kmask));
  asm volatile("" : "+v"(kmagic));
  const V8 ones = __builtin_bit_cast(V8, make_uint4(Mma<DT>::ONE2, Mma<DT>::ONE2, Mma<DT>::ONE2, Mma<DT>::ONE2));
  const int floc = (fg >> 1) * 8 + 4 * (fg & 1);  // first of the lane's four features inside a group of 16

  auto compute_tile = [&](int kt, const u32x4_t (&w)[FG][2], const V8 (&x)[TF][4]) {
    f32x4 accg[FG][TF], accx[TF];
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) {
      accx[tf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < FG; ++q) accg[q][tf] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int tf = 0; tf < TF; ++tf) accx[tf] = Mma<DT>::run(ones, x[tf][t], accx[tf]);
#pragma unroll
      for (int q = 0; q < FG; ++q) {
        // k-step t uses bytes 8 (t & 1) .. +7 of chunk t >> 1: two dwords -> four operand dwords (natural k order)
        const uint32_t d0 = (t & 1) ? w[q][t >> 1].z : w[q][t >> 1].x, d1 = (t & 1) ? w[q][t >> 1].w : w[q][t >> 1].y;
        const uint32_t s0 = d0 >> nib_shift, s1 = d1 >> nib_shift;
        uint32_t op[4];
        op[0] = (__builtin_amdgcn_perm(0u, s0, 0x0C010C00u) & kmask) | kmagic;
        op[1] = (__builtin_amdgcn_perm(0u, s0, 0x0C030C02u) & kmask) | kmagic;
        op[2] = (__builtin_amdgcn_perm(0u, s1, 0x0C010C00u) & kmask) | kmagic;
        op[3] = (__builtin_amdgcn_perm(0u, s1, 0x0C030C02u) & kmask) | kmagic;
        const V8 wa = __builtin_bit_cast(V8, make_uint4(op[0], op[1], op[2], op[3]));
#pragma unroll
        for (int tf = 0; tf < TF; ++tf) accg[q][tf] = Mma<DT>::run(wa, x[tf][t], accg[q][tf]);
      }
    }
    // fold the group: acc += s * acc_g - (z + OFFSET s) * XS
#pragma unroll
    for (int q = 0; q < FG; ++q) {
      T s4t[4], z4t[4];
      *reinterpret_cast<uint2*>(s4t) = *reinterpret_cast<const uint2*>(sz + (kt * 2 + 0) * NFP + q * 16 + floc);
      *reinterpret_cast<uint2*>(z4t) = *reinterpret_cast<const uint2*>(sz + (kt * 2 + 1) * NFP + q * 16 + floc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = E::to_f32(s4t[r]);
        const float z = E::to_f32(z4t[r]);
        const float zz = INT_SHIFT ? s * (z + Mma<DT>::OFFSET) : z + Mma<DT>::OFFSET * s;
#pragma unroll
        for (int tf = 0; tf < TF; ++tf) acc[q][tf][r] += s * accg[q][tf][r] - zz * accx[tf][0];
      }
    }
  };

  // ---- main loop: tile j lives in ring slot j % D -----------------------------------------------------------------------------
  for (int j0 = 0; j0 < J; j0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int j = j0 + u;
      if (j < J) {
        const int younger = J - 1 - j < D - 1 ? J - 1 - j : D - 1;  // tiles issued after tile j
        wait_vmcnt<(D - 1) * L, L>(younger);
        touch_tile(wr[u], xr[u]);
        compute_tile(wave + NW * j, wr[u], xr[u]);
        if (j + D < J) issue_tile(j + D, wr[u], xr[u]);
      }
    }
  }

  // ---- add the four waves' sums in wave order, wave 0 writes ---------------------------------------------------------------
  __syncthreads();  // everybody is done with the table
  f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
  for (int q = 0; q < FG; ++q)
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) red[((q * TF + tf) * NW + wave) * 64 + lane] = acc[q][tf];
  __syncthreads();
  if (wave != 0) return;
  T* yg = reinterpret_cast<T*>(a.y);
  const bool has_bias = a.bias != nullptr;
#pragma unroll
  for (int q = 0; q < FG; ++q) {
    const int n0 = p0 + q * 8 + 4 * (fg & 1) + (fg >> 1) * P;  // 4 consecutive output features
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (has_bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = E::to_f32(reinterpret_cast<const T*>(a.bias)[n0 + r]);
    }
#pragma unroll
    for (int tf = 0; tf < TF; ++tf) {
      f32x4 v = red[((q * TF + tf) * NW + 0) * 64 + lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const f32x4 o = red[((q * TF + tf) * NW + w) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += o[r];
      }
      const int m = tf * 16 + fi;
      if (m < M) {
        T out[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float f = v[r];
          if (has_bias) f = E::to_f32(E::from_f32(f)) + bv[r];  // the reference rounds the product, then adds the bias
          out[r] = E::from_f32(f);
        }
        *reinterpret_cast<uint2*>(yg + (size_t)m * N + n0) = *reinterpret_cast<const uint2*>(out);
      }
    }
  }
}

constexpr int lds_bytes(int tf, int fg, int G, int nw = 8) {
  const int table = G * 2 * (16 * fg + 4) * 2, red = fg * tf * nw * 64 * 16;
  return table > red ? table : red;
}

template <int DT, int TF, int FG, int D, bool INT_SHIFT, int NW = 4>
static int launch(const Args& a, hipStream_t stream) {
  const int lds = lds_bytes(TF, FG, a.G, NW);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qbits_mmv_kernel<DT, TF, FG, D, INT_SHIFT, NW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((qbits_mmv_kernel<DT, TF, FG, D, INT_SHIFT, NW>), dim3(a.N / (16 * FG)), dim3(NW * 64), lds, stream, a);
  return launch_status();
}

// feature groups per block: one until the grid would exceed ~two blocks per CU (every block re-reads all of x)
inline int pick_fg(int64_t N) {
  const int forced = env_int("QUANTO_HIP_MMV_FG", 0);
  if ((forced == 1 || forced == 2) && N % (16 * forced) == 0) return forced;
  return (N % 32 == 0 && N / 16 > 512) ? 2 : 1;
}

template <int DT, bool INT_SHIFT>
static int launch_shape(const Args& a, hipStream_t stream) {
  const int fg = pick_fg(a.N);
  if (a.M <= 16) {
    if (fg == 2) return launch<DT, 1, 2, 4, INT_SHIFT>(a, stream);
    // (8,4096,4096), us: ring of 4 tiles, 4 waves 7.67; ring of 8 (everything in flight at once) 9.24 - the table wait then
    // covers the whole stream; 8 waves per block 7.98
    return launch<DT, 1, 1, 4, INT_SHIFT>(a, stream);
  }
  if (fg == 2) return launch<DT, 2, 2, 4, INT_SHIFT>(a, stream);
  return launch<DT, 2, 1, 4, INT_SHIFT>(a, stream);
}

}  // namespace mmv

bool qbits_mmv_supported(int64_t M, const PackedGeom& g, int dtype) {
  return g.bits == 4 && g.C == 128 && g.N % 16 == 0 && g.K % 128 == 0 && M >= 1 && M <= 32 &&
         (dtype == QUANTO_HIP_BF16 || dtype == QUANTO_HIP_F16) && (g.N / 2) * g.K < (int64_t)1 << 31 && 32 * g.K * 2 < (int64_t)1 << 31 &&
         mmv::lds_bytes(2, 2, (int)g.G) <= 160 * 1024;
}

int qbits_mm_mmv(const void* x, const uint8_t* packed, const void* scale, const void* shift, const void* bias, void* y, int64_t M,
                 const PackedGeom& g, int dtype, bool int_shift, hipStream_t stream) {
  if (!qbits_mmv_supported(M, g, dtype)) return QUANTO_HIP_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed)) % 16) return QUANTO_HIP_EALIGN;
  mmv::Args a{x, packed, scale, shift, bias, y, (int)M, (int)g.N, (int)g.K, (int)g.G};
  if (dtype == QUANTO_HIP_BF16)
    return int_shift ? mmv::launch_shape<QUANTO_HIP_BF16, true>(a, stream) : mmv::launch_shape<QUANTO_HIP_BF16, false>(a, stream);
  return int_shift ? mmv::launch_shape<QUANTO_HIP_F16, true>(a, stream) : mmv::launch_shape<QUANTO_HIP_F16, false>(a, stream);
}

}  // namespace qh
