// Stand-alone probe for profiles/r04_packed_fp32_next_to_mfma.md: does `v_pk_add_f32 ... op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]` (low result
// = src0.lo - src1.HI) return a stale src1.HI in lanes 48-63 when that register was written by the VALU instruction(s) just before, in the
// shadow of a v_mfma_f32_16x16x32_bf16, with two waves per SIMD?  Everything that matters is inline asm on FIXED registers, so the
// instruction stream is exactly what is written here; the MFMA's registers are disjoint from the VALU chain (no software-managed MFMA
// hazard is involved: a VALU -> VALU read-after-write is interlocked by the hardware).
//   hipcc --offload-arch=gfx950 -O2 -o pk_probe pk_f32_opsel_probe.hip && ./pk_probe [launches]
// Variant v = distance (0,1,2,4,8 independent VALU instructions) between the write of the HIGH half and the packed add, x with / without
// the MFMA in front, x op_sel on the high / low half.  Each thread checks its own results against scalar arithmetic done with v_sub_f32
// and counts mismatches per quarter wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define FILL0 ""
#define FILL1 "v_add_u32 v90, v90, v91\n\t"
#define FILL2 FILL1 "v_xor_b32 v92, v92, v91\n\t"
#define FILL4 FILL2 "v_add_u32 v93, v93, v91\n\t" "v_xor_b32 v94, v94, v91\n\t"
#define FILL8 FILL4 FILL4

// one trial: v2 / v3 = two "weights" (floats built by shifts, as bf16 -> f32), v72 / v73 = (other plane's shift, this plane's shift),
// v73 written LAST by a shift, then DIST filler VALUs, then the packed add; the stale candidate left in v73 before is 0x7fc00000 (NaN).
#define TRIAL(MFMA, DIST, OPSEL) TRIAL_I(MFMA, DIST, "v_pk_add_f32 v[82:83], v[2:3], v[72:73] " OPSEL " neg_lo:[0,1] neg_hi:[0,1]\n\t")
#define TRIAL_I(MFMA, DIST, INSTR)                                                     \
  asm volatile(                                                                         \
      "v_mov_b32 v73, 0x7fc00000\n\t"                                                 \
      "v_mov_b32 v72, 0x7fc00000\n\t"                                                 \
      "s_nop 4\n\t" MFMA                                                               \
      "v_lshlrev_b32 v2, 16, %2\n\t"                                                   \
      "v_lshlrev_b32 v3, 16, %3\n\t"                                                   \
      "v_and_b32 v95, 0xffff0000, %4\n\t"                                              \
      "v_lshlrev_b32 v72, 16, %4\n\t"                                                 \
      "v_mov_b32 v73, v95\n\t" DIST INSTR                                             \
      "s_nop 4\n\t"                                                                     \
      "v_mov_b32 %0, v82\n\t"                                                          \
      "v_mov_b32 %1, v83\n\t"                                                          \
      : "=v"(lo), "=v"(hi)                                                              \
      : "v"(w0), "v"(w1), "v"(sh)                                                       \
      : "v2", "v3", "v82", "v83", "v90", "v91", "v92", "v93", "v94", "v95", "v72", "v73", "a0", "a1", "a2", "a3", "v100", "v101", "v102", \
        "v103", "v104", "v105", "v106", "v107", "memory")

#define MFMA_ON "v_mfma_f32_16x16x32_bf16 a[0:3], v[100:103], v[104:107], a[0:3]\n\t"

template <int V>
__device__ __forceinline__ void trial(uint32_t w0, uint32_t w1, uint32_t sh, float& lo, float& hi) {
  if constexpr (V == 0) TRIAL(MFMA_ON, FILL0, "op_sel:[0,1]");
  if constexpr (V == 1) TRIAL(MFMA_ON, FILL1, "op_sel:[0,1]");
  if constexpr (V == 2) TRIAL(MFMA_ON, FILL2, "op_sel:[0,1]");
  if constexpr (V == 3) TRIAL(MFMA_ON, FILL4, "op_sel:[0,1]");
  if constexpr (V == 4) TRIAL(MFMA_ON, FILL8, "op_sel:[0,1]");
  if constexpr (V == 5) TRIAL("", FILL0, "op_sel:[0,1]");
  if constexpr (V == 6) TRIAL("", FILL8, "op_sel:[0,1]");
  if constexpr (V == 7) TRIAL(MFMA_ON, FILL0, "op_sel_hi:[1,0]");
  if constexpr (V == 8) TRIAL(MFMA_ON, FILL8, "op_sel_hi:[1,0]");
  if constexpr (V == 9) TRIAL_I(MFMA_ON, FILL0, "v_pk_add_f32 v[82:83], v[2:3], v[72:73] op_sel:[0,1]\n\t");   // no neg modifiers: f + s_hi
  if constexpr (V == 10) TRIAL_I(MFMA_ON, FILL0, "v_pk_mul_f32 v[82:83], v[2:3], v[72:73] op_sel:[0,1]\n\t");  // f * s_hi
  if constexpr (V == 11) TRIAL_I(MFMA_ON, FILL0, "v_pk_add_f32 v[82:83], v[72:73], v[2:3] op_sel:[1,0] op_sel_hi:[1,1]\n\t");  // high half as src0: lo = s_hi + f0, hi = s_hi + f1
  if constexpr (V == 12) TRIAL_I("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" MFMA_ON "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t", FILL0,
                                 "v_pk_add_f32 v[82:83], v[2:3], v[72:73] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t");  // 32+ idle cycles around this wave's own MFMA
}

template <int V>
__global__ void __launch_bounds__(512, 2) probe(const uint32_t* __restrict__ in, unsigned long long* __restrict__ bad, unsigned* __restrict__ nex, float* __restrict__ ex, int iters) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  uint32_t s = in[tid];
  asm volatile("v_mov_b32 v100, %0\n\tv_mov_b32 v101, %0\n\tv_mov_b32 v102, %0\n\tv_mov_b32 v103, %0\n\tv_mov_b32 v104, %0\n\tv_mov_b32 v105, %0\n\t"
               "v_mov_b32 v106, %0\n\tv_mov_b32 v107, %0\n\tv_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\t"
               "v_accvgpr_write_b32 a3, 0\n\tv_mov_b32 v90, 0\n\tv_mov_b32 v91, 1\n\tv_mov_b32 v92, 0\n\tv_mov_b32 v93, 0\n\tv_mov_b32 v94, 0\n\ts_nop 4"
               :: "v"(0x3c003c00u ^ (s & 0x00ff00ffu))
               : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "a0", "a1", "a2", "a3", "v90", "v91", "v92", "v93", "v94");
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const uint32_t w0 = 0x3f80u + ((s >> 3) & 0x7f), w1 = 0x4000u + ((s >> 11) & 0x7f);  // bf16 bit patterns of small positive floats
    const uint32_t sh = (0x3e80u + ((s >> 19) & 0x7f)) << 16 | (0x3d00u + ((s >> 25) & 0x7f));  // high half: this plane's shift, low: the other
    float lo, hi;
    trial<V>(w0, w1, sh, lo, hi);
    const float f0 = __builtin_bit_cast(float, w0 << 16), f1 = __builtin_bit_cast(float, w1 << 16);
    const float s_hi = __builtin_bit_cast(float, sh & 0xffff0000u), s_lo = __builtin_bit_cast(float, sh << 16);
    // op_sel:[0,1]: both results subtract the HIGH half; op_sel_hi:[1,0]: both subtract the LOW half
    const float sub = (V == 7 || V == 8) ? s_lo : s_hi;
    float want_lo, want_hi;  // scalar arithmetic, kept scalar (hipcc would pack it into the very instruction under test)
    if constexpr (V == 9 || V == 11)
      asm volatile("s_nop 4\n\tv_add_f32 %0, %2, %4\n\tv_add_f32 %1, %3, %4\n\ts_nop 4" : "=&v"(want_lo), "=&v"(want_hi) : "v"(f0), "v"(f1), "v"(sub));
    else if constexpr (V == 10)
      asm volatile("s_nop 4\n\tv_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4\n\ts_nop 4" : "=&v"(want_lo), "=&v"(want_hi) : "v"(f0), "v"(f1), "v"(sub));
    else
      asm volatile("s_nop 4\n\tv_sub_f32 %0, %2, %4\n\tv_sub_f32 %1, %3, %4\n\ts_nop 4" : "=&v"(want_lo), "=&v"(want_hi) : "v"(f0), "v"(f1), "v"(sub));
    const bool blo = __builtin_bit_cast(uint32_t, lo) != __builtin_bit_cast(uint32_t, want_lo), bhi = __builtin_bit_cast(uint32_t, hi) != __builtin_bit_cast(uint32_t, want_hi);
    if (blo || bhi) {
      nbad += blo + bhi;
      const unsigned slot = atomicAdd(&nex[V], 1u);
      if (slot < 4) {  // a few examples per variant: which result, what came out, what should have, the operands
        float* e = ex + (V * 4 + slot) * 8;
        e[0] = blo ? 0.f : 1.f; e[1] = blo ? lo : hi; e[2] = blo ? want_lo : want_hi; e[3] = blo ? f0 : f1; e[4] = s_lo; e[5] = s_hi; e[6] = (float)(threadIdx.x & 63); e[7] = (float)it;
      }
    }
  }
  if (nbad) atomicAdd(&bad[V * 4 + ((threadIdx.x & 63) >> 4)], (unsigned long long)nbad);
}

template <int V>
void run(const uint32_t* in, unsigned long long* bad, unsigned* nex, float* ex, int launches, int iters) {
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(probe<V>, dim3(512), dim3(512), 0, 0, in, bad, nex, ex, iters);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 10000, iters = 64;
  const int n = 512 * 512;
  uint32_t* h = (uint32_t*)malloc(n * 4);
  for (int i = 0; i < n; ++i) h[i] = 2654435761u * (i + 1);
  uint32_t* in;
  unsigned long long* bad;
  unsigned* nex;
  float* ex;
  constexpr int NV = 13;
  hipMalloc(&in, n * 4);
  hipMalloc(&bad, NV * 4 * 8);
  hipMalloc(&nex, NV * 4);
  hipMalloc(&ex, NV * 4 * 8 * 4);
  hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
  hipMemset(bad, 0, NV * 4 * 8);
  hipMemset(nex, 0, NV * 4);
  hipMemset(ex, 0, NV * 4 * 8 * 4);
  run<0>(in, bad, nex, ex, launches, iters); run<1>(in, bad, nex, ex, launches, iters); run<2>(in, bad, nex, ex, launches, iters);
  run<3>(in, bad, nex, ex, launches, iters); run<4>(in, bad, nex, ex, launches, iters); run<5>(in, bad, nex, ex, launches, iters);
  run<6>(in, bad, nex, ex, launches, iters); run<7>(in, bad, nex, ex, launches, iters); run<8>(in, bad, nex, ex, launches, iters);
  run<9>(in, bad, nex, ex, launches, iters); run<10>(in, bad, nex, ex, launches, iters); run<11>(in, bad, nex, ex, launches, iters);
  run<12>(in, bad, nex, ex, launches, iters);
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
  unsigned long long r[NV * 4];
  float e[NV * 4 * 8];
  hipMemcpy(r, bad, sizeof(r), hipMemcpyDeviceToHost);
  hipMemcpy(e, ex, sizeof(e), hipMemcpyDeviceToHost);
  const char* names[NV] = {"mfma, distance 0, src1 HIGH half (op_sel:[0,1]), neg", "mfma, distance 1, src1 HIGH half, neg", "mfma, distance 2, src1 HIGH half, neg",
                           "mfma, distance 4, src1 HIGH half, neg", "mfma, distance 8, src1 HIGH half, neg", "no mfma, distance 0, src1 HIGH half, neg",
                           "no mfma, distance 8, src1 HIGH half, neg", "mfma, distance 0, src1 LOW half (op_sel_hi:[1,0]), neg", "mfma, distance 8, src1 LOW half, neg",
                           "mfma, distance 0, src1 HIGH half, no neg (add)", "mfma, distance 0, v_pk_mul_f32 src1 HIGH half", "mfma, distance 0, HIGH half as src0 (op_sel:[1,0])",
                           "mfma with 32+ idle cycles on both sides, src1 HIGH half, neg"};
  const double trials = (double)launches * n * iters * 2;
  for (int v = 0; v < NV; ++v) {
    printf("{\"variant\": \"%s\", \"results_checked\": %.3g, \"wrong_by_quarter_wave\": [%llu, %llu, %llu, %llu]", names[v], trials, r[v * 4], r[v * 4 + 1],
           r[v * 4 + 2], r[v * 4 + 3]);
    const float* x = e + v * 32;
    if (r[v * 4] + r[v * 4 + 1] + r[v * 4 + 2] + r[v * 4 + 3])
      printf(", \"example\": {\"result\": \"%s\", \"got\": %.9g, \"want\": %.9g, \"src0\": %.9g, \"src1_low_half\": %.9g, \"src1_high_half\": %.9g, \"lane\": %d}",
             x[0] == 0.f ? "low" : "high", x[1], x[2], x[3], x[4], x[5], (int)x[6]);
    printf("}\n");
  }
  return 0;
}
