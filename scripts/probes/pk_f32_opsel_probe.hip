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
#define TRIAL(MFMA, DIST, OPSEL)                                                       \
  asm volatile(                                                                         \
      "v_mov_b32 v73, 0x7fc00000\n\t"                                                 \
      "v_mov_b32 v72, 0x7fc00000\n\t"                                                 \
      "s_nop 4\n\t" MFMA                                                               \
      "v_lshlrev_b32 v2, 16, %2\n\t"                                                   \
      "v_lshlrev_b32 v3, 16, %3\n\t"                                                   \
      "v_and_b32 v95, 0xffff0000, %4\n\t"                                              \
      "v_lshlrev_b32 v72, 16, %4\n\t"                                                 \
      "v_mov_b32 v73, v95\n\t" DIST                                                   \
      "v_pk_add_f32 v[82:83], v[2:3], v[72:73] " OPSEL " neg_lo:[0,1] neg_hi:[0,1]\n\t" \
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
}

template <int V>
__global__ void __launch_bounds__(512, 2) probe(const uint32_t* __restrict__ in, unsigned long long* __restrict__ bad, int iters) {
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
    const float sub = V >= 7 ? s_lo : s_hi;
    float want_lo, want_hi;  // scalar subtractions, kept scalar (hipcc would pack them into the very instruction under test)
    asm volatile("s_nop 4\n\tv_sub_f32 %0, %2, %4\n\tv_sub_f32 %1, %3, %4\n\ts_nop 4" : "=&v"(want_lo), "=&v"(want_hi) : "v"(f0), "v"(f1), "v"(sub));
    nbad += (__builtin_bit_cast(uint32_t, lo) != __builtin_bit_cast(uint32_t, want_lo)) + (__builtin_bit_cast(uint32_t, hi) != __builtin_bit_cast(uint32_t, want_hi));
  }
  if (nbad) atomicAdd(&bad[V * 4 + ((threadIdx.x & 63) >> 4)], (unsigned long long)nbad);
}

template <int V>
void run(const uint32_t* in, unsigned long long* bad, int launches, int iters) {
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(probe<V>, dim3(512), dim3(512), 0, 0, in, bad, iters);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 10000, iters = 64;
  const int n = 512 * 512;
  uint32_t* h = (uint32_t*)malloc(n * 4);
  for (int i = 0; i < n; ++i) h[i] = 2654435761u * (i + 1);
  uint32_t* in;
  unsigned long long* bad;
  hipMalloc(&in, n * 4);
  hipMalloc(&bad, 9 * 4 * 8);
  hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
  hipMemset(bad, 0, 9 * 4 * 8);
  run<0>(in, bad, launches, iters); run<1>(in, bad, launches, iters); run<2>(in, bad, launches, iters); run<3>(in, bad, launches, iters);
  run<4>(in, bad, launches, iters); run<5>(in, bad, launches, iters); run<6>(in, bad, launches, iters); run<7>(in, bad, launches, iters);
  run<8>(in, bad, launches, iters);
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
  unsigned long long r[36];
  hipMemcpy(r, bad, sizeof(r), hipMemcpyDeviceToHost);
  const char* names[9] = {"mfma, distance 0, high half", "mfma, distance 1, high half", "mfma, distance 2, high half", "mfma, distance 4, high half",
                          "mfma, distance 8, high half", "no mfma, distance 0, high half", "no mfma, distance 8, high half",
                          "mfma, distance 0, LOW half", "mfma, distance 8, LOW half"};
  const double trials = (double)launches * n * iters * 2;
  for (int v = 0; v < 9; ++v)
    printf("{\"variant\": \"%s\", \"results_checked\": %.3g, \"wrong_by_quarter_wave\": [%llu, %llu, %llu, %llu]}\n", names[v], trials, r[v * 4], r[v * 4 + 1],
           r[v * 4 + 2], r[v * 4 + 3]);
  return 0;
}
