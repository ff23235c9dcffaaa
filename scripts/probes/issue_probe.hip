// Issue-rate probe (not part of the library): one wave per SIMD on every CU runs a loop of independent 16x16x32 bf16 MFMAs
// with K filler instructions of one kind behind each MFMA, and reports time per MFMA in ns and in shader clocks.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/issue_probe.hip -o scripts/issue_probe.bin && scripts/issue_probe.bin
// Answers: how many VALU / LDS / SALU instructions fit in the shadow of an MFMA in a single-wave instruction stream, which
// conversion instructions are full rate, and what the sustained shader clock is under MFMA load.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

enum { F_NONE, F_CVT_SDWA, F_CVT_PK, F_AND_OR, F_PERM, F_DSREAD, F_SNOP, F_CVT_UBYTE, F_MIX3, F_I8MFMA, F_CVT_FP8, F_MFMA32, F_MFMA32_MIX };
typedef __attribute__((ext_vector_type(16))) float f32x16;

// DATA: 0 = the structured constants above every variant used in r1, 1 = eight different pseudo-random operand pairs (N(0,1)-like
// bf16 values, a different pair for consecutive MFMAs: the toggling a real GEMM causes), 2 = all-zero operands
template <int KIND, int K, int WAVES, int DATA = 0>
__global__ void __launch_bounds__(WAVES * 64) probe(unsigned long long* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  f32x4 acc[8];
  f32x16 acc32[4];
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 16; ++e) acc32[j][e] = seed;
  i32x4 iacc[8];
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{seed, 0.f, 0.f, 0.f}, iacc[j] = i32x4{0, 0, 0, 0};
  bf16x8 a, b, ar[8], br[8];
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)(seed + e), b[e] = (__bf16)(seed - e);
  for (int j = 0; j < 8; ++j)
    for (int e = 0; e < 8; ++e) {
      unsigned h = (threadIdx.x * 8 + e + j * 1031 + blockIdx.x * 7919) * 2654435761u;
      h ^= h >> 15;
      h *= 2246822519u;
      h ^= h >> 13;
      // sum of four uniform bytes, centred: roughly normal, |v| < 2
      const float u = ((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)) * (1.f / 128.f) - 3.984375f;
      const unsigned g = h * 3266489917u;
      const float v = ((g & 255) + ((g >> 8) & 255) + ((g >> 16) & 255) + (g >> 24)) * (1.f / 128.f) - 3.984375f;
      ar[j][e] = DATA == 1 ? (__bf16)(u * seed) : (__bf16)0.f;
      br[j][e] = DATA == 1 ? (__bf16)(v * seed) : (__bf16)0.f;
    }
  unsigned r0 = lane * 0x01010101u + (unsigned)seed, r1 = 0, r2 = 0, r3 = 0;
  float f0 = seed, f1 = seed + 1.f, f2 = 0.f, f3 = 0.f;
  const unsigned laddr = (threadIdx.x * 16) & 16383;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (KIND == F_MFMA32 || KIND == F_MFMA32_MIX) {
        // same flops per loop iteration: four 32x32x16 MFMAs instead of eight 16x16x32
        if (j < 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc32[j]) : "v"(DATA ? ar[j] : a), "v"(DATA ? br[j + 4] : b));
        if (j >= 4 && KIND == F_MFMA32) continue;
      } else if constexpr (KIND == F_I8MFMA)
        asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(iacc[j]) : "v"(__builtin_bit_cast(i32x4, a)), "v"(__builtin_bit_cast(i32x4, b)));
      else
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(DATA ? ar[j] : a), "v"(DATA ? br[(j + 3) & 7] : b));
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if constexpr (KIND == F_CVT_SDWA)
          asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(f2) : "v"(r0));
        else if constexpr (KIND == F_CVT_PK)
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r1) : "v"(f0), "v"(f1));
        else if constexpr (KIND == F_AND_OR)
          asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(r2) : "v"(r0), "v"(r1), "v"(r3));
        else if constexpr (KIND == F_PERM)
          asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r2) : "v"(r0), "v"(r1), "v"(r3));
        else if constexpr (KIND == F_DSREAD) {
          i32x4 d;
          asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(laddr));
        } else if constexpr (KIND == F_SNOP)
          asm volatile("s_nop 0");
        else if constexpr (KIND == F_CVT_UBYTE)
          asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f2) : "v"(r0));
        else if constexpr (KIND == F_CVT_FP8)
          asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(*reinterpret_cast<double*>(&f2)) : "v"(r0));
        else if constexpr (KIND == F_MIX3 || KIND == F_MFMA32_MIX) {  // the real conversion recipe: 2 sdwa cvt + 1 pk per output dword
          if (k % 3 == 2)
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r1) : "v"(f2), "v"(f3));
          else if (k % 3 == 1)
            asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(f3) : "v"(r0));
          else
            asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(f2) : "v"(r0));
        }
      }
    }
    if constexpr (KIND == F_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = f2 + f3 + __builtin_bit_cast(float, r1 ^ r2);
  for (int j = 0; j < 8; ++j) s += acc[j][0] + (float)iacc[j][0];
  for (int j = 0; j < 4; ++j) s += acc32[j][0];
  if (lane == 0 && blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
  if (s == 12345.678f) out[1] = 1;
}

template <int KIND, int K, int WAVES, int DATA = 0>
void run(const char* name, unsigned long long* dbg, int blocks, int iters = 4000) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<KIND, K, WAVES, DATA>), dim3(blocks), dim3(WAVES * 64), 0, 0, dbg, iters, 1.0f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<KIND, K, WAVES, DATA>), dim3(blocks), dim3(WAVES * 64), 0, 0, dbg, iters, 1.0f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
  const double n = 8.0 * iters;
  // 16x16x32-equivalent MFMAs: 16384 flops each, WAVES per CU, 256 CUs
  printf("%-22s K=%d waves/CU=%d : %6.2f ns/MFMA  %6.1f memtime-ticks/MFMA  (%.2f ticks/ns)  %.2f PFLOP/s chip\n", name, K, WAVES, ms * 1e6 / n, h[0] / n,
         h[0] / (ms * 1e6), 16384.0 * n * WAVES * blocks / (ms * 1e-3) * 1e-15);
}

#define RUN_K(KIND, NAME, W)            \
  run<KIND, 1, W>(NAME, dbg, blocks);   \
  run<KIND, 2, W>(NAME, dbg, blocks);   \
  run<KIND, 3, W>(NAME, dbg, blocks);   \
  run<KIND, 4, W>(NAME, dbg, blocks);   \
  run<KIND, 6, W>(NAME, dbg, blocks)

int main() {
  unsigned long long* dbg;
  hipMalloc(&dbg, 64);
  hipMemset(dbg, 0, 64);
  const int blocks = 256;
  // operand data and the MFMA rate: long runs (~100 ms each) so that the clock settles at that stream's power state
  run<F_NONE, 0, 4, 2>("16x16x32 zeros", dbg, blocks, 400000);
  run<F_NONE, 0, 4, 0>("16x16x32 constants", dbg, blocks, 400000);
  run<F_NONE, 0, 4, 1>("16x16x32 random", dbg, blocks, 400000);
  run<F_MFMA32, 0, 4, 2>("32x32x16 zeros", dbg, blocks, 400000);
  run<F_MFMA32, 0, 4, 0>("32x32x16 constants", dbg, blocks, 400000);
  run<F_MFMA32, 0, 4, 1>("32x32x16 random", dbg, blocks, 400000);
  run<F_MIX3, 2, 4, 1>("16x16x32 rnd+2 cvt", dbg, blocks, 400000);
  run<F_MFMA32_MIX, 2, 4, 1>("32x32x16 rnd+4 cvt", dbg, blocks, 400000);
  run<F_NONE, 0, 4>("mfma only", dbg, blocks);
  run<F_NONE, 0, 8>("mfma only", dbg, blocks);
  run<F_MFMA32, 0, 4>("32x32x16 (x0.5)", dbg, blocks);  // reported per 16x16x32-equivalent: 8 per iteration
  run<F_MFMA32, 0, 8>("32x32x16 (x0.5)", dbg, blocks);
  run<F_MFMA32_MIX, 1, 8>("32x32 + mix", dbg, blocks);   // K fillers behind each of 8 slots (4 of them hold an MFMA): 2K VALU per 32x32 MFMA
  run<F_MFMA32_MIX, 2, 8>("32x32 + mix", dbg, blocks);
  run<F_MFMA32_MIX, 3, 8>("32x32 + mix", dbg, blocks);
  run<F_I8MFMA, 0, 4>("i8 mfma", dbg, blocks);
  run<F_I8MFMA, 0, 8>("i8 mfma", dbg, blocks);
  RUN_K(F_SNOP, "s_nop", 4);
  RUN_K(F_AND_OR, "and_or", 4);
  RUN_K(F_PERM, "perm", 4);
  RUN_K(F_CVT_SDWA, "cvt_sdwa", 4);
  RUN_K(F_CVT_UBYTE, "cvt_ubyte", 4);
  RUN_K(F_CVT_PK, "cvt_pk_bf16", 4);
  RUN_K(F_CVT_FP8, "cvt_pk_fp8", 4);
  RUN_K(F_MIX3, "mix(2s+1p)", 4);
  RUN_K(F_DSREAD, "ds_read128", 4);
  RUN_K(F_MIX3, "mix(2s+1p)", 8);
  RUN_K(F_AND_OR, "and_or", 8);
  return 0;
}
