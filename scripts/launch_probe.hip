// Launch-floor probe (not part of the library): per-launch time of back-to-back kernels shaped like the GEMM tiles
// (256 workgroups, 512 threads, 144 KiB LDS): empty, 32 MiB streaming write, 48 MiB read + 32 MiB write.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value scripts/launch_probe.hip -o scripts/launch_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ void __launch_bounds__(512) k_empty(u32x4* y, int n) {
  extern __shared__ unsigned char smem[];
  if (n == -1) y[0] = u32x4{(unsigned)smem[threadIdx.x], 0, 0, 0};
}
__global__ void __launch_bounds__(512) k_write(u32x4* y, int per_thread) {
  extern __shared__ unsigned char smem[];
  const size_t base = (size_t)blockIdx.x * per_thread * 512;
  for (int i = 0; i < per_thread; ++i) y[base + (size_t)i * 512 + threadIdx.x] = u32x4{(unsigned)i, blockIdx.x, threadIdx.x, 7u};
}
__global__ void __launch_bounds__(512) k_copy(const u32x4* x, u32x4* y, int per_thread) {
  extern __shared__ unsigned char smem[];
  const size_t base = (size_t)blockIdx.x * per_thread * 512;
  for (int i = 0; i < per_thread; ++i) y[base + (size_t)i * 512 + threadIdx.x] = x[base + (size_t)i * 512 + threadIdx.x];
}
template <typename F>
void timeit(const char* name, F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 100; ++i) f();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %7.2f us per launch\n", name, ms * 10);
}
int main() {
  u32x4 *x, *y; hipMalloc(&x, 64 << 20); hipMalloc(&y, 64 << 20); hipMemset(x, 1, 64 << 20);
  const int lds = 144 * 1024;
  hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)k_write, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)k_copy, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  timeit("empty, 256 x 512, no LDS", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, 0, y, 0); });
  timeit("empty, 256 x 512, 144 KiB LDS", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), lds, 0, y, 0); });
  timeit("write 32 MiB, 256 x 512", [&] { hipLaunchKernelGGL(k_write, dim3(256), dim3(512), lds, 0, y, 16); });
  timeit("write 8 MiB, 256 x 512", [&] { hipLaunchKernelGGL(k_write, dim3(256), dim3(512), lds, 0, y, 4); });
  timeit("copy 32 MiB -> 32 MiB, 256 x 512", [&] { hipLaunchKernelGGL(k_copy, dim3(256), dim3(512), lds, 0, x, y, 16); });
  timeit("write 32 MiB, 2048 x 512 (no LDS)", [&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(512), 0, 0, y, 2); });
  return 0;
}
