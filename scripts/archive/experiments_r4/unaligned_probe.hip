// Probe for round 5's convolution gather: are 4- and 8-byte global loads at 2-byte-aligned addresses legal and correct on gfx950 (the driver's
// SH_MEM alignment mode), and what do they cost against aligned ones?  Prints one JSON line.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void probe(const uint16_t* x, uint32_t* out32, uint64_t* out64, int shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint16_t* p = x + 4 * i + shift;  // shift odd: 2-byte aligned only
  uint32_t v32;
  uint64_t v64;
  asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v32) : "v"(p) : "memory");
  asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v64) : "v"(p) : "memory");
  out32[i] = v32;
  out64[i] = v64;
}

__global__ void stream(const uint16_t* x, uint32_t* sink, int shift, int reps, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int r = 0; r < reps; ++r) {
    const uint16_t* p = x + ((2 * i + 2 * 64 * 1024 * r) % n) + shift;
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    acc ^= v;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const int n = 1 << 22;
  std::vector<uint16_t> h(n + 16);
  for (int i = 0; i < n + 16; ++i) h[i] = (uint16_t)(i * 40503u);
  uint16_t* x;
  uint32_t* o32;
  uint64_t* o64;
  const int threads = 1 << 16;
  hipMalloc(&x, (n + 16) * 2);
  hipMalloc(&o32, threads * 4);
  hipMalloc(&o64, threads * 8);
  hipMemcpy(x, h.data(), (n + 16) * 2, hipMemcpyHostToDevice);
  int bad[2] = {0, 0};
  for (int shift = 0; shift < 2; ++shift) {
    hipLaunchKernelGGL(probe, dim3(threads / 256), dim3(256), 0, 0, x, o32, o64, shift);
    if (hipDeviceSynchronize() != hipSuccess) { printf("{\"unaligned_probe\": \"fault at shift %d\"}\n", shift); return 0; }
    std::vector<uint32_t> r32(threads);
    std::vector<uint64_t> r64(threads);
    hipMemcpy(r32.data(), o32, threads * 4, hipMemcpyDeviceToHost);
    hipMemcpy(r64.data(), o64, threads * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < threads; ++i) {
      const uint16_t* p = h.data() + 4 * i + shift;
      const uint32_t w32 = p[0] | ((uint32_t)p[1] << 16);
      const uint64_t w64 = (uint64_t)w32 | ((uint64_t)p[2] << 32) | ((uint64_t)p[3] << 48);
      bad[shift] += (r32[i] != w32) + (r64[i] != w64);
    }
  }
  float ms[2];
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int shift = 0; shift < 2; ++shift) {
    hipLaunchKernelGGL(stream, dim3(1024), dim3(256), 0, 0, x, o32, shift, 64, n);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(stream, dim3(1024), dim3(256), 0, 0, x, o32, shift, 64, n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[shift], e0, e1);
  }
  printf("{\"unaligned_probe\": \"ok\", \"mismatches_aligned\": %d, \"mismatches_2byte_aligned\": %d, \"dword_stream_ms_aligned\": %.4f, \"dword_stream_ms_2byte_aligned\": %.4f}\n",
         bad[0], bad[1], ms[0], ms[1]);
  return 0;
}
