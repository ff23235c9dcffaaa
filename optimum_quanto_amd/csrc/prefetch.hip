// quanto_hip_prefetch: pull a byte range into the memory-side Infinity Cache (256 MiB) ahead of the kernel that will stream it.
//
// A decode step of a quantized Linear is a few microseconds of weight streaming behind ~2 us of first-byte latency; the
// int4 weights of a whole Llama-3-8B layer (109 MB) fit the Infinity Cache, and between two quantized Linears the model runs kernels
// that leave the HBM idle (norms, RoPE, attention over a short cache, SiLU).  This kernel touches ONE dword per 64 bytes of the range
// from a handful of workgroups: the fabric fetches whole lines, the data is dropped, nothing is written.  It is meant for a side
// stream (low priority), next to the compute stream's kernels; the GEMV that follows then finds its rows on the die.
//
// No reference counterpart (the reference has no ROCm decode kernel to feed); used by bench.py's layer-level decode record and by
// optimum_quanto_amd.models.prefetch_next_linear.
#include "qh_common.h"

namespace qh {

// Each lane touches one 64-byte granule per load; a wave instruction covers 4 KiB, a 256-thread block 16 KiB per round.  The loads
// are asm so that hipcc cannot drop them (results unused); 8 rounds in flight per wave before one wait.
template <bool NT>
__global__ void __launch_bounds__(256) prefetch_touch_kernel(const uint8_t* __restrict__ base, size_t bytes) {
  const size_t granules = (bytes + 63) / 64;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  while (g < granules) {
    uint32_t sink[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const size_t gi = g + (size_t)r * stride;
      const uint8_t* p = base + (gi < granules ? gi : granules - 1) * 64;
      if constexpr (NT)  // streaming policy in the L2 of the touching XCD (the consumer runs on all eight: only the Infinity Cache is shared)
        asm volatile("global_load_dword %0, %1, off nt" : "=v"(sink[r]) : "v"(p) : "memory");
      else
        asm volatile("global_load_dword %0, %1, off" : "=v"(sink[r]) : "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 8; ++r) asm volatile("" ::"v"(sink[r]));
    g += 8 * stride;
  }
}

int prefetch_range(const void* ptr, size_t bytes, int workgroups, hipStream_t stream) {
  if (bytes == 0) return QUANTO_HIP_OK;
  const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
  const uint8_t* base = reinterpret_cast<const uint8_t*>(a & ~(uintptr_t)63);  // whole granules: never reads outside the lines the range touches
  bytes += (size_t)(a & 63);
  if (workgroups <= 0) workgroups = 32;
  if (workgroups > 1024) workgroups = 1024;
  const size_t need = (bytes + 63) / 64;
  const size_t max_useful = (need + 255) / 256;
  if ((size_t)workgroups > max_useful) workgroups = (int)max_useful;
  if (env_int("QUANTO_HIP_PREFETCH_NT", 1))  // experiments
    hipLaunchKernelGGL(prefetch_touch_kernel<true>, dim3(workgroups), dim3(256), 0, stream, base, bytes);
  else
    hipLaunchKernelGGL(prefetch_touch_kernel<false>, dim3(workgroups), dim3(256), 0, stream, base, bytes);
  return launch_status();
}

}  // namespace qh
