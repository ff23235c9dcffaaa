// Debug harness (not part of the library): phase-by-phase s_memtime stamps of qbytes_mfma_v2_kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DQH_PHASE_TIMING scripts/phase_timing.hip -o /tmp/phase_timing && /tmp/phase_timing
#include <cstdio>
#include <vector>
#include "../optimum-quanto_amd/csrc/qmm_mfma_v2.hip"
namespace qh { int launch_status() { return hipGetLastError() == hipSuccess ? 0 : -3; } void set_last_kernel(const char*) {} }
int main(int argc, char** argv) {
  const int M = 4096, N = 4096, K = 4096;
  std::vector<uint16_t> hx((size_t)M * K); std::vector<int8_t> hw((size_t)N * K); std::vector<uint16_t> hs(N, 0x3C00);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0x3F80 ^ ((i * 2654435761u >> 20) & 0x807F);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (int8_t)(i * 40503u >> 8);
  void *x, *w, *sc, *y; unsigned long long* dbg;
  hipMalloc(&x, hx.size() * 2); hipMalloc(&w, hw.size()); hipMalloc(&sc, N * 2); hipMalloc(&y, (size_t)M * N * 2); hipMalloc(&dbg, 8 * 34 * 8);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size(), hipMemcpyHostToDevice);
  hipMemcpy(sc, hs.data(), N * 2, hipMemcpyHostToDevice); hipMemset(dbg, 0, 8 * 34 * 8);
  qh::v2::Args a{x, (const uint8_t*)w, sc, nullptr, y, M, N, K, dbg, 0};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep)
    for (int v = 0; v < 3; ++v) {
      a.variant = v;
      for (int i = 0; i < 2; ++i) qh::v2::launch<QUANTO_HIP_BF16, qh::v2::W_I8>(a, 0);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 10; ++i) qh::v2::launch<QUANTO_HIP_BF16, qh::v2::W_I8>(a, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("variant %d: %.1f us per launch (%.0f TFLOP/s)\n", v, ms * 100, 2.0 * M * N * K / (ms * 1e-4) / 1e12);
    }
  a.variant = argc > 1 ? atoi(argv[1]) : 0;
  qh::v2::launch<QUANTO_HIP_BF16, qh::v2::W_I8>(a, 0);
  hipDeviceSynchronize();
  unsigned long long h[8 * 34]; hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < 8; ++w) if (h[w * 34] < t0) t0 = h[w * 34];
  const char* names[17] = {"start", "La done", "La rel", "Ca done", "Ca rel", "Lb done", "Lb rel", "Cb done", "Cb rel",
                           "-", "-", "-", "-", "-", "-", "-", "-"};
  printf("%-8s", "stamp");
  for (int w = 0; w < 8; ++w) printf("   wave%d", w);
  printf("\n");
  printf("HW_ID   ");
  for (int w = 0; w < 8; ++w) { unsigned v = (unsigned)h[w * 34 + 33]; printf(" s%u/w%u/cu%u", (v >> 4) & 3, v & 15, (v >> 8) & 15); }
  printf("\n");
  for (int i = 0; i < 9; ++i) {
    printf("%-8s", names[i % 17]);
    for (int w = 0; w < 8; ++w) printf(" %7llu", h[w * 34 + i] - t0);
    printf("\n");
  }
  return 0;
}
