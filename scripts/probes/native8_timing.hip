// Debug harness (not part of the library): where the time of one 128-byte-row native8 launch goes (prologue / K loop / epilogue / drain), int8 x int8
// and fp8 x fp8 at (4096, 4096, K).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DQH_N8_STAMPS scripts/probes/native8_timing.hip -o scripts/probes/native8_timing.bin
#include <algorithm>
#include <cstdio>
#include <vector>
#include "../../optimum_quanto_amd/csrc/qmm_native8.hip"
namespace qh { int launch_status() { return hipGetLastError() == hipSuccess ? 0 : -3; } }
int main() {
  const int M = 4096, N = 4096;
  for (int fp8 = 0; fp8 < 2; ++fp8)
    for (int K : {1024, 4096}) {
      std::vector<uint8_t> ha((size_t)M * K), hw((size_t)N * K);
      for (size_t i = 0; i < ha.size(); ++i) ha[i] = fp8 ? (uint8_t)(0x30 + ((i * 2654435761u) >> 28)) : (uint8_t)((i * 2654435761u) >> 24);
      for (size_t i = 0; i < hw.size(); ++i) hw[i] = fp8 ? (uint8_t)(0x30 + ((i * 40503u) >> 12 & 15)) : (uint8_t)((i * 40503u) >> 8);
      std::vector<uint16_t> hs(N, 0x3C00);
      void *x, *w, *sc, *y;
      hipMalloc(&x, ha.size()); hipMalloc(&w, hw.size()); hipMalloc(&sc, N * 2); hipMalloc(&y, (size_t)M * N * 2);
      hipMemcpy(x, ha.data(), ha.size(), hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size(), hipMemcpyHostToDevice);
      hipMemcpy(sc, hs.data(), N * 2, hipMemcpyHostToDevice);
      const int dt = fp8 ? QUANTO_HIP_F8_E4M3FN : QUANTO_HIP_I8;
      auto run = [&]() { return qh::qbytes_mm_native8(x, w, sc, nullptr, y, M, N, K, dt, dt, QUANTO_HIP_BF16, 0); };
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      int st = 0;
      for (int i = 0; i < 300; ++i) st |= run();
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) st |= run();
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      static unsigned long long h[4096 * 8];
      hipMemcpyFromSymbol(h, HIP_SYMBOL(qh::n8::g_stamps), sizeof(h));
      const int nb = 256;
      unsigned long long t0 = ~0ull, tend = 0;
      for (int b = 0; b < nb; ++b) { t0 = std::min(t0, h[b * 8]); tend = std::max(tend, h[b * 8 + 4]); }
      printf("%s K=%d: %.2f us per launch back to back (status %d); first entry -> last exit %.2f us\n", fp8 ? "fp8" : "int8", K, ms * 50, st, (tend - t0) * 0.01);
      const char* names[5] = {"entry", "first K-tile visible (loop start)", "loop end", "stores issued", "stores acknowledged"};
      for (int i = 0; i < 5; ++i) {
        std::vector<double> v; for (int b = 0; b < nb; ++b) v.push_back((h[b * 8 + i] - t0) * 0.01);
        std::sort(v.begin(), v.end());
        printf("  %-36s min %7.2f  median %7.2f  max %7.2f us\n", names[i], v[0], v[nb / 2], v[nb - 1]);
      }
      hipFree(x); hipFree(w); hipFree(sc); hipFree(y);
    }
  return 0;
}
