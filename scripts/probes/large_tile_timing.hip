// Debug harness (not part of the library): where the time of one qbytes_mfma_large launch goes (prologue / K loop / epilogue).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DQH_LT_STAMPS scripts/large_tile_timing.hip -o scripts/large_tile_timing.bin
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../optimum_quanto_amd/csrc/qmm_mfma_large.hip"
namespace qh { int launch_status() { return hipGetLastError() == hipSuccess ? 0 : -3; } void set_last_kernel(const char*) {} }
int main() {
  const int M = 4096, N = 4096;
  for (int K : {128, 1024, 4096}) {
    std::vector<uint16_t> hx((size_t)M * K, 0x3F80); std::vector<int8_t> hw((size_t)N * K, 1); std::vector<uint16_t> hs(N, 0x3F80);
    void *x, *w, *sc, *y;
    hipMalloc(&x, hx.size() * 2); hipMalloc(&w, hw.size()); hipMalloc(&sc, N * 2); hipMalloc(&y, (size_t)M * N * 2);
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size(), hipMemcpyHostToDevice);
    hipMemcpy(sc, hs.data(), N * 2, hipMemcpyHostToDevice);
    qh::lt::Args a{x, (const uint8_t*)w, sc, nullptr, y, M, N, K, 4, 1, nullptr, nullptr};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) qh::lt::launch<QUANTO_HIP_BF16, qh::lt::W_I8>(a, qh::lt::CFG_256_8W, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) qh::lt::launch<QUANTO_HIP_BF16, qh::lt::W_I8>(a, qh::lt::CFG_256_8W, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256 * 8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(qh::lt::g_stamps), sizeof(h));
    unsigned long long t0 = ~0ull, tend = 0;
    for (int b = 0; b < 256; ++b) { t0 = std::min(t0, h[b * 8]); tend = std::max(tend, h[b * 8 + 6]); }
    printf("K=%d: %.2f us per launch (events); first entry -> last exit %.2f us\n", K, ms * 50, (tend - t0) * 0.01);
    const char* names[7] = {"entry", "prologue loads landed", "after barrier", "loop start", "loop end", "stores issued", "stores done"};
    for (int i = 0; i < 7; ++i) {
      std::vector<double> v; for (int b = 0; b < 256; ++b) v.push_back((h[b * 8 + i] - t0) * 0.01);
      std::sort(v.begin(), v.end());
      printf("  %-24s min %7.2f  median %7.2f  max %7.2f us\n", names[i], v[0], v[128], v[255]);
    }
    hipFree(x); hipFree(w); hipFree(sc); hipFree(y);
  }
  return 0;
}
