// Debug harness (not part of the library): where the time of one qbytes_mfma_large launch goes (prologue / K loop / epilogue).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DQH_LT_STAMPS scripts/probes/large_tile_timing.hip -o scripts/probes/large_tile_timing.bin   (argv[1] = "random": N(0,1)-like bf16 activations, uniform int8 weights - the power state of real data)
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../optimum_quanto_amd/csrc/qmm_mfma_large.hip"
namespace qh { int launch_status() { return hipGetLastError() == hipSuccess ? 0 : -3; } void set_last_kernel(const char*) {} }
int main(int argc, char** argv) {
  const int M = 4096, N = 4096;
  const bool random = argc > 1 && argv[1][0] == 'r';
  for (int K : {1024, 4096, 8192}) {
    std::vector<uint16_t> hx((size_t)M * K, 0x3F80); std::vector<int8_t> hw((size_t)N * K, 1); std::vector<uint16_t> hs(N, 0x3F80);
    if (random) {
      unsigned long long st = 88172645463325252ull;
      auto next = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
      for (auto& v : hx) {  // sum of four uniforms, centred: bell-shaped, |v| < 2; bf16 by truncation
        const unsigned long long r = next();
        const float f = ((float)(r & 0xFFFF) + (float)((r >> 16) & 0xFFFF) + (float)((r >> 32) & 0xFFFF) + (float)((r >> 48) & 0xFFFF)) * (1.f / 65536.f) - 2.f;
        unsigned u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16);
      }
      for (auto& v : hw) v = (int8_t)(next() >> 33);
    }
    void *x, *w, *sc, *y;
    hipMalloc(&x, hx.size() * 2); hipMalloc(&w, hw.size()); hipMalloc(&sc, N * 2); hipMalloc(&y, (size_t)M * N * 2);
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size(), hipMemcpyHostToDevice);
    hipMemcpy(sc, hs.data(), N * 2, hipMemcpyHostToDevice);
    qh::lt::Args a{x, (const uint8_t*)w, sc, nullptr, y, M, N, K, 4, 1, nullptr, nullptr};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 200; ++i) qh::lt::launch<QUANTO_HIP_BF16, qh::lt::W_I8>(a, qh::lt::CFG_256_8W, 0);
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
    // r6: the loop's duration per XCD (workgroup b runs on XCD b % 8) - are the slow workgroups of a launch one XCD's, and the same ones launch after launch?
    for (int rep = 0; rep < 3; ++rep) {
      if (rep) {
        qh::lt::launch<QUANTO_HIP_BF16, qh::lt::W_I8>(a, qh::lt::CFG_256_8W, 0);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(h, HIP_SYMBOL(qh::lt::g_stamps), sizeof(h));
      }
      printf("  launch %d, loop us per XCD (min / mean / max):", rep);
      for (int xcd = 0; xcd < 8; ++xcd) {
        double mn = 1e9, mx = 0, sum = 0;
        for (int b = xcd; b < 256; b += 8) { const double d = (h[b * 8 + 4] - h[b * 8 + 3]) * 0.01; mn = std::min(mn, d); mx = std::max(mx, d); sum += d; }
        printf("  %d: %.1f/%.1f/%.1f", xcd, mn, sum / 32, mx);
      }
      printf("\n    slowest five workgroups:");
      std::vector<std::pair<double, int>> v;
      for (int b = 0; b < 256; ++b) v.push_back({(h[b * 8 + 4] - h[b * 8 + 3]) * 0.01, b});
      std::sort(v.begin(), v.end());
      for (int i = 251; i < 256; ++i) printf(" wg %d %.1f", v[i].second, v[i].first);
      printf("; fastest five:");
      for (int i = 0; i < 5; ++i) printf(" wg %d %.1f", v[i].second, v[i].first);
      printf("\n");
    }
    hipFree(x); hipFree(w); hipFree(sc); hipFree(y);
  }
  return 0;
}
