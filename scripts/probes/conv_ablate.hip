// Debug harness (not part of the library): where the time of the implicit-GEMM convolution goes.  The kernel file is compiled once per ablation
// (-DQH_CONV_ABLATE=bits: qconv_mfma.hip, top) into its own binary; each prints us per launch for two int8 3x3 layers.
//   for A in 0 1 2 4 9 16 32 63; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DQH_CONV_ABLATE=$A scripts/probes/conv_ablate.hip -o scripts/probes/conv_ablate_$A.bin; done
#include <cstdio>
#include <vector>
#include "../../optimum_quanto_amd/csrc/qconv_mfma.hip"
namespace qh { int launch_status() { return hipGetLastError() == hipSuccess ? 0 : -3; } }
int main() {
  struct Shape { int B, C, H, OC; } shapes[] = {{8, 128, 56, 128}, {8, 256, 56, 256}, {8, 128, 28, 128},
                                               {8, 32, 56, 128}, {8, 64, 56, 128}, {8, 256, 56, 128}, {8, 512, 56, 128}};  // (K scaling: run with QUANTO_HIP_CONV_SPLIT=1)
  for (const Shape& s : shapes) {
    const int K = s.C * 9, OH = s.H, OW = s.H;
    std::vector<uint16_t> hx((size_t)s.B * s.C * s.H * s.H);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0x3F00 + (uint16_t)((i * 2654435761u) >> 25);  // bf16 values in [0.5, 1)
    std::vector<int8_t> hw((size_t)s.OC * K);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (int8_t)((i * 40503u) >> 9);
    std::vector<uint16_t> hs(s.OC, 0x3C00);
    void *x, *w, *sc, *y, *ws;
    const size_t ws_bytes = 64u << 20;
    hipMalloc(&x, hx.size() * 2); hipMalloc(&w, hw.size()); hipMalloc(&sc, s.OC * 2); hipMalloc(&y, (size_t)s.B * s.OC * OH * OW * 2); hipMalloc(&ws, ws_bytes);
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size(), hipMemcpyHostToDevice);
    hipMemcpy(sc, hs.data(), s.OC * 2, hipMemcpyHostToDevice);
    auto run = [&]() {
      return qh::qbytes_conv2d_mfma(x, w, sc, nullptr, y, s.B, s.C, s.H, s.H, s.OC, 3, 3, OH, OW, 1, 1, 1, 1, 1, 1, QUANTO_HIP_BF16, QUANTO_HIP_I8, QUANTO_HIP_BF16, ws, ws_bytes, 0);
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int st = 0;
    for (int i = 0; i < 200; ++i) st |= run();   // warm-up + clock ramp
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 50; ++i) st |= run();
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("{\"ablate\": %d, \"B\": %d, \"C\": %d, \"H\": %d, \"OC\": %d, \"us_per_launch_back_to_back\": %.2f, \"status\": %d}\n", QH_CONV_ABLATE, s.B, s.C, s.H, s.OC, best * 20.f, st);
    hipFree(x); hipFree(w); hipFree(sc); hipFree(y); hipFree(ws);
  }
  return 0;
}
