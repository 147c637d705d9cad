// p64_ab_bench — A/B timing of variants of the pipelined 4096 kernel in ONE process, interleaved round-robin (box-to-box and
// DVFS drift is +-3 %: only interleaved runs on one box separate small effects).  (256, 4096, 768) fp32, random data.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/p64_ab_bench.hip -o tools/p64_ab_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#include <cmath>
#include <algorithm>
#include <cstdint>
#include "p64x.h"   // the round-2 kernel with its experiment switches (frozen copy; the library header no longer has them)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace sfft;

struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; };

template <class K> std::function<void()> make(K kern, XRegtileArgs a, int tpw, size_t lds) {
  a.tiles_per_row = a.D / 16; a.n_tiles = a.B * a.tiles_per_row;
  a.tpw = tpw; a.n_wg = 2 * ((a.n_tiles + 2 * tpw - 1) / (2 * tpw));
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return [=] { hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), lds, 0, a); };
}

int main() {
  const int B = 256, N = 4096, D = 768, G = 4, F = N / 2 + 1;
  float *v, *out; float2 *gate, *tw;
  CK(hipMalloc(&v, (size_t)B * N * D * 4)); CK(hipMalloc(&out, (size_t)B * N * D * 4));
  CK(hipMalloc(&gate, (size_t)B * G * F * 8)); CK(hipMalloc(&tw, N * 8));
  {
    std::vector<float> hr(1 << 24);
    uint32_t st = 12345u;
    for (auto& x : hr) { st = st * 1664525u + 1013904223u; x = ((st >> 8) & 0xffff) / 32768.0f - 1.0f; }
    for (size_t off = 0; off < (size_t)B * N * D; off += hr.size())
      CK(hipMemcpy(v + off, hr.data(), std::min(hr.size(), (size_t)B * N * D - off) * 4, hipMemcpyHostToDevice));
    for (size_t off = 0; off < (size_t)B * G * F * 2; off += hr.size())
      CK(hipMemcpy((float*)gate + off, hr.data(), std::min(hr.size(), (size_t)B * G * F * 2 - off) * 4, hipMemcpyHostToDevice));
  }
  std::vector<float2> h(N);
  for (int m = 0; m < N; ++m) h[m] = make_float2((float)cos(2 * M_PI * m / N), (float)-sin(2 * M_PI * m / N));
  CK(hipMemcpy(tw, h.data(), N * 8, hipMemcpyHostToDevice));
  XRegtileArgs a{};
  a.v = v; a.gate = gate; a.mem = nullptr; a.out = out; a.tw = tw;
  a.B = B; a.N_in = N; a.D = D; a.G = G; a.d_g = D / G; a.F = F;
  a.v_sb = (long long)N * D; a.v_sn = D; a.out_sb = (long long)N * D; a.out_sn = D;

  std::vector<Variant> vs;
  unsigned* cnt; CK(hipMalloc(&cnt, 256 * 128));
  auto with_sync = [&](auto kern, XRegtileArgs x, int tpw) {
    auto f = make(kern, x, tpw, xkP64LdsTotal);
    return std::function<void()>([=] { CK(hipMemsetAsync(cnt, 0, 256 * 128)); f(); });
  };
  XRegtileArgs as = a; as.gang_cnt = cnt;
  vs.push_back({"pipelined PF=1                      tpw=48", make(spectre_mix_p64x<4, 1, 0, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=2                      tpw=48", make(spectre_mix_p64x<4, 2, 0, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=3                      tpw=48", make(spectre_mix_p64x<4, 3, 0, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=3, F1 groups in index order    ", make(spectre_mix_p64x<4, 3, 16384, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=2, F1 groups in index order    ", make(spectre_mix_p64x<4, 2, 16384, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=3, gate of the first tile reused (no fetch, no commit)", make(spectre_mix_p64x<4, 3, 32768, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=3, gate fetched before the stores   ", make(spectre_mix_p64x<4, 3, 65536, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=2, gate fetched before the stores   ", make(spectre_mix_p64x<4, 2, 65536, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined SPLIT=3 PF=3, gate before the stores", make(spectre_mix_p64x<3, 3, 65536, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined SPLIT=2 PF=3, gate before the stores", make(spectre_mix_p64x<2, 3, 65536, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined SPLIT=3 PF=2, gate before the stores", make(spectre_mix_p64x<3, 2, 65536, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined SPLIT=2 PF=2, gate before the stores", make(spectre_mix_p64x<2, 2, 65536, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined SPLIT=1 PF=3, gate before the stores", make(spectre_mix_p64x<1, 3, 65536, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined SPLIT=3 PF=3, deferred requests at the top of the tile", make(spectre_mix_p64x<3, 3, 131072, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined SPLIT=3 PF=2, deferred requests at the top of the tile", make(spectre_mix_p64x<3, 2, 131072, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined SPLIT=3 PF=4                       ", make(spectre_mix_p64x<3, 4, 0, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=4                      tpw=48", make(spectre_mix_p64x<4, 4, 0, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined SPLIT=3 PF=3              tpw=48", make(spectre_mix_p64x<3, 3, 0, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined (3,3) + rendezvous at the deferred stores only", with_sync(spectre_mix_p64x<3, 3, 32 + 262144, true>, as, 48), {}});
  vs.push_back({"pipelined (3,3) + rendezvous everywhere", with_sync(spectre_mix_p64x<3, 3, 32, true>, as, 48), {}});
  vs.push_back({"pipelined (4,4)=spills + rendezvous at the deferred stores only", with_sync(spectre_mix_p64x<4, 4, 32 + 262144, true>, as, 48), {}});
  vs.push_back({"pipelined PF=1 + wave-pair rendezvous tpw=48", with_sync(spectre_mix_p64x<4, 1, 32, true>, as, 48), {}});
  vs.push_back({"pipelined PF=1, stores dropped (empty range) ", make(spectre_mix_p64x<4, 1, 256, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=1, loads answered with 0        ", make(spectre_mix_p64x<4, 1, 512, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=1, neither (VALU + LDS only)    ", make(spectre_mix_p64x<4, 1, 768, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=1, stores stay in the L2        ", make(spectre_mix_p64x<4, 1, 1024, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=1, loads hit the L2             ", make(spectre_mix_p64x<4, 1, 2048, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=1, both inside the L2           ", make(spectre_mix_p64x<4, 1, 3072, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=1, loads hit the L2, no stores  ", make(spectre_mix_p64x<4, 1, 2048 + 256, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=1, stores stay in the L2, no loads", make(spectre_mix_p64x<4, 1, 1024 + 512, true>, a, 48, xkP64LdsTotal), {}});
  for (int step : {1, 5, 7, 11, 17}) {
    XRegtileArgs ar = a; ar.pf_dist = step;
    vs.push_back({"pipelined PF=1, pairs start " + std::to_string(step) + " tiles apart in their ranges", make(spectre_mix_p64x<4, 1, 8192, true>, ar, 48, xkP64LdsTotal), {}});
  }
  vs.push_back({"pipelined PF=1, chip-wide sweep              ", make(spectre_mix_p64x<4, 1, 4096, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=1, chip-wide sweep + rendezvous ", with_sync(spectre_mix_p64x<4, 1, 4096 + 32, true>, as, 48), {}});
  vs.push_back({"pipelined PF=1 + rendezvous every 2nd group", with_sync(spectre_mix_p64x<4, 1, 32 + 64, true>, as, 48), {}});
  vs.push_back({"pipelined PF=1 + rendezvous every 4th group", with_sync(spectre_mix_p64x<4, 1, 32 + 128, true>, as, 48), {}});
  vs.push_back({"pipelined PF=0 fenced               tpw=48", make(spectre_mix_p64x<4, 0, 0, true>, a, 48, xkP64LdsTotal), {}});
  vs.push_back({"pipelined PF=0 fenced + rendezvous  tpw=48", with_sync(spectre_mix_p64x<4, 0, 32, true>, as, 48), {}});

  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& x : vs) { x.launch(); x.launch(); }
  CK(hipDeviceSynchronize());
  for (int round = 0; round < 6; ++round)
    for (auto& x : vs) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) x.launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); x.ms.push_back(ms / 5);
    }
  const double bytes = 2.0 * B * N * D * 4 + (double)B * G * F * 8;
  for (auto& x : vs) {
    std::sort(x.ms.begin(), x.ms.end());
    const float med = x.ms[x.ms.size() / 2];
    printf("%-52s min %.3f  median %.3f  max %.3f ms   %.0f GB/s  frac %.3f\n", x.name.c_str(), x.ms.front(), med, x.ms.back(), bytes / med / 1e6, bytes / med / 1e6 / 8000);
  }
  // rendezvous statistics of one launch: arrivals per wave pair (2 x rendezvous per wave if nobody dropped out), failed polls, survivors
  for (int pf : {1, 0}) {
    auto f = pf ? with_sync(spectre_mix_p64x<4, 1, 32, true>, as, 48) : with_sync(spectre_mix_p64x<4, 0, 32, true>, as, 48);
    f(); CK(hipDeviceSynchronize());
    std::vector<unsigned> h(256 * 32);
    CK(hipMemcpy(h.data(), cnt, 256 * 128, hipMemcpyDeviceToHost));
    unsigned lo = ~0u, hi = 0, live = 0; double polls = 0;
    for (int i = 0; i < 128 * 8; ++i) { lo = std::min(lo, h[i * 4]); hi = std::max(hi, h[i * 4]); polls += h[i * 4 + 1]; live += h[i * 4 + 2]; }
    printf("PF=%d: arrivals per wave pair min %u max %u; failed polls per wave and launch %.1f (last writer of each pair); pairs whose last writer was still live %u / 1024\n", pf, lo, hi, polls / 1024, live);
  }
  return 0;
}
