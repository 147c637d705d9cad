// mixedp_stage_bench — round 4: the persistent mixed-radix kernel (kernel_regtile_mixedp.h) at n_fft = 3000 with S row blocks of the next tile
// staged by LDS-DMA before the stores: (P, S) variants timed interleaved in one process, outputs compared with the first variant, and the
// ablation table (loads / stores answered by an empty buffer range) of the shipped one.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-inline-asm tools/mixedp_stage_bench.hip -o tools/mixedp_stage_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <functional>
#include <cmath>
#include <algorithm>
#include <cstdint>
#include "../fft_amd/csrc/kernel_regtile_mixedp.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace sfft;
struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; };

int main(int argc, char** argv) {
  #ifndef RFX
#define RFX 60
#define RSX 50
#endif
  constexpr int RF = RFX, RS = RSX;
  const int rounds = argc > 1 ? atoi(argv[1]) : 6;
  const int B = 256, N = RF * RS, D = 768, G = 4, F = N / 2 + 1;
  float *v, *out, *ref; float2 *gate, *tw;
  const size_t n = (size_t)B * N * D;
  CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&ref, n * 4));
  CK(hipMalloc(&gate, (size_t)B * G * F * 8)); CK(hipMalloc(&tw, N * 8));
  {
    std::vector<float> hr(1 << 24);
    uint32_t st = 12345u;
    for (auto& x : hr) { st = st * 1664525u + 1013904223u; x = ((st >> 8) & 0xffff) / 32768.0f - 1.0f; }
    for (size_t off = 0; off < n; off += hr.size()) CK(hipMemcpy(v + off, hr.data(), std::min(hr.size(), n - off) * 4, hipMemcpyHostToDevice));
    for (size_t off = 0; off < (size_t)B * G * F * 2; off += hr.size())
      CK(hipMemcpy((float*)gate + off, hr.data(), std::min(hr.size(), (size_t)B * G * F * 2 - off) * 4, hipMemcpyHostToDevice));
  }
  std::vector<float2> h(N);
  for (int m = 0; m < N; ++m) h[m] = make_float2((float)cos(2 * M_PI * m / N), (float)-sin(2 * M_PI * m / N));
  CK(hipMemcpy(tw, h.data(), N * 8, hipMemcpyHostToDevice));
  RegtileArgs a{};
  a.v = v; a.gate = gate; a.mem = nullptr; a.out = out; a.tw = tw;
  a.B = B; a.N_in = N; a.D = D; a.G = G; a.d_g = D / G; a.F = F; a.rows_in = a.rows_out = N;
  a.v_sb = (long long)N * D; a.v_sn = D; a.out_sb = (long long)N * D; a.out_sn = D;
  a.tiles_per_row = D / 16; a.n_tiles = B * (D / 16); a.tpw = 48; a.n_wg = 2 * ((a.n_tiles + 2 * a.tpw - 1) / (2 * a.tpw));
  auto mk = [&](auto kern, size_t lds, RegtileArgs x) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return std::function<void()>([=] { hipLaunchKernelGGL(kern, dim3(x.n_wg), dim3(mixedp_launch_threads<RF, RS>()), lds, 0, x); });
  };
  std::vector<Variant> vs;
  auto add = [&](const char* name, std::function<void()> f) { vs.push_back({name, f, {}}); };
  RegtileArgs nl = a; nl.rows_in = 0;
  RegtileArgs ns = a; ns.rows_out = 0;
  RegtileArgs nn = a; nn.rows_in = 0; nn.rows_out = 0;
#define VAR(P_, S_, X_) add("P = " #P_ ", S = " #S_ ", XP = " #X_, mk(spectre_mix_regtile_mixedp<RF, RS, P_, false, S_, X_>, mixedp_lds_total<RF, RS, S_>(), a))
  VAR(28, 0, 0);              // round 3's kernel (with the round-4 engine)
#include "mixedp_stage_variants.inc"
  {
    auto k = spectre_mix_regtile_mixedp<RF, RS, 24, false, 16, 0>;
    const size_t lds = mixedp_lds_total<RF, RS, 16>();
    add("P = 24, S = 16: no loads", mk(k, lds, nl));
    add("P = 24, S = 16: no stores", mk(k, lds, ns));
    add("P = 24, S = 16: no loads, no stores", mk(k, lds, nn));
  }
  // ---- correctness against the first variant (the unstaged kernel)
  {
    RegtileArgs r = a; r.out = ref;
    CK(hipMemset(ref, 0xff, n * 4));
    mk(spectre_mix_regtile_mixedp<RF, RS, 28, false, 0, 0>, mixedp_lds_total<RF, RS, 0>(), r)();
    CK(hipDeviceSynchronize());
    std::vector<float> ho((size_t)N * D), hr((size_t)N * D);
    for (auto& x : vs) {
      if (strstr(x.name.c_str(), "no ")) continue;
      CK(hipMemset(out, 0xff, n * 4));
      x.launch(); CK(hipDeviceSynchronize());
      double worst = 0; size_t bad = 0;
      for (int b : {0, 97, 255}) {
        CK(hipMemcpy(ho.data(), out + (size_t)b * N * D, (size_t)N * D * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), ref + (size_t)b * N * D, (size_t)N * D * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < (size_t)N * D; ++i) { const double d = std::fabs((double)ho[i] - hr[i]); if (!(d <= worst)) worst = d; if (!(d < 1e-4)) ++bad; }
      }
      printf("check %-40s max |diff| vs the unstaged kernel %.3e, elements off by > 1e-4: %zu\n", x.name.c_str(), worst, bad);
    }
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 40; ++w) vs[0].launch();
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (size_t k = 0; k < vs.size(); ++k) {
      Variant& x = vs[(k + r) % vs.size()];
      for (int i = 0; i < 6; ++i) x.launch();
      CK(hipEventRecord(e0));
      for (int i = 0; i < 12; ++i) x.launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); x.ms.push_back(ms / 12);
    }
  const double bytes = 2.0 * n * 4 + (double)B * G * F * 8;
  const float base = [&] { auto m = vs[0].ms; std::sort(m.begin(), m.end()); return m[m.size() / 2]; }();
  for (auto& x : vs) {
    auto m = x.ms; std::sort(m.begin(), m.end());
    const float med = m[m.size() / 2];
    printf("%-40s min %.4f  median %.4f (%+5.1f%%)  %.0f GB/s  frac %.3f\n", x.name.c_str(), m.front(), med, 100.0 * (med / base - 1.0), bytes / med / 1e6, bytes / med / 1e6 / 8000);
  }
  return 0;
}
