// mixedp_ab_bench — the persistent mixed-radix kernel (kernel_regtile_mixedp.h) at n_fft = 3000 = 60 x 50, (256, 3000, 768) fp32:
// the ablation table VERDICT r02 asked for — the same instruction stream with part
// of the memory traffic switched off through the ARGUMENTS (rows_in / rows_out = 0: loads answered with 0 / stores dropped before they
// leave the CU; row stride 64 B and batch stride 0: every request served by the L2).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/mixedp_ab_bench.hip -o tools/mixedp_ab_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#include <cmath>
#include <algorithm>
#include <cstdint>
#include "../fft_amd/csrc/kernel_regtile_mixedp.h"
#if __has_include("_old/mixedp_old.h")
#include "_old/mixedp_old.h"      // a frozen copy of an earlier kernel, when one is being compared (not committed)
#define HAVE_OLD 1
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace sfft;
struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; };

int main() {
  constexpr int RF = 60, RS = 50;
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
  const size_t lds = mixed_lds_total<RF, RS>();
  auto mk = [&](auto kern, RegtileArgs x) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return std::function<void()>([=] { hipLaunchKernelGGL(kern, dim3(x.n_wg), dim3(mixed_threads<RF, RS>()), lds, 0, x); });
  };
  auto k0 = spectre_mix_regtile_mixedp<RF, RS, 24>;
  std::vector<Variant> vs;
  auto add = [&](const char* name, std::function<void()> f) { vs.push_back({name, f, {}}); };
  RegtileArgs nl = a; nl.rows_in = 0;
  RegtileArgs ns = a; ns.rows_out = 0;
  RegtileArgs nn = a; nn.rows_in = 0; nn.rows_out = 0;
  RegtileArgs l2 = a; l2.v_sn = 16; l2.v_sb = 0; l2.out_sn = 16; l2.out_sb = 0;          // 64-byte rows, every tile on the same 188 KiB
  RegtileArgs l2l = a; l2l.v_sn = 16; l2l.v_sb = 0;                                       // loads from the L2, stores to HBM
  RegtileArgs l2s = a; l2s.out_sn = 16; l2s.out_sb = 0;                                   // loads from HBM, stores into the L2
  add("the product (loads from HBM, stores to HBM)", mk(k0, a));
#ifdef HAVE_OLD
  auto kold = spectre_mix_regtile_mixedp_old<RF, RS, 24>;
  add("EARLIER kernel: the product", mk(kold, a));
  add("EARLIER kernel: no loads, no stores", mk(kold, nn));
  {
    RegtileArgs ar = a; ar.out = ref;
    mk(kold, ar)(); mk(k0, a)();
    CK(hipDeviceSynchronize());
    std::vector<float> ha(1 << 22), hb(1 << 22);
    double worst = 0, rms = 0;
    for (size_t off : {(size_t)0, n / 2, n - ha.size()}) {
      CK(hipMemcpy(ha.data(), out + off, ha.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), ref + off, hb.size() * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < ha.size(); ++i) { worst = std::max(worst, (double)fabsf(ha[i] - hb[i])); rms += (double)hb[i] * hb[i]; }
    }
    printf("new vs earlier kernel on 3 x 4M samples: max |diff| %.3e, rms %.3e\n", worst, sqrt(rms / (3.0 * ha.size())));
  }
#endif
  for (auto kv : {std::make_pair("ablation", k0)}) {
    std::string t = kv.first;
    add((t + ": no loads, no stores (VALU + LDS + barriers)").c_str(), mk(kv.second, nn));
    add((t + ": loads and stores served by the L2").c_str(), mk(kv.second, l2));
    add((t + ": loads from HBM, stores dropped").c_str(), mk(kv.second, ns));
    add((t + ": loads answered with 0, stores to HBM").c_str(), mk(kv.second, nl));
    add((t + ": loads from HBM, stores into the L2").c_str(), mk(kv.second, l2s));
    add((t + ": loads from the L2, stores to HBM").c_str(), mk(kv.second, l2l));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 12; ++w) vs[0].launch();                              // power-state ramp
  for (auto& x : vs) { x.launch(); x.launch(); }
  CK(hipDeviceSynchronize());
  for (int round = 0; round < 6; ++round)
    for (auto& x : vs) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) x.launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); x.ms.push_back(ms / 5);
    }
  const double bytes = 2.0 * n * 4 + (double)B * G * F * 8;
  for (auto& x : vs) {
    std::sort(x.ms.begin(), x.ms.end());
    const float med = x.ms[x.ms.size() / 2];
    printf("%-62s min %.3f  median %.3f  max %.3f ms   %.0f GB/s  frac %.3f\n", x.name.c_str(), x.ms.front(), med, x.ms.back(), bytes / med / 1e6, bytes / med / 1e6 / 8000);
  }
  return 0;
}
