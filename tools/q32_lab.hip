// q32_lab — round 6: the 16-wave / 32-points-per-thread form of the n_fft = 4096 kernel (tools/q32.h) against the shipped 8-wave kernel
// (kernel_regtile64p.h), one process, one (V, out) pair, interleaved timing; with and without the HBM traffic (rows_in = rows_out = 0:
// the buffer ranges are empty, every request is still issued).   usage: q32_lab [rounds] [name-filter]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-inline-asm tools/q32_lab.hip -o tools/q32_lab
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
#include "../fft_amd/csrc/kernel_regtile64p.h"
#include "q32.h"
#ifdef Q32P
#include "q32p.h"
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace sfft;

struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; bool check; };

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 5;
  const char* filter = argc > 2 ? argv[2] : "";
  const int B = 256, N = 4096, D = 768, G = 4, F = N / 2 + 1;
  float *v, *out, *out_ref; float2 *gate, *tw;
  CK(hipMalloc(&v, (size_t)B * N * D * 4)); CK(hipMalloc(&out, (size_t)B * N * D * 4)); CK(hipMalloc(&out_ref, (size_t)B * N * D * 4));
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
  RegtileArgs la{};
  la.v = v; la.gate = gate; la.mem = nullptr; la.out = out; la.tw = tw;
  la.B = B; la.N_in = N; la.D = D; la.G = G; la.d_g = D / G; la.F = F; la.rows_in = la.rows_out = N;
  la.v_sb = (long long)N * D; la.v_sn = D; la.out_sb = (long long)N * D; la.out_sn = D;
  la.tiles_per_row = D / 16; la.n_tiles = B * (D / 16);
  RegtileArgs dry = la; dry.rows_in = dry.rows_out = 0;

  unsigned* cnt_uc = nullptr;
  if (hipExtMallocWithFlags((void**)&cnt_uc, 65536, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); CK(hipMalloc(&cnt_uc, 65536)); }
  CK(hipMemset(cnt_uc, 0, 65536));

  auto lib = [&](RegtileArgs a, bool tickets) {
    a.tpw = 48; a.n_wg = 2 * ((a.n_tiles + 2 * a.tpw - 1) / (2 * a.tpw));
    if (tickets) {
      a.tickets = cnt_uc;
      auto k = spectre_mix_regtile64p<3, 3, false, false, false, true, true, 1>;
      CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kP64LdsTotalT));
      return std::function<void()>([=] { CK(hipMemsetAsync(cnt_uc, 0, 65536, 0)); hipLaunchKernelGGL(k, dim3(a.n_wg), dim3(512), kP64LdsTotalT, 0, a); });
    }
    auto k = spectre_mix_regtile64p<3, 3, false, false, false, true, true>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kP64LdsTotal));
    return std::function<void()>([=] { hipLaunchKernelGGL(k, dim3(a.n_wg), dim3(512), kP64LdsTotal, 0, a); });
  };
  auto q32 = [&](RegtileArgs a, auto kern, int tpw) {
    a.tpw = tpw; a.n_wg = 2 * ((a.n_tiles + 2 * a.tpw - 1) / (2 * a.tpw));
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kQ32LdsTotal));
    return std::function<void()>([=] { hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(1024), kQ32LdsTotal, 0, a); });
  };
  // the 8-wave kernel's own one-tile-per-workgroup ancestor (kernel_regtile.h): the like-for-like partner of version 0
  auto simple64 = [&](RegtileArgs a) {
    a.tpw = 1; a.n_wg = a.n_tiles;
    auto k = spectre_mix_regtile<64, 64, false, false, 0>;
    constexpr int lds = regtile_lds_total<64, 64>();
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return std::function<void()>([=] { hipLaunchKernelGGL(k, dim3(a.n_wg), dim3(512), lds, 0, a); });
  };

  std::vector<Variant> vs;
  auto add = [&](const char* name, std::function<void()> f, bool check = true) {
    std::string t = filter; size_t p0 = 0;
    for (;;) { const size_t p1 = t.find('|', p0); const std::string alt = t.substr(p0, p1 == std::string::npos ? p1 : p1 - p0);
      if (strstr(name, alt.c_str())) { vs.push_back({name, f, {}, check}); return; } if (p1 == std::string::npos) return; p0 = p1 + 1; } };
  add("8 waves: shipped <3,3,burst,spread> static map", lib(la, false));
  add("8 waves: shipped, tickets", lib(la, true));
  add("8 waves: shipped static, NO TRAFFIC", lib(dry, false), false);
  add("8 waves: one tile per workgroup (kernel_regtile.h 64x64)", simple64(la));
  add("16 waves: one tile per workgroup", q32(la, spectre_mix_q32<1>, 1));
  add("16 waves: persistent, NOT pipelined, builtin DPP", q32(la, spectre_mix_q32<0>, 48));
  add("16 waves: persistent, NOT pipelined, builtin DPP, NO TRAFFIC", q32(dry, spectre_mix_q32<0>, 48), false);
  add("16 waves: persistent, NOT pipelined, fmac_dpp", q32(la, spectre_mix_q32<1>, 48));
  add("16 waves: persistent, NOT pipelined, fmac_dpp, NO TRAFFIC", q32(dry, spectre_mix_q32<1>, 48), false);
  add("16 waves: ... NO TRAFFIC, no LDS exchange traffic (arithmetic + barriers)", q32(dry, spectre_mix_q32<3>, 48), false);
  add("16 waves: ... NO TRAFFIC, no transforms (exchanges + cross-lane + gate)", q32(dry, spectre_mix_q32<5>, 48), false);
  add("16 waves: ... NO TRAFFIC, neither (barriers, cross-lane steps, gate)", q32(dry, spectre_mix_q32<7>, 48), false);
#ifdef Q32P
#include "q32p_variants.inc"
#endif

  // ---- correctness against the shipped kernel (different factorisation: not the same bits; error relative to the RMS of the output)
  {
    RegtileArgs r = la; r.out = out_ref;
    CK(hipMemset(out_ref, 0xff, (size_t)B * N * D * 4));
    lib(r, false)();
    CK(hipDeviceSynchronize());
    std::vector<float> ho((size_t)N * D), hrf((size_t)N * D);
    for (auto& x : vs) {
      if (!x.check) continue;
      CK(hipMemset(out, 0xff, (size_t)B * N * D * 4));
      x.launch(); CK(hipDeviceSynchronize());
      double worst = 0, rms = 0; size_t bad = 0, cnt = 0;
      for (int b : {0, 1, 97, 254, 255}) {
        CK(hipMemcpy(ho.data(), out + (size_t)b * N * D, (size_t)N * D * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hrf.data(), out_ref + (size_t)b * N * D, (size_t)N * D * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < (size_t)N * D; ++i) { rms += (double)hrf[i] * hrf[i]; ++cnt; }
        for (size_t i = 0; i < (size_t)N * D; ++i) { const double d = std::fabs((double)ho[i] - hrf[i]); if (!(d <= worst)) worst = d; if (!(d < 1e-4)) ++bad; }
      }
      rms = std::sqrt(rms / cnt);
      printf("check %-62s max |diff| vs shipped %.3e (%.2e of the output's RMS %.3f), elements off by > 1e-4: %zu\n", x.name.c_str(), worst, worst / rms, rms, bad);
    }
  }
  // ---- interleaved timing
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 40; ++i) vs[0].launch();
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (size_t k = 0; k < vs.size(); ++k) {
      Variant& x = vs[(k + r) % vs.size()];
      for (int i = 0; i < 6; ++i) x.launch();
      CK(hipEventRecord(e0));
      for (int i = 0; i < 12; ++i) x.launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      x.ms.push_back(ms / 12);
    }
  printf("\n%-64s   min     median   | per round\n", "variant");
  const float base = [&] { auto m = vs[0].ms; std::sort(m.begin(), m.end()); return m[m.size() / 2]; }();
  for (auto& x : vs) {
    auto m = x.ms; std::sort(m.begin(), m.end());
    printf("%-64s %7.4f %7.4f (%+5.1f%%) |", x.name.c_str(), m[0], m[m.size() / 2], 100.0 * (m[m.size() / 2] / base - 1.0));
    for (float t : x.ms) printf(" %.4f", t);
    printf("\n");
  }
  return 0;
}
