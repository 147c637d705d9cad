// mixedp_opt_bench — the persistent mixed-radix kernels (tools/mixedpx.h = kernel_regtile_mixedp.h with switches) at (B, RF*RS, 768) fp32 with each of the round-3 tidy-ups
// switched on and off (template parameter OPT: 1 = compile-time gate side, 2 = single ds_read_b32 with immediate offsets, 4 = second
// write base beyond 64 KiB, 8 = previous output base recomputed), interleaved on one box.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/mixedp_opt_bench.hip -o tools/mixedp_opt_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#include <cmath>
#include <algorithm>
#include <cstdint>
#include "mixedpx.h"
#if __has_include("_old/mixedp_old.h")
#include "_old/mixedp_old.h"      // a frozen copy of an earlier kernel, when one is being compared (not committed)
#define HAVE_OLD 1
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace sfft;
struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; };

template <int RF, int RS, int P>
void run(int short_rows) {
  const int N = RF * RS, B = (256 * 3000) / N, D = 768, G = 4, F = N / 2 + 1;
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
  a.B = B; a.N_in = N - short_rows; a.D = D; a.G = G; a.d_g = D / G; a.F = F; a.rows_in = a.rows_out = N - short_rows;
  a.v_sb = (long long)N * D; a.v_sn = D; a.out_sb = (long long)N * D; a.out_sn = D;
  a.tiles_per_row = D / 16; a.n_tiles = B * (D / 16); a.tpw = (a.n_tiles + 255) / 256; a.n_wg = 2 * ((a.n_tiles + 2 * a.tpw - 1) / (2 * a.tpw));
  const size_t lds = mixed_lds_total<RF, RS>();
  auto mk = [&](auto kern, RegtileArgs x) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return std::function<void()>([=] { hipLaunchKernelGGL(kern, dim3(x.n_wg), dim3(mixed_threads<RF, RS>()), lds, 0, x); });
  };
  std::vector<Variant> vs;
  auto add = [&](const char* name, std::function<void()> f) { vs.push_back({name, f, {}}); };
  RegtileArgs ar = a; ar.out = ref;
  add("warm-up slot", mk(spectre_mix_regtile_mixedpx<RF, RS, P, 0>, ar));
  add("OPT=0 (none)", mk(spectre_mix_regtile_mixedpx<RF, RS, P, 0>, ar));
#ifdef HAVE_OLD
  add("EARLIER kernel", mk(spectre_mix_regtile_mixedp_old<RF, RS, P>, a));
#endif
  add("OPT=23 (library), P as shipped", mk(spectre_mix_regtile_mixedpx<RF, RS, P, 23>, a));
  add("OPT=23, P + 2", mk(spectre_mix_regtile_mixedpx<RF, RS, P + 2, 23>, a));
  add("OPT=23, P + 4", mk(spectre_mix_regtile_mixedpx<RF, RS, P + 4, 23>, a));
  add("OPT=23, P + 6", mk(spectre_mix_regtile_mixedpx<RF, RS, P + 6, 23>, a));
  add("OPT=23, P - 4", mk(spectre_mix_regtile_mixedpx<RF, RS, P - 4, 23>, a));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 30; ++w) vs[0].launch();                              // power-state ramp
  for (auto& x : vs) { x.launch(); x.launch(); }
  CK(hipDeviceSynchronize());
  {
    std::vector<float> ha(1 << 22), hb(1 << 22);
    double worst = 0;
    for (size_t off : {(size_t)0, n / 2, n - ha.size()}) {
      CK(hipMemcpy(ha.data(), out + off, ha.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), ref + off, hb.size() * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < ha.size(); ++i) worst = std::max(worst, (double)fabsf(ha[i] - hb[i]));
    }
    printf("%d x %d, %d rows short: last variant vs OPT=0 on 3 x 4M samples: max |diff| %.3e\n", RF, RS, short_rows, worst);
  }
  for (int round = 0; round < 6; ++round)
    for (auto& x : vs) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) x.launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); x.ms.push_back(ms / 5);
    }
  for (auto& x : vs) {
    std::sort(x.ms.begin(), x.ms.end());
    printf("  %-22s min %.3f  median %.3f  max %.3f ms\n", x.name.c_str(), x.ms.front(), x.ms[x.ms.size() / 2], x.ms.back());
  }
  CK(hipFree(v)); CK(hipFree(out)); CK(hipFree(ref)); CK(hipFree(gate)); CK(hipFree(tw));
}

int main() {
  run<60, 50, 24>(0);
  run<64, 40, 20>(0);
  run<60, 40, 24>(0);
  return 0;
}
