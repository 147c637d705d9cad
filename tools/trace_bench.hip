// trace_bench — per-workgroup phase timestamps of spectre_mix_regtile<64,64> at (256, 4096, 768) fp32:
// when does each workgroup start, when have its loads landed, when is the middle phase done, when are the
// stores issued / acknowledged, which CU ran it.  Answers: how long does a CU wait for HBM, how big is the
// gap between consecutive workgroups on a CU, and are the CUs in lock-step (all loading at once)?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/trace_bench.hip -o tools/trace_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <cmath>
#include <algorithm>
#include <cstdint>
#include "../fft_amd/csrc/kernel_regtile.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace sfft;

static double pct(std::vector<double> v, double p) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; }

template <int ABL>
void run(const char* name, RegtileArgs a, int tpw = 1) {
  auto kern = spectre_mix_regtile<64, 64, false, false, 0, ABL, 1>;
  const size_t lds = regtile_lds_total<64, 64, 1>();
  a.tiles_per_row = a.D / 16; a.n_tiles = a.B * a.tiles_per_row;
  a.tpw = tpw;
  a.n_wg = 2 * ((a.n_tiles + 2 * a.tpw - 1) / (2 * a.tpw));
  unsigned long long* tr;
  CK(hipMalloc(&tr, (size_t)a.n_wg * 64));
  CK(hipMemset(tr, 0, (size_t)a.n_wg * 64));
  a.trace = tr;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), lds, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), lds, 0, a);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)a.n_wg * 8);
  CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
  CK(hipFree(tr));
  const bool ack = (ABL & 32) != 0;
  const int last = ack ? 6 : 5;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int w = 0; w < a.n_wg; ++w) { tmin = std::min(tmin, h[w * 8]); tmax = std::max(tmax, h[w * 8 + last]); }
  auto us = [&](unsigned long long t) { return (double)(t - tmin) * 0.01; };   // 100 MHz
  std::vector<double> d_issue, d_wait, d_f1, d_mid, d_st, d_ack, d_tot, d_gap;
  std::map<unsigned long long, std::vector<int>> per_cu;
  for (int w = 0; w < a.n_wg; ++w) {
    const unsigned long long* t = &h[w * 8];
    d_issue.push_back((t[1] - t[0]) * 0.01); d_wait.push_back((t[2] - t[1]) * 0.01); d_f1.push_back((t[3] - t[2]) * 0.01);
    d_mid.push_back((t[4] - t[3]) * 0.01); d_st.push_back((t[5] - t[4]) * 0.01);
    if (ack) d_ack.push_back((t[6] - t[5]) * 0.01);
    d_tot.push_back((t[last] - t[0]) * 0.01);
    const unsigned long long key = ((t[7] >> 32) << 16) | ((t[7] >> 8) & 0xff);   // xcc, se/sh/cu
    per_cu[key].push_back(w);
  }
  for (auto& kv : per_cu) {
    auto& v = kv.second;
    std::sort(v.begin(), v.end(), [&](int x, int y) { return h[x * 8] < h[y * 8]; });
    for (size_t i = 0; i + 1 < v.size(); ++i) d_gap.push_back(((double)h[v[i + 1] * 8] - (double)h[v[i] * 8 + last]) * 0.01);
  }
  printf("== %s (ABL=%d tpw=%d): %.3f ms, %zu CUs seen, %.1f us per tile per CU\n", name, ABL, tpw, ms, per_cu.size(), ms * 1e3 * 256 / a.n_tiles);
  auto row = [&](const char* n, std::vector<double>& v) { if (!v.empty()) printf("   %-34s p10 %7.2f  p50 %7.2f  p90 %7.2f  mean %7.2f us\n", n, pct(v, .1), pct(v, .5), pct(v, .9), [&] { double s = 0; for (double x : v) s += x; return s / v.size(); }()); };
  row("start -> loads issued", d_issue); row("loads issued -> landed (wave 0)", d_wait); row("landed -> F1+E1 done (all waves)", d_f1);
  row("middle + E2", d_mid); row("I2 + stores issued", d_st); row("stores issued -> acknowledged", d_ack); row("workgroup total", d_tot);
  row("gap to next workgroup on the CU", d_gap);
  // phase census at 200 instants in the middle 70 % of the launch
  std::vector<double> nl, nc, ns;
  const double T = us(tmax);
  for (int i = 0; i < 200; ++i) {
    const double t = T * (0.15 + 0.7 * i / 199.0);
    int l = 0, c = 0, s = 0;
    for (int w = 0; w < a.n_wg; ++w) {
      const unsigned long long* q = &h[w * 8];
      if (t < us(q[0]) || t >= us(q[last])) continue;
      if (t < us(q[2])) ++l; else if (t < us(q[4])) ++c; else ++s;
    }
    nl.push_back(l); nc.push_back(c); ns.push_back(s);
  }
  auto ms_ = [&](std::vector<double>& v) { double m = 0, q = 0; for (double x : v) m += x; m /= v.size(); for (double x : v) q += (x - m) * (x - m); return std::make_pair(m, std::sqrt(q / v.size())); };
  auto a1 = ms_(nl), a2 = ms_(nc), a3 = ms_(ns);
  printf("   census over time: loading %.1f +- %.1f, computing %.1f +- %.1f, storing %.1f +- %.1f workgroups (binomial sd would be %.1f)\n",
         a1.first, a1.second, a2.first, a2.second, a3.first, a3.second, std::sqrt(a1.first * (1 - a1.first / 256)));
  // timeline of two CUs
  int shown = 0;
  for (auto& kv : per_cu) {
    if (shown++ >= 2) break;
    printf("   CU %llx:", kv.first);
    for (size_t i = 0; i < std::min<size_t>(5, kv.second.size()); ++i) {
      const unsigned long long* q = &h[kv.second[i] * 8];
      printf(" [wg%5d %.1f|%.1f|%.1f|%.1f|%.1f]", kv.second[i], us(q[0]), us(q[2]), us(q[3]), us(q[4]), us(q[last]));
    }
    printf("\n");
  }
  fflush(stdout);
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
  RegtileArgs a{};
  a.v = v; a.gate = gate; a.mem = nullptr; a.out = out; a.tw = tw;
  a.B = B; a.N_in = N; a.D = D; a.G = G; a.d_g = D / G; a.F = F;
  a.v_sb = (long long)N * D; a.v_sn = D; a.out_sb = (long long)N * D; a.out_sn = D;
  run<16>("full kernel", a);
  unsigned* sem; CK(hipMalloc(&sem, 8 * 64 * 4)); CK(hipMemset(sem, 0, 8 * 64 * 4));
  a.sem = sem;
  for (int k : {6, 8, 10, 12, 14, 16, 20, 24}) {
    a.sem_k = k;
    char nm[64]; snprintf(nm, 64, "full kernel, load semaphore K=%d per XCD", k);
    run<16 | 128>(nm, a);
  }
  return 0;
}
