// trace64p_bench — phase timestamps of the persistent pipelined 4096 kernel (kernel_regtile64p.h) at (256, 4096, 768) fp32.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/trace64p_bench.hip -o tools/trace64p_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#include <cstdint>
#include "p64x.h"   // the round-2 kernel with its experiment switches (frozen copy; the library header no longer has them)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace sfft;
static double pct(std::vector<double> v, double p) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; }

template <int SPLIT, int PF = 0, bool FEN = (PF > 0), int ABLX = 0>
void run(const char* name, XRegtileArgs a, int tpw) {
  auto kern = spectre_mix_p64x<SPLIT, PF, 16 | ABLX, FEN>;
  a.tiles_per_row = a.D / 16; a.n_tiles = a.B * a.tiles_per_row;
  a.tpw = tpw; a.n_wg = 2 * ((a.n_tiles + 2 * tpw - 1) / (2 * tpw));
  unsigned long long* tr;
  const size_t nrec = (size_t)a.n_wg * tpw;
  CK(hipMalloc(&tr, nrec * 64)); CK(hipMemset(tr, 0, nrec * 64));
  a.trace = tr;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, xkP64LdsTotal));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), xkP64LdsTotal, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), xkP64LdsTotal, 0, a);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(nrec * 8);
  CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost)); CK(hipFree(tr));
  std::vector<double> d[6];
  const char* nm[6] = {"top -> staged half read (wait DMA + B)", "F1 (swap, stage 1, gate commit, stage 2, twiddle)", "barrier + E1", "middle + E2", "DMA issue + I2 + stores + reloads + gate fetch", "tile period"};
  for (int w = 0; w < a.n_wg; ++w)
    for (int it = 1; it + 1 < tpw; ++it) {
      const unsigned long long* t = &h[((size_t)w * tpw + it) * 8];
      for (int k = 0; k < 5; ++k) d[k].push_back((t[k + 1] - t[k]) * 0.01);
      d[5].push_back((h[((size_t)w * tpw + it + 1) * 8] - t[0]) * 0.01);
    }
  printf("== %s SPLIT=%d PF=%d tpw=%d: %.3f ms, %.1f us per tile per CU\n", name, SPLIT, PF, tpw, ms, ms * 1e3 * 256 / a.n_tiles);
  for (int k = 0; k < 6; ++k) { double s = 0; for (double x : d[k]) s += x; printf("   %-52s p10 %6.2f p50 %6.2f p90 %6.2f mean %6.2f us\n", nm[k], pct(d[k], .1), pct(d[k], .5), pct(d[k], .9), s / std::max<size_t>(1, d[k].size())); }
  // lock-step census: number of workgroups inside "top -> F1 start" at 200 instants
  unsigned long long tmin = ~0ull, tmax = 0;
  for (size_t i = 0; i < nrec; ++i) { if (h[i * 8]) tmin = std::min(tmin, h[i * 8]); tmax = std::max(tmax, h[i * 8 + 5]); }
  std::vector<double> nw;
  for (int i = 0; i < 200; ++i) {
    const unsigned long long t = tmin + (unsigned long long)((tmax - tmin) * (0.15 + 0.7 * i / 199.0));
    int c = 0;
    for (size_t r = 0; r + 1 < nrec; ++r) if (h[r * 8 + 4] && h[r * 8 + 4] <= t && h[(r + 1) * 8 + 1] && t < h[(r + 1) * 8 + 1] && ((r + 1) % tpw) != 0) ++c;
    nw.push_back(c);
  }
  double m = 0, q = 0; for (double x : nw) m += x; m /= nw.size(); for (double x : nw) q += (x - m) * (x - m);
  printf("   workgroups inside their store/load burst (after E2 .. next F1) at an instant: %.1f +- %.1f of %d\n", m, std::sqrt(q / nw.size()), a.n_wg);
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
  XRegtileArgs a{};
  a.v = v; a.gate = gate; a.mem = nullptr; a.out = out; a.tw = tw;
  a.B = B; a.N_in = N; a.D = D; a.G = G; a.d_g = D / G; a.F = F;
  a.v_sb = (long long)N * D; a.v_sn = D; a.out_sb = (long long)N * D; a.out_sn = D;
  run<4, 0>("pipelined", a, 48);
  run<4, 0, true>("pipelined fenced", a, 48);
  run<4, 1>("pipelined", a, 48);
  run<4, 2>("pipelined", a, 48);
  // the same instruction stream with part of the traffic kept out of HBM (kernel_regtile64p.h ABL bits 8-11)
  run<4, 1, true, 1024>("stores stay in the L2", a, 48);
  run<4, 1, true, 2048>("loads hit the L2", a, 48);
  run<4, 1, true, 3072>("loads and stores inside the L2", a, 48);
  run<4, 1, true, 768>("no loads, no stores", a, 48);
  return 0;
}
