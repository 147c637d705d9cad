// p64x_bench — round-3 experiments on the pipelined 4096 kernel (tools/p64x.h = experimental copy of kernel_regtile64p.h):
// interleaved A/B timing of variants in ONE process + per-wave phase timelines (s_memtime stamps) + output comparison against
// the baseline variant.  (256, 4096, 768) fp32, random data.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/p64x_bench.hip -o tools/p64x_bench
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
#include "p64x.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace sfft;

struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; bool check; };
constexpr int kLds = xkP64LdsTotal + 512;

template <class K> std::function<void()> make(K kern, XRegtileArgs a, int tpw) {
  a.tiles_per_row = a.D / 16; a.n_tiles = a.B * a.tiles_per_row;
  a.tpw = tpw; a.n_wg = 2 * ((a.n_tiles + 2 * tpw - 1) / (2 * tpw));
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  return [=] { hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), kLds, 0, a); };
}

static const char* kSlot[15] = {"top", "F1 stage 1 done", "F1 done (twiddle, commit, deferred I/O)", "E1 B1 passed (Wre)", "E1 B2 passed (Rre)", "E1 B3 passed (Wim)",
                                "E1 done (Rim [+B])", "middle up to stage B2", "middle done", "E2 B1 passed (Wre)", "E2 B2 passed (Rre)", "E2 B3 passed (Wim)",
                                "E2 done (Rim + B)", "I2 twiddle + stage 1", "tile end (stage 2, stores, reloads)"};

struct Geo { int n_wg, n_tiles; };
void timeline_run(const char* name, Geo a, int tpw, std::function<void(unsigned*)> go);
template <class K> void timeline(const char* name, K kern, XRegtileArgs a, int tpw) {
  a.tiles_per_row = a.D / 16; a.n_tiles = a.B * a.tiles_per_row;
  a.tpw = tpw; a.n_wg = 2 * ((a.n_tiles + 2 * tpw - 1) / (2 * tpw));
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  timeline_run(name, Geo{a.n_wg, a.n_tiles}, tpw, [=](unsigned* tr) { XRegtileArgs x = a; x.trace = reinterpret_cast<unsigned long long*>(tr);
                                                                       hipLaunchKernelGGL(kern, dim3(x.n_wg), dim3(512), kLds, 0, x); });
}
void timeline_run(const char* name, Geo a, int tpw, std::function<void(unsigned*)> go) {
  const size_t nrec = (size_t)a.n_wg * tpw * 8 * 16;
  unsigned* tr; CK(hipMalloc(&tr, nrec * 4)); CK(hipMemset(tr, 0, nrec * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) go(tr);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  go(tr);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned> h(nrec);
  CK(hipMemcpy(h.data(), tr, nrec * 4, hipMemcpyDeviceToHost)); CK(hipFree(tr));
  // mean cycles between consecutive stamps, per wave (0..7) and over all waves; the tile period from stamp 0 of consecutive tiles
  double sum[8][15] = {}, per[8] = {}; size_t cnt = 0;
  for (int w = 0; w < a.n_wg; ++w)
    for (int it = 4; it + 2 < tpw; ++it) {
      for (int wv = 0; wv < 8; ++wv) {
        const unsigned* t = &h[(((size_t)w * tpw + it) * 8 + wv) * 16];
        const unsigned* tn = &h[(((size_t)w * tpw + it + 1) * 8 + wv) * 16];
        for (int k = 1; k < 15; ++k) sum[wv][k] += (double)(unsigned)(t[k] - t[k - 1]);
        per[wv] += (double)(unsigned)(tn[0] - t[0]);
      }
      ++cnt;
    }
  printf("== timeline %s: %.3f ms; mean shader cycles per phase (s_memtime), waves 0..7 | mean over waves\n", name, ms);
  double tot = 0;
  for (int k = 1; k < 15; ++k) {
    printf("   %-42s", kSlot[k]);
    double m = 0;
    for (int wv = 0; wv < 8; ++wv) { printf(" %6.0f", sum[wv][k] / cnt); m += sum[wv][k] / cnt / 8; }
    printf(" | %6.0f\n", m); tot += m;
  }
  printf("   %-42s", "tile period");
  double mp = 0;
  for (int wv = 0; wv < 8; ++wv) { printf(" %6.0f", per[wv] / cnt); mp += per[wv] / cnt / 8; }
  printf(" | %6.0f  (sum of phases %.0f; %.2f us per tile from the launch time => %.0f MHz)\n", mp, tot, ms * 1e3 * 256 / a.n_tiles, mp / (ms * 1e3 * 256 / a.n_tiles));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
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
  XRegtileArgs a{};
  a.v = v; a.gate = gate; a.mem = nullptr; a.out = out; a.tw = tw;
  a.B = B; a.N_in = N; a.D = D; a.G = G; a.d_g = D / G; a.F = F;
  a.v_sb = (long long)N * D; a.v_sn = D; a.out_sb = (long long)N * D; a.out_sn = D;

  RegtileArgs la{};
  la.v = v; la.gate = gate; la.mem = nullptr; la.out = out; la.tw = tw;
  la.B = B; la.N_in = N; la.D = D; la.G = G; la.d_g = D / G; la.F = F; la.rows_in = la.rows_out = N;
  la.v_sb = (long long)N * D; la.v_sn = D; la.out_sb = (long long)N * D; la.out_sn = D;
  la.tiles_per_row = D / 16; la.n_tiles = B * (D / 16); la.tpw = 48; la.n_wg = 256;
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kP64LdsTotal));
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  auto lib_biglds = [=](RegtileArgs x) { return std::function<void()>([=] { hipLaunchKernelGGL((spectre_mix_regtile64p<3, 3>), dim3(x.n_wg), dim3(512), kLds, 0, x); }); };
  auto lib = [=](RegtileArgs x) { return std::function<void()>([=] { hipLaunchKernelGGL((spectre_mix_regtile64p<3, 3>), dim3(x.n_wg), dim3(512), kP64LdsTotal, 0, x); }); };
  // ---- correctness of the restructured variants against the baseline variant (same arithmetic: expected bit-equal)
  {
    XRegtileArgs r = a; r.out = out_ref;
    make(spectre_mix_p64x<3, 3, 0, true>, r, 48)();
    CK(hipDeviceSynchronize());
    std::vector<float> ho((size_t)4 * N * D), hr((size_t)4 * N * D);
    auto check = [&](const char* name, std::function<void()> f) {
      CK(hipMemset(out, 0xff, (size_t)B * N * D * 4));
      f(); CK(hipDeviceSynchronize());
      double worst = 0; size_t bad = 0;
      for (int b : {0, 97, 255}) {
        CK(hipMemcpy(ho.data(), out + (size_t)b * N * D, (size_t)N * D * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), out_ref + (size_t)b * N * D, (size_t)N * D * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < (size_t)N * D; ++i) { const double d = std::fabs((double)ho[i] - hr[i]); if (!(d <= worst)) worst = d; if (!(d < 1e-4)) ++bad; }
      }
      printf("check %-48s max |diff| vs baseline %.3e, elements off by > 1e-4: %zu\n", name, worst, bad);
    };
    check("early Wre KB0=0", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 8, 0>, a, 48));
    check("early Wre KB0=4", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 8, 4>, a, 48));
    check("early Wre KB0=4 + late E1 barrier", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 24, 4>, a, 48));
    check("LIBRARY kernel_regtile64p.h <3,3>", lib(la));
    check("stamped baseline", [&] { XRegtileArgs t = a; unsigned* tr; CK(hipMalloc(&tr, (size_t)256 * 48 * 8 * 16 * 4)); t.trace = (unsigned long long*)tr;
                                    make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 1>, t, 48)(); CK(hipDeviceSynchronize()); CK(hipFree(tr)); });
  }

  // bf16 rows in (and out): the same values rounded to bf16, in the first half of a second buffer
  uint16_t* vb16; CK(hipMalloc(&vb16, (size_t)B * N * D * 2));
  {
    std::vector<float> hf(1 << 22); std::vector<uint16_t> hb(1 << 22);
    for (size_t off = 0; off < (size_t)B * N * D; off += hf.size()) {
      const size_t cnt = std::min(hf.size(), (size_t)B * N * D - off);
      CK(hipMemcpy(hf.data(), v + off, cnt * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < cnt; ++i) { uint32_t x; memcpy(&x, &hf[i], 4); hb[i] = (uint16_t)((x + 0x7fffu + ((x >> 16) & 1u)) >> 16); }
      CK(hipMemcpy(vb16 + off, hb.data(), cnt * 2, hipMemcpyHostToDevice));
    }
  }
  RegtileArgs lb = la; lb.v = vb16;                      // bf16 in, fp32 out (gangs of 4 workgroups: n_wg stays 256)
  RegtileArgs lbb = lb; lbb.out = out_ref;               // bf16 in, bf16 out
  std::vector<Variant> vs;
  auto add = [&](const char* name, std::function<void()> f) { vs.push_back({name, f, {}, false}); };
  add("baseline (3,3)", make(spectre_mix_p64x<3, 3, 0, true>, a, 48));
  add("LIBRARY kernel_regtile64p.h <3,3>", lib(la));
  add("early Wre KB0=0 + late E1 barrier (2nd slot)", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 24, 0>, a, 48));
  add("LIBRARY kernel_regtile64p.h <3,3> (2nd slot)", lib(la));
  add("early Wre KB0=0 + late E1 barrier (3rd slot)", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 24, 0>, a, 48));
  add("baseline (3,3) (2nd slot)", make(spectre_mix_p64x<3, 3, 0, true>, a, 48));
  add("early Wre KB0=0 + late E1 barrier + Wim by columns", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 56, 0>, a, 48));
  add("LIBRARY kernel_regtile64p.h <3,3> (3rd slot)", lib(la));
  add("LIBRARY launched with 512 more bytes of LDS", lib_biglds(la));
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY <4,3>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<4, 3>), dim3(la.n_wg), dim3(512), kP64LdsTotal, 0, la); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY <3,2>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<3, 2>), dim3(la.n_wg), dim3(512), kP64LdsTotal, 0, la); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY <4,2>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<4, 2>), dim3(la.n_wg), dim3(512), kP64LdsTotal, 0, la); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY <4,1>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<4, 1>), dim3(la.n_wg), dim3(512), kP64LdsTotal, 0, la); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<8, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY bf16 -> f32 <8,0>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<8, 0, false, true>), dim3(lb.n_wg), dim3(512), kP64LdsTotal, 0, lb); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<8, 0, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY bf16 -> bf16 <8,0>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<8, 0, false, true, true>), dim3(lbb.n_wg), dim3(512), kP64LdsTotal, 0, lbb); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<6, 2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY bf16 -> f32 <6,2>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<6, 2, false, true>), dim3(lb.n_wg), dim3(512), kP64LdsTotal, 0, lb); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<6, 2, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY bf16 -> bf16 <6,2>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<6, 2, false, true, true>), dim3(lbb.n_wg), dim3(512), kP64LdsTotal, 0, lbb); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<6, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY bf16 -> f32 <6,0>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<6, 0, false, true>), dim3(lb.n_wg), dim3(512), kP64LdsTotal, 0, lb); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<6, 0, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY bf16 -> bf16 <6,0>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<6, 0, false, true, true>), dim3(lbb.n_wg), dim3(512), kP64LdsTotal, 0, lbb); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<4, 2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY bf16 -> f32 <4,2>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<4, 2, false, true>), dim3(lb.n_wg), dim3(512), kP64LdsTotal, 0, lb); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<4, 2, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY bf16 -> bf16 <4,2>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<4, 2, false, true, true>), dim3(lbb.n_wg), dim3(512), kP64LdsTotal, 0, lbb); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<3, 3, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY bf16 -> f32 <3,3>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<3, 3, false, true>), dim3(lb.n_wg), dim3(512), kP64LdsTotal, 0, lb); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<3, 3, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY bf16 -> bf16 <3,3>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<3, 3, false, true, true>), dim3(lbb.n_wg), dim3(512), kP64LdsTotal, 0, lbb); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<7, 1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY bf16 -> f32 <7,1>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<7, 1, false, true>), dim3(lb.n_wg), dim3(512), kP64LdsTotal, 0, lb); });
  CK(hipFuncSetAttribute((const void*)spectre_mix_regtile64p<7, 1, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  add("LIBRARY bf16 -> bf16 <7,1>", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<7, 1, false, true, true>), dim3(lbb.n_wg), dim3(512), kP64LdsTotal, 0, lbb); });
  add("exp early+late (4,2)", make(spectre_mix_p64x<4, 2, 0, true, false, false, false, 24, 0>, a, 48));
  add("exp early+late (4,3)", make(spectre_mix_p64x<4, 3, 0, true, false, false, false, 24, 0>, a, 48));
  add("exp early+late (4,1)", make(spectre_mix_p64x<4, 1, 0, true, false, false, false, 24, 0>, a, 48));
  add("LIBRARY <4,2> (2nd slot)", [=] { hipLaunchKernelGGL((spectre_mix_regtile64p<4, 2>), dim3(la.n_wg), dim3(512), kP64LdsTotal, 0, la); });
  add("LIBRARY <3,3> again", lib(la));
  add("early Wre KB0=0 + late E1 barrier (5th slot)", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 24, 0>, a, 48));
  add("early+late, stores dropped", make(spectre_mix_p64x<3, 3, 256, true, false, false, false, 24, 0>, a, 48));
  add("early+late, loads answered with 0", make(spectre_mix_p64x<3, 3, 512, true, false, false, false, 24, 0>, a, 48));
  add("early Wre KB0=0 + late E1 barrier (4th slot)", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 24, 0>, a, 48));
  { RegtileArgs x = la; x.rows_in = 0; x.rows_out = 0; add("LIBRARY, no traffic (rows_in = rows_out = 0)", lib(x)); }
  { RegtileArgs x = la; x.rows_out = 0; add("LIBRARY, stores dropped (rows_out = 0)", lib(x)); }
  { RegtileArgs x = la; x.rows_in = 0; add("LIBRARY, loads answered with 0 (rows_in = 0)", lib(x)); }
  add("early Wre KB0=0", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 8, 0>, a, 48));
  add("early Wre KB0=2", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 8, 2>, a, 48));
  add("early Wre KB0=4", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 8, 4>, a, 48));
  add("early Wre KB0=6", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 8, 6>, a, 48));
  add("early Wre KB0=0 + late E1 barrier", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 24, 0>, a, 48));
  add("early Wre KB0=4 + late E1 barrier", make(spectre_mix_p64x<3, 3, 0, true, false, false, false, 24, 4>, a, 48));
  // no HBM traffic (ABL 768: empty buffer ranges)
  add("no traffic: baseline", make(spectre_mix_p64x<3, 3, 768, true>, a, 48));
  add("no traffic: exchanges without LDS ops (VALU + barriers)", make(spectre_mix_p64x<3, 3, 768, true, false, false, false, 2>, a, 48));
  add("no traffic: no butterflies (LDS + barriers)", make(spectre_mix_p64x<3, 3, 768, true, false, false, false, 4>, a, 48));
  add("no traffic: neither (barriers, swaps, address code)", make(spectre_mix_p64x<3, 3, 768, true, false, false, false, 6>, a, 48));
  add("no traffic: early Wre KB0=0", make(spectre_mix_p64x<3, 3, 768, true, false, false, false, 8, 0>, a, 48));
  add("no traffic: early Wre KB0=4", make(spectre_mix_p64x<3, 3, 768, true, false, false, false, 8, 4>, a, 48));
  add("no traffic: early Wre KB0=0 + late E1 barrier", make(spectre_mix_p64x<3, 3, 768, true, false, false, false, 24, 0>, a, 48));
  add("no traffic: early Wre KB0=4 + late E1 barrier", make(spectre_mix_p64x<3, 3, 768, true, false, false, false, 24, 4>, a, 48));

  if (argc > 1 && !strcmp(argv[1], "pair")) {   // for rocprofv3 --pmc: only the library kernel, the experimental equivalent and the round-2 baseline
    std::vector<Variant> keep;
    for (auto& x : vs)
      if (x.name == "LIBRARY <4,2>" || x.name == "LIBRARY <4,3>" || x.name == "LIBRARY <4,1>" || x.name == "exp early+late (4,3)" || x.name == "exp early+late (4,2)" ||
          x.name == "exp early+late (4,1)" || x.name == "baseline (3,3)") keep.push_back(x);
    for (int r = 0; r < 6; ++r) for (auto& x : keep) { for (int i = 0; i < 8; ++i) x.launch(); CK(hipDeviceSynchronize()); }
    return 0;
  }
  if (argc > 3 && !strcmp(argv[1], "loop")) {   // tools/power_probe.sh: run ONE variant back to back for argv[3] seconds (power / clock sampling from outside)
    for (auto& x : vs)
      if (x.name == argv[2]) {
        const double secs = atof(argv[3]);
        hipEvent_t l0, l1; CK(hipEventCreate(&l0)); CK(hipEventCreate(&l1));
        int launches = 0; float total = 0;
        while (total < secs * 1e3f) {
          CK(hipEventRecord(l0));
          for (int i = 0; i < 50; ++i) x.launch();
          CK(hipEventRecord(l1)); CK(hipEventSynchronize(l1));
          float ms; CK(hipEventElapsedTime(&ms, l0, l1)); total += ms; launches += 50;
        }
        printf("loop \"%s\": %d launches, %.4f ms per launch\n", argv[2], launches, total / launches);
        return 0;
      }
    fprintf(stderr, "no variant named \"%s\"\n", argv[2]); return 1;
  }
  const bool tl_only = argc > 1 && !strcmp(argv[1], "tl");
  if (tl_only) vs.clear();
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& x : vs) { x.launch(); x.launch(); }
  CK(hipDeviceSynchronize());
  for (int round = 0; round < (quick ? 3 : 6); ++round)
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
    printf("%-58s min %.3f  median %.3f  max %.3f ms   %.0f GB/s  frac %.3f\n", x.name.c_str(), x.ms.front(), med, x.ms.back(), bytes / med / 1e6, bytes / med / 1e6 / 8000);
  }
  fflush(stdout);

  if (argc > 1 && !strcmp(argv[1], "tl")) {
    timeline("early+late (stamped), no traffic", spectre_mix_p64x<3, 3, 768, true, false, false, false, 25, 0>, a, 48);
    timeline("early+late (stamped), loads only (stores dropped)", spectre_mix_p64x<3, 3, 256, true, false, false, false, 25, 0>, a, 48);
    timeline("early+late (stamped), stores only (loads answered with 0)", spectre_mix_p64x<3, 3, 512, true, false, false, false, 25, 0>, a, 48);
    timeline("early+late (stamped), everything", spectre_mix_p64x<3, 3, 0, true, false, false, false, 25, 0>, a, 48);
    return 0;
  }
  timeline("baseline (3,3)", spectre_mix_p64x<3, 3, 0, true, false, false, false, 1>, a, 48);
  timeline("baseline (3,3), no traffic", spectre_mix_p64x<3, 3, 768, true, false, false, false, 1>, a, 48);
  timeline("no traffic, exchanges without LDS ops", spectre_mix_p64x<3, 3, 768, true, false, false, false, 3>, a, 48);
  timeline("no traffic, no butterflies", spectre_mix_p64x<3, 3, 768, true, false, false, false, 5>, a, 48);
  timeline("early Wre KB0=4 + late E1 barrier", spectre_mix_p64x<3, 3, 0, true, false, false, false, 25, 4>, a, 48);
  timeline("early Wre KB0=4 + late E1 barrier, no traffic", spectre_mix_p64x<3, 3, 768, true, false, false, false, 25, 4>, a, 48);
  return 0;
}
