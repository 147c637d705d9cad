// p64v_phases — phase times of the shipped static forms (derived from p64v_bench.hip; round 5)
// p64v_bench — round-4 experiments on the pipelined 4096 kernel: tools/p64v.h (a copy of kernel_regtile64p.h with knobs) against the
// library kernel, interleaved A/B timing of every variant in ONE process on ONE (V, out) pair, outputs compared with the library's.
// (256, 4096, 768) fp32, pseudo-random data.   usage: p64v_bench [rounds] [name-filter]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-inline-asm tools/p64v_bench.hip -o tools/p64v_bench
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
#include "p64v.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace sfft;

__global__ void to_bf16(const float* src, uint16_t* dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint16_t)sfft::f32_to_bf16_rne(src[i]);
}

struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; };

__global__ void count_diff(const uint32_t* a, const uint32_t* b, size_t n, unsigned long long* out) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}

template <class K> std::function<void()> mk(K kern, RegtileArgs a, int gang, int lds = kV64LdsTotal) {
  a.tpw = 48;
  a.n_wg = gang * ((a.n_tiles + gang * a.tpw - 1) / (gang * a.tpw));
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  return [=] { hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), lds, 0, a); };
}

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

  unsigned* cnt_buf; CK(hipMalloc(&cnt_buf, 65536)); CK(hipMemset(cnt_buf, 0, 65536));
  RegtileArgs ls = la; ls.mem = reinterpret_cast<const float*>(cnt_buf);      // SYNCP variants: a.mem carries the gang counters
  auto synced = [&](std::function<void()> f) { return std::function<void()>([=] { CK(hipMemsetAsync(cnt_buf, 0, 65536, 0)); f(); }); };
  // the same in UNCACHED device memory (MTYPE UC: the L2 does not keep it, atomics are performed at the memory side): what a counter shared
  // by workgroups on DIFFERENT XCDs needs when it is driven by scalar atomics, which carry no scope bits (round 5, MAPX = 5)
  unsigned* cnt_uc = nullptr;
  if (hipExtMallocWithFlags((void**)&cnt_uc, 65536, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); printf("hipDeviceMallocUncached refused: UC variants use plain memory\n"); CK(hipMalloc(&cnt_uc, 65536)); }
  CK(hipMemset(cnt_uc, 0, 65536));
  RegtileArgs lsu = la; lsu.mem = reinterpret_cast<const float*>(cnt_uc);
  auto synced_uc = [&](std::function<void()> f) { return std::function<void()>([=] { CK(hipMemsetAsync(cnt_uc, 0, 65536, 0)); f(); }); };
  // bf16 rows in: the same values rounded to bf16 (device-side conversion)
  uint16_t* vb16; CK(hipMalloc(&vb16, (size_t)B * N * D * 2));
  hipLaunchKernelGGL(to_bf16, dim3(4096), dim3(256), 0, 0, v, vb16, (size_t)B * N * D);
  CK(hipDeviceSynchronize());
  RegtileArgs lb = la; lb.v = vb16;
  const bool memv = argc > 3 && !strcmp(argv[3], "mem");        // fp32 rows + memory_fft
  float* memb = nullptr;
  if (memv) { CK(hipMalloc(&memb, (size_t)F * D * 8)); CK(hipMemcpy(memb, gate, (size_t)F * D * 8, hipMemcpyDeviceToDevice)); }
  RegtileArgs lm = la; lm.mem = memb;
  const bool bfo = argc > 3 && !strcmp(argv[3], "bf16out");      // bf16 rows in AND out
  const bool bf = (argc > 3 && !strcmp(argv[3], "bf16")) || bfo;
  std::vector<Variant> vs;
  auto add = [&](const char* name, std::function<void()> f) {      // filter: alternatives separated by '|'
    std::string t = filter; size_t p0 = 0;
    for (;;) { const size_t p1 = t.find('|', p0); const std::string alt = t.substr(p0, p1 == std::string::npos ? p1 : p1 - p0);
      if (strstr(name, alt.c_str())) { vs.push_back({name, f, {}}); return; } if (p1 == std::string::npos) return; p0 = p1 + 1; } };
  // tools/p64v_phases.hip: phase times (TSTAMP = 2) of the SHIPPED static forms with and without their traffic, at 256 and at 128 workgroups
  // (half the chip: the memory system is lightly loaded, what is left over the no-traffic run is the per-CU cost of the requests themselves)
  auto mkt = [&](auto kern, RegtileArgs a, int gang, int tpw) {
    a.tpw = tpw; a.n_wg = gang * ((a.n_tiles + gang * a.tpw - 1) / (gang * a.tpw));
    const int lds = kV64LdsTotal;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return std::function<void()>([=] { hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), lds, 0, a); });
  };
  for (int tpw : {48, 96}) {
    char nm[128];
    if (!bf) {
      auto k0 = spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 4>;
      auto k2 = spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 0, 3, 12, 2, 0, 0, 4>;
      snprintf(nm, sizeof nm, "f32 (3,3) shipped, tpw %d", tpw); add(nm, mkt(k0, la, 2, tpw));
      if (tpw == 48) {
        add("f32 (3,3) SKEW 1", mkt(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 4, 0, 0, 0, 1, 0, 1>, la, 2, tpw));
        add("f32 (3,3) SKEW 2", mkt(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 4, 0, 0, 0, 1, 0, 2>, la, 2, tpw));
        add("f32 (3,3) SKEW 4", mkt(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 4, 0, 0, 0, 1, 0, 4>, la, 2, tpw));
      }
      snprintf(nm, sizeof nm, "f32 (3,3) EARLY1 = 1, tpw %d", tpw); add(nm, mkt(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 4, 0, 0, 0, 1, 1>, la, 2, tpw));
      snprintf(nm, sizeof nm, "f32 (3,3) EARLY1 = 2 (control), tpw %d", tpw); add(nm, mkt(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 4, 0, 0, 0, 1, 2>, la, 2, tpw));
      { RegtileArgs x = ls; snprintf(nm, sizeof nm, "TSTAMP PHASES f32 (3,3) EARLY1 = 1 tpw %d", tpw); add(nm, synced(mkt(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 0, 3, 12, 2, 0, 0, 4, 0, 0, 0, 1, 1>, x, 2, tpw))); }
      for (int m = 0; m < 4; ++m) {
        RegtileArgs x = ls; if (m & 1) x.rows_in = 0; if (m & 2) x.rows_out = 0;
        snprintf(nm, sizeof nm, "TSTAMP PHASES f32 (3,3) tpw %d%s%s", tpw, (m & 1) ? ", no loads" : "", (m & 2) ? ", no stores" : "");
        add(nm, synced(mkt(k2, x, 2, tpw)));
      }
      if (tpw == 48) {   // ... and under the ticket order (pair tickets from one counter in uncached memory: the harness form of the library's TICKETS)
        auto mku = [&](auto kern) { RegtileArgs a2 = lsu; a2.tpw = 48; a2.n_wg = 256; const int lds = kV64LdsTotal + 16;
          CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
          return synced_uc(std::function<void()>([=] { hipLaunchKernelGGL(kern, dim3(a2.n_wg), dim3(512), lds, 0, a2); })); };
        add("f32 (3,3) shipped, pair tickets", mku(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 5, 3, 12, 0, 0, 0, 4>));
        // the phased I/O (round 4: barriers around the store burst, -4 ... -6 % on the static map) under the TICKET order: still worth its barriers?
        add("f32 (3,3) pair tickets, SYNCP 0 (stores interleaved with the reloads)", mku(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 5, 3, 0, 0, 0, 0, 4>));
        add("f32 (3,3) pair tickets, SYNCP 4 (burst, no barrier)", mku(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 5, 3, 4, 0, 0, 0, 4>));
        add("f32 (3,3) pair tickets, SYNCP 9 (barrier in front of the burst)", mku(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 5, 3, 9, 0, 0, 0, 4>));
        add("f32 (3,3) pair tickets, SYNCP 11 (barriers around the burst)", mku(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 5, 3, 11, 0, 0, 0, 4>));
        // where the spread requests sit, re-measured under the ticket order (round 4 chose DSPREAD = 3, PFSP = 4 on the static map)
        add("f32 (3,3) pair tickets, DSPREAD 2", mku(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 5, 2, 12, 0, 0, 0, 4>));
        add("f32 (3,3) pair tickets, DSPREAD 1", mku(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 5, 1, 12, 0, 0, 0, 4>));
        add("f32 (3,3) pair tickets, PFSP 1", mku(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 5, 3, 12, 0, 0, 0, 1>));
        add("f32 (3,3) pair tickets, PFSP 5", mku(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 5, 3, 12, 0, 0, 0, 5>));
        add("f32 (3,3) pair tickets, PFSP 8", mku(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 5, 3, 12, 0, 0, 0, 8>));
        add("f32 (3,3) pair tickets, PFSP 11", mku(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 5, 3, 12, 0, 0, 0, 11>));
        add("f32 (3,3) EARLY1 = 1, pair tickets", mku(spectre_mix_p64v<3, 3, false, false, false, 2, 0, 0, 0, 0, 5, 3, 12, 0, 0, 0, 4, 0, 0, 0, 1, 1>));
      }
    } else if (!bfo) {
      snprintf(nm, sizeof nm, "bf16->f32 (5,3) shipped, tpw %d", tpw); add(nm, mkt(spectre_mix_p64v<5, 3, false, true, false, 4, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 2>, lb, 4, tpw));
      snprintf(nm, sizeof nm, "bf16->f32 (5,3) EARLY1 = 1, tpw %d", tpw); add(nm, mkt(spectre_mix_p64v<5, 3, false, true, false, 4, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 2, 0, 0, 0, 1, 1>, lb, 4, tpw));
    } else if (bfo) {
      auto k0 = spectre_mix_p64v<5, 3, false, true, true, 4, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 2>;
      auto k2 = spectre_mix_p64v<5, 3, false, true, true, 4, 0, 0, 0, 0, 0, 3, 12, 2, 0, 0, 2>;
      snprintf(nm, sizeof nm, "bf16->bf16 (5,3) shipped, tpw %d", tpw); add(nm, mkt(k0, lb, 4, tpw));
      if (tpw == 48) {
        add("bf16->bf16 (5,3) SKEW 1", mkt(spectre_mix_p64v<5, 3, false, true, true, 4, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 2, 0, 0, 0, 1, 0, 1>, lb, 4, tpw));
        add("bf16->bf16 (5,3) SKEW 2", mkt(spectre_mix_p64v<5, 3, false, true, true, 4, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 2, 0, 0, 0, 1, 0, 2>, lb, 4, tpw));
        add("bf16->bf16 (5,3) SKEW 4", mkt(spectre_mix_p64v<5, 3, false, true, true, 4, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 2, 0, 0, 0, 1, 0, 4>, lb, 4, tpw));
        add("bf16->bf16 (5,3) EARLY1 + SKEW 2", mkt(spectre_mix_p64v<5, 3, false, true, true, 4, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 2, 0, 0, 0, 1, 1, 2>, lb, 4, tpw));
      }
      snprintf(nm, sizeof nm, "bf16->bf16 (5,3) EARLY1 = 1, tpw %d", tpw); add(nm, mkt(spectre_mix_p64v<5, 3, false, true, true, 4, 0, 0, 0, 0, 0, 3, 12, 0, 0, 0, 2, 0, 0, 0, 1, 1>, lb, 4, tpw));
      { RegtileArgs x = lb; x.mem = reinterpret_cast<const float*>(cnt_buf); snprintf(nm, sizeof nm, "TSTAMP PHASES bf16->bf16 (5,3) EARLY1 = 1 tpw %d", tpw); add(nm, synced(mkt(spectre_mix_p64v<5, 3, false, true, true, 4, 0, 0, 0, 0, 0, 3, 12, 2, 0, 0, 2, 0, 0, 0, 1, 1>, x, 4, tpw))); }
      for (int m = 0; m < 4; ++m) {
        RegtileArgs x = lb; x.mem = reinterpret_cast<const float*>(cnt_buf); if (m & 1) x.rows_in = 0; if (m & 2) x.rows_out = 0;
        snprintf(nm, sizeof nm, "TSTAMP PHASES bf16->bf16 (5,3) tpw %d%s%s", tpw, (m & 1) ? ", no loads" : "", (m & 2) ? ", no stores" : "");
        add(nm, synced(mkt(k2, x, 4, tpw)));
      }
    }
  }
  // ---- correctness against the library kernel
  {
    RegtileArgs r = memv ? lm : bf ? lb : la; r.out = out_ref;
    CK(hipMemset(out_ref, 0xff, (size_t)B * N * D * 4));
    if (memv) mk(spectre_mix_regtile64p<4, 1, true>, r, 2, kP64LdsTotal)();
    else if (bfo) mk(spectre_mix_regtile64p<3, 3, false, true, true>, r, 4, kP64LdsTotal)();
    else if (bf) mk(spectre_mix_regtile64p<3, 3, false, true>, r, 4, kP64LdsTotal)(); else mk(spectre_mix_regtile64p<3, 3, false, false, false, true, true>, r, 2, kP64LdsTotal)();
    CK(hipDeviceSynchronize());
    std::vector<float> ho((size_t)N * D), hr((size_t)N * D);
    for (auto& x : vs) {
      CK(hipMemset(out, 0xff, (size_t)B * N * D * 4));
      x.launch(); CK(hipDeviceSynchronize());
      double worst = 0; size_t bad = 0, bits = 0;
      for (int b : {0, 97, 255}) {
        CK(hipMemcpy(ho.data(), out + (size_t)b * N * D, (size_t)N * D * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), out_ref + (size_t)b * N * D, (size_t)N * D * 4, hipMemcpyDeviceToHost));
        if (bfo) { for (size_t i = 0; i < (size_t)N * D; ++i) { uint32_t x, y; memcpy(&x, &ho[i], 4); memcpy(&y, &hr[i], 4); if (x != y) { ++bad; worst = 1; } } }
        else for (size_t i = 0; i < (size_t)N * D; ++i) { const double d = std::fabs((double)ho[i] - hr[i]); if (!(d <= worst)) worst = d; if (!(d < 1e-4)) ++bad; if (memcmp(&ho[i], &hr[i], 4)) ++bits; }
      }
      unsigned long long whole = 0;                      // ... and the WHOLE tensor, bit for bit (dynamic maps: a lost or doubled ticket is a lost tile anywhere)
      if (!bf) {
        unsigned long long* dc; CK(hipMalloc(&dc, 8));
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemset(out, 0xff, (size_t)B * N * D * 4)); CK(hipMemset(dc, 0, 8));
          x.launch();
          hipLaunchKernelGGL(count_diff, dim3(4096), dim3(256), 0, 0, (const uint32_t*)out, (const uint32_t*)out_ref, (size_t)B * N * D, dc);
          unsigned long long h = 0; CK(hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost)); whole += h;
        }
        CK(hipFree(dc));
      }
      printf("check %-56s max |diff| vs library %.3e, elements off by > 1e-4: %zu, elements with different bits: %zu, whole tensor x 3 launches: %llu\n", x.name.c_str(), worst, bad, bits, whole);
    }
  }
  // ---- interleaved timing
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 40; ++i) vs[0].launch();
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (size_t k = 0; k < vs.size(); ++k) {
      Variant& x = vs[(k + r) % vs.size()];            // rotate the order: no variant always follows the same neighbour
      for (int i = 0; i < 8; ++i) x.launch();
      CK(hipEventRecord(e0));
      for (int i = 0; i < 16; ++i) x.launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      x.ms.push_back(ms / 16);
    }
  printf("\n%-58s   min     median   | per round\n", "variant");
  const float base = [&] { auto m = vs[0].ms; std::sort(m.begin(), m.end()); return m[m.size() / 2]; }();
  for (auto& x : vs) {
    auto m = x.ms; std::sort(m.begin(), m.end());
    printf("%-58s %7.4f %7.4f (%+5.1f%%) |", x.name.c_str(), m[0], m[m.size() / 2], 100.0 * (m[m.size() / 2] / base - 1.0));
    for (float t : x.ms) printf(" %.4f", t);
    printf("\n");
  }
  // ---- TSTAMP variants: when does every workgroup start and finish (s_memrealtime, 100 MHz)?
  for (auto& x : vs) {
    if (!strstr(x.name.c_str(), "TSTAMP")) continue;
    const int NW = strstr(x.name.c_str(), "tpw 96") ? 128 : 256;
    for (int rep = 0; rep < 2; ++rep) {
      x.launch(); CK(hipDeviceSynchronize());
      std::vector<unsigned long long> t(512);
      CK(hipMemcpy(t.data(), reinterpret_cast<char*>(strstr(x.name.c_str(), "uncached") ? cnt_uc : cnt_buf) + 1024, 4096, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull; for (int w = 0; w < NW; ++w) t0 = std::min(t0, t[2 * w]);
      std::vector<double> st, en; for (int w = 0; w < NW; ++w) { st.push_back((t[2 * w] - t0) * 0.01); en.push_back((t[2 * w + 1] - t0) * 0.01); }
      std::vector<double> es = en; std::sort(es.begin(), es.end()); std::sort(st.begin(), st.end());
      double xm[8] = {0}; for (int w = 0; w < NW; ++w) xm[w % 8] = std::max(xm[w % 8], en[w]);
      printf("%-34s starts 0 .. %.1f us; finishes min %.1f p10 %.1f median %.1f p90 %.1f max %.1f us; last finish per XCD:", x.name.c_str(), st[NW - 1], es[0], es[NW / 10], es[NW / 2], es[NW * 9 / 10], es[NW - 1]);
      for (double v : xm) printf(" %.0f", v);
      printf("\n");
      {
        std::vector<unsigned long long> cy(512);
        CK(hipMemcpy(cy.data(), reinterpret_cast<char*>(strstr(x.name.c_str(), "uncached") ? cnt_uc : cnt_buf) + 32768, 4096, hipMemcpyDeviceToHost));
        double lo = 1e9, hi = 0, m = 0;
        for (int w = 0; w < NW; ++w) { const double mhz = (double)(cy[2 * w + 1] - cy[2 * w]) / ((t[2 * w + 1] - t[2 * w]) * 0.01); lo = std::min(lo, mhz); hi = std::max(hi, mhz); m += mhz / NW; }
        printf("    shader clocks per microsecond over the kernel (s_memtime / s_memrealtime): mean %.0f MHz, workgroups %.0f .. %.0f\n", m, lo, hi);
      }
      if (strstr(x.name.c_str(), "PHASES")) {
        std::vector<unsigned long long> ph(256 * 12);
        CK(hipMemcpy(ph.data(), reinterpret_cast<char*>(strstr(x.name.c_str(), "uncached") ? cnt_uc : cnt_buf) + 8192, ph.size() * 8, hipMemcpyDeviceToHost));
        static const char* pn[12] = {"back edge", "stage 1 of the deferred groups + wait for the LDS-DMA", "read staged groups + their stage 1", "wait for the reloaded groups + their stage 1",
          "F1 stage 2, twiddles, barrier, real-plane writes", "gate commit (waits for the gate loads)", "deferred stores / loads, E1, middle, E2", "DMA issue, twiddles, I2",
          "barrier in front of the burst", "store issue (+ trade with the prefetched rows)", "barrier behind the burst", "reload issue + gate fetch issue"};
        double tot = 0;
        for (int k = 0; k < 12; ++k) {
          std::vector<double> v; for (int w = 0; w < NW; ++w) v.push_back(ph[w * 12 + k] * 0.01 / (strstr(x.name.c_str(), "tpw 96") ? 96.0 : 48.0));
          std::sort(v.begin(), v.end()); double m = 0; for (double y : v) m += y; m /= NW; tot += m;
          printf("    %-58s mean %6.2f us per tile (workgroups: min %6.2f median %6.2f max %6.2f)\n", pn[k], m, v[0], v[NW / 2], v[NW - 1]);
        }
        printf("    sum %.2f us per tile\n", tot);
      }
    }
  }
  return 0;
}
