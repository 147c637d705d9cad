// tickets_lab — the dynamic tile order of kernel_regtile64p.h (TICKETS, round 5) against the static map, on the library's own kernel
// templates: bit equality of the whole output in the normal case AND in the cases the claim bits / the sweep exist for (no tickets left
// for anybody, half the stream missing, poisoned mailboxes, a partial last gang), plus interleaved timing.
// tests/test_tickets_gpu.py runs it; __graft_entry__.build() compiles it in-tree.    usage: tickets_lab [rounds]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-inline-asm tools/tickets_lab.hip -o tools/tickets_lab
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
#include "../fft_amd/csrc/kernel_regtile_mixedp.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
using namespace sfft;

__global__ void count_diff(const uint32_t* a, const uint32_t* b, size_t n, unsigned long long* out) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}
// per (tile, row-group-of-512-rows) mismatch counts of an (B, N, D) fp32 tensor in 16-channel tiles: which tiles / which parts of them differ
__global__ void diff_map(const uint32_t* a, const uint32_t* b, int n_rows, int d, int n_tiles, unsigned* out) {
  const int tile = blockIdx.x, tpr = d / 16, bb = tile / tpr, ct = tile % tpr;
  for (int i = threadIdx.x; i < n_rows * 16; i += blockDim.x) {
    const int r = i / 16, c = i % 16;
    const size_t off = ((size_t)bb * n_rows + r) * d + ct * 16 + c;
    if (a[off] != b[off]) atomicAdd(out + tile * 8 + (r * 8) / n_rows, 1u);
  }
}
__global__ void to_bf16(const float* src, uint16_t* dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint16_t)sfft::f32_to_bf16_rne(src[i]);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 5;
  const int only_how = argc > 2 ? atoi(argv[2]) : -1;       // debugging aid: run the fp32 4096 case with this slice state only, repeatedly
  const int B = 256, N = 4096, D = 768, G = 4, F = N / 2 + 1;
  const size_t n_el = (size_t)B * N * D;
  float *v, *out, *out_ref; float2 *gate, *tw;
  CK(hipMalloc(&v, n_el * 4)); CK(hipMalloc(&out, n_el * 4)); CK(hipMalloc(&out_ref, n_el * 4));
  CK(hipMalloc(&gate, (size_t)B * G * F * 8)); CK(hipMalloc(&tw, N * 8));
  {
    std::vector<float> hr(1 << 24);
    uint32_t st = 12345u;
    for (auto& x : hr) { st = st * 1664525u + 1013904223u; x = ((st >> 8) & 0xffff) / 32768.0f - 1.0f; }
    for (size_t off = 0; off < n_el; off += hr.size()) CK(hipMemcpy(v + off, hr.data(), std::min(hr.size(), n_el - off) * 4, hipMemcpyHostToDevice));
    for (size_t off = 0; off < (size_t)B * G * F * 2; off += hr.size())
      CK(hipMemcpy((float*)gate + off, hr.data(), std::min(hr.size(), (size_t)B * G * F * 2 - off) * 4, hipMemcpyHostToDevice));
  }
  std::vector<float2> h(N);
  for (int m = 0; m < N; ++m) h[m] = make_float2((float)cos(2 * M_PI * m / N), (float)-sin(2 * M_PI * m / N));
  CK(hipMemcpy(tw, h.data(), N * 8, hipMemcpyHostToDevice));
  uint16_t* vb16; CK(hipMalloc(&vb16, n_el * 2));
  hipLaunchKernelGGL(to_bf16, dim3(4096), dim3(256), 0, 0, v, vb16, n_el);
  unsigned* slice = nullptr;
  bool uncached = true;
  if (hipExtMallocWithFlags((void**)&slice, kP64TkSliceWords * 4, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); uncached = false; CK(hipMalloc(&slice, kP64TkSliceWords * 4)); }
  printf("ticket slice in %s device memory\n", uncached ? "uncached" : "PLAIN (hipDeviceMallocUncached refused)");
  unsigned long long* dc; CK(hipMalloc(&dc, 8));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;

  auto args = [&](int b, int d, const void* vin, void* o, int gang) {
    RegtileArgs a{};
    a.v = vin; a.gate = gate; a.mem = nullptr; a.out = o; a.tw = tw;
    a.B = b; a.N_in = N; a.D = d; a.G = G; a.d_g = d / G; a.F = F; a.rows_in = a.rows_out = N;
    a.v_sb = (long long)N * d; a.v_sn = d; a.out_sb = (long long)N * d; a.out_sn = d;
    a.tiles_per_row = d / 16; a.n_tiles = b * (d / 16);
    const int slots = std::max(gang, ncu / gang * gang);
    a.tpw = std::max(1, (a.n_tiles + slots - 1) / slots);
    a.n_wg = gang * ((a.n_tiles + gang * a.tpw - 1) / (gang * a.tpw));
    return a;
  };
  auto launch = [&](auto kern, RegtileArgs a, int lds) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), lds, 0, a);
  };
  // how the slice looks when the kernel starts: 0 clean | 1 counter already at the end (nobody gets a ticket: everything is swept) |
  // 2 counter half way | 3 every gang's first two mailbox slots already say "end of stream" (followers may stop following at once)
  auto prepare = [&](const RegtileArgs& a, int gang, int how) {
    std::vector<unsigned> hs(kP64TkClaim + (a.n_tiles + 31) / 32, 0u);
    const unsigned total = (unsigned)((a.n_tiles + gang - 1) / gang);
    if (how == 1) hs[0] = total;
    if (how == 2) hs[0] = total / 2;
    if (how == 3) for (int g = 0; g < a.n_wg / gang; ++g) { hs[kP64TkBox + 8 * g] = (1u << 24) | kP64TkEnd; hs[kP64TkBox + 8 * g + 1] = (2u << 24) | kP64TkEnd; }
    CK(hipMemcpy(slice, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  };
  auto diff = [&](size_t n32) { CK(hipMemset(dc, 0, 8)); hipLaunchKernelGGL(count_diff, dim3(4096), dim3(256), 0, 0, (const uint32_t*)out, (const uint32_t*)out_ref, n32, dc);
                                unsigned long long hcount = 0; CK(hipMemcpy(&hcount, dc, 8, hipMemcpyDeviceToHost)); return hcount; };
  int failures = 0;
  static const char* hown[4] = {"clean slice", "counter at the end (all tiles swept)", "counter half way (half swept)", "mailboxes say end of stream"};
  auto check = [&](const char* what, int b, int d, int gang, const void* vin, size_t out_bytes, auto k_static, auto k_tick, std::initializer_list<int> hows) {
    RegtileArgs as = args(b, d, vin, out_ref, gang), at = args(b, d, vin, out, gang);
    at.tickets = slice;
    CK(hipMemset(out_ref, 0xff, out_bytes));
    launch(k_static, as, kP64LdsTotal);
    for (int how : hows) for (int rep = 0; rep < (how == 0 ? 3 : 1); ++rep) {
      CK(hipMemset(out, 0xff, out_bytes));
      prepare(at, gang, how);
      launch(k_tick, at, kP64LdsTotalT);
      CK(hipDeviceSynchronize());
      const unsigned long long bad = diff(out_bytes / 4);
      printf("check %-34s B=%-3d D=%-4d %-40s differing dwords: %llu\n", what, b, d, hown[how], bad);
      if (bad) ++failures;
      if (bad && out_bytes == (size_t)b * N * d * 4) {     // where?  (fp32 outputs only)
        const int nt = b * (d / 16);
        unsigned* dm; CK(hipMalloc(&dm, (size_t)nt * 8 * 4)); CK(hipMemset(dm, 0, (size_t)nt * 8 * 4));
        hipLaunchKernelGGL(diff_map, dim3(nt), dim3(256), 0, 0, (const uint32_t*)out, (const uint32_t*)out_ref, N, d, nt, dm);
        std::vector<unsigned> hm((size_t)nt * 8); CK(hipMemcpy(hm.data(), dm, hm.size() * 4, hipMemcpyDeviceToHost)); CK(hipFree(dm));
        int shown = 0;
        for (int t = 0; t < nt && shown < 24; ++t) {
          unsigned tot = 0; for (int q = 0; q < 8; ++q) tot += hm[(size_t)t * 8 + q];
          if (!tot) continue;
          printf("    tile %5d (ticket %5d member %d): ", t, t / gang, t % gang);
          for (int q = 0; q < 8; ++q) printf("%6u", hm[(size_t)t * 8 + q]);
          printf("   (mismatching dwords per eighth of the rows)\n"); ++shown;
        }
      }
    }
  };
  if (only_how >= 0) {
    for (int r = 0; r < rounds; ++r) check("fp32", B, D, 2, v, n_el * 4, spectre_mix_regtile64p<3, 3, false, false, false, true, true>, spectre_mix_regtile64p<3, 3, false, false, false, true, true, true>, {only_how});
    printf("%s\n", failures ? "FAILED" : "all checks passed");
    return failures ? 1 : 0;
  }
  check("fp32", B, D, 2, v, n_el * 4, spectre_mix_regtile64p<3, 3, false, false, false, true, true>, spectre_mix_regtile64p<3, 3, false, false, false, true, true, true>, {0, 1, 2, 3});
  check("fp32, 9 tiles (partial last pair)", 3, 48, 2, v, (size_t)3 * N * 48 * 4, spectre_mix_regtile64p<3, 3, false, false, false, true, true>, spectre_mix_regtile64p<3, 3, false, false, false, true, true, true>, {0, 1, 3});
  check("fp32, 2 tiles", 1, 32, 2, v, (size_t)1 * N * 32 * 4, spectre_mix_regtile64p<3, 3, false, false, false, true, true>, spectre_mix_regtile64p<3, 3, false, false, false, true, true, true>, {0, 1});
  check("bf16 -> fp32 (gangs of four)", B, D, 4, vb16, n_el * 4, spectre_mix_regtile64p<5, 3, false, true, false, true, true>, spectre_mix_regtile64p<5, 3, false, true, false, true, true, true>, {0, 2, 3});
  check("bf16 -> bf16 (gangs of four)", B, D, 4, vb16, n_el * 2, spectre_mix_regtile64p<5, 3, false, true, true, true, true>, spectre_mix_regtile64p<5, 3, false, true, true, true, true, true>, {0, 2});
  check("bf16 -> bf16, 10 tiles", 5, 32, 4, vb16, (size_t)5 * N * 32 * 2, spectre_mix_regtile64p<5, 3, false, true, true, true, true>, spectre_mix_regtile64p<5, 3, false, true, true, true, true, true>, {0, 1});

  // ---- n_fft = 3000 (kernel_regtile_mixedp.h, the reusable form of the protocol in kernel_tickets.h): same cases
  {
    const int N3 = 3000, F3 = N3 / 2 + 1;
    float2* tw3; CK(hipMalloc(&tw3, N3 * 8));
    std::vector<float2> h3(N3);
    for (int m = 0; m < N3; ++m) h3[m] = make_float2((float)cos(2 * M_PI * m / N3), (float)-sin(2 * M_PI * m / N3));
    CK(hipMemcpy(tw3, h3.data(), N3 * 8, hipMemcpyHostToDevice));
    auto args3 = [&](int b, int d, void* o) {
      RegtileArgs a{};
      a.v = v; a.gate = gate; a.mem = nullptr; a.out = o; a.tw = tw3;
      a.B = b; a.N_in = N3; a.D = d; a.G = G; a.d_g = d / G; a.F = F3; a.rows_in = a.rows_out = N3;
      a.v_sb = (long long)N3 * d; a.v_sn = d; a.out_sb = (long long)N3 * d; a.out_sn = d;
      a.tiles_per_row = d / 16; a.n_tiles = b * (d / 16);
      const int slots = std::max(2, ncu / 2 * 2);
      a.tpw = std::max(1, (a.n_tiles + slots - 1) / slots);
      a.n_wg = 2 * ((a.n_tiles + 2 * a.tpw - 1) / (2 * a.tpw));
      return a;
    };
    auto k_s = spectre_mix_regtile_mixedp<60, 50, 28, false, 16, 8>;
    auto k_t = spectre_mix_regtile_mixedp<60, 50, 28, false, 16, 8, true>;
    const int lds_s = mixedp_lds_total<60, 50, 16>(), lds_t = mixedp_lds_bytes<60, 50, 16, true>();
    for (auto shp : {std::pair<int, int>{256, 768}, std::pair<int, int>{3, 48}}) {
      const int b = shp.first, d = shp.second;
      const size_t ob = (size_t)b * N3 * d * 4;
      RegtileArgs as = args3(b, d, out_ref), at = args3(b, d, out);
      at.tickets = slice;
      CK(hipMemset(out_ref, 0xff, ob));
      CK(hipFuncSetAttribute((const void*)k_s, hipFuncAttributeMaxDynamicSharedMemorySize, lds_s));
      CK(hipFuncSetAttribute((const void*)k_t, hipFuncAttributeMaxDynamicSharedMemorySize, lds_t));
      hipLaunchKernelGGL(k_s, dim3(as.n_wg), dim3(mixedp_launch_threads<60, 50>()), lds_s, 0, as);
      for (int how : {0, 0, 1, 2, 3}) {
        CK(hipMemset(out, 0xff, ob));
        prepare(at, 2, how);
        hipLaunchKernelGGL(k_t, dim3(at.n_wg), dim3(mixedp_launch_threads<60, 50>()), lds_t, 0, at);
        CK(hipDeviceSynchronize());
        const unsigned long long bad = diff(ob / 4);
        printf("check %-34s B=%-3d D=%-4d %-40s differing dwords: %llu\n", "fp32 n_fft = 3000", b, d, hown[how], bad);
        if (bad) ++failures;
      }
    }
    // timing
    RegtileArgs as = args3(B, D, out), at = args3(B, D, out);
    at.tickets = slice;
    const size_t used = ((size_t)kTkClaim + (at.n_tiles + 31) / 32) * 4;
    hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    float best[2] = {1e9f, 1e9f};
    for (int r = 0; r < rounds; ++r)
      for (int w = 0; w < 2; ++w) {
        auto go = [&] { if (w) { CK(hipMemsetAsync(slice, 0, used, 0)); hipLaunchKernelGGL(k_t, dim3(at.n_wg), dim3(mixedp_launch_threads<60, 50>()), lds_t, 0, at); }
                        else hipLaunchKernelGGL(k_s, dim3(as.n_wg), dim3(mixedp_launch_threads<60, 50>()), lds_s, 0, as); };
        for (int i = 0; i < 8; ++i) go();
        CK(hipEventRecord(f0)); for (int i = 0; i < 16; ++i) go(); CK(hipEventRecord(f1)); CK(hipEventSynchronize(f1));
        float ms; CK(hipEventElapsedTime(&ms, f0, f1)); best[w] = std::min(best[w], ms / 16);
      }
    printf("n_fft = 3000 (256, 3000, 768): static %.4f ms, tickets %.4f ms (%+.1f%%)\n", best[0], best[1], 100.0 * (best[1] / best[0] - 1.0));
  }

  // ---- interleaved timing, static map against tickets (the reset of the slice is part of a ticket launch, as in the library)
  struct V { const char* name; std::function<void()> go; std::vector<float> ms; };
  std::vector<V> vs;
  auto add = [&](const char* name, auto kern, RegtileArgs a, int gang, bool tick) {
    const int lds = tick ? kP64LdsTotalT : kP64LdsTotal;
    if (tick) a.tickets = slice;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const size_t used = ((size_t)kP64TkClaim + (a.n_tiles + 31) / 32) * 4;
    vs.push_back({name, [=] { if (tick) CK(hipMemsetAsync(slice, 0, used, 0)); hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), lds, 0, a); }, {}});
  };
  add("fp32 static", spectre_mix_regtile64p<3, 3, false, false, false, true, true>, args(B, D, v, out, 2), 2, false);
  add("fp32 tickets", spectre_mix_regtile64p<3, 3, false, false, false, true, true, true>, args(B, D, v, out, 2), 2, true);
  add("fp32 tickets, no sweep", spectre_mix_regtile64p<3, 3, false, false, false, true, true, 2>, args(B, D, v, out, 2), 2, true);
  add("bf16->fp32 static", spectre_mix_regtile64p<5, 3, false, true, false, true, true>, args(B, D, vb16, out, 4), 4, false);
  add("bf16->fp32 tickets", spectre_mix_regtile64p<5, 3, false, true, false, true, true, true>, args(B, D, vb16, out, 4), 4, true);
  add("bf16->bf16 static", spectre_mix_regtile64p<5, 3, false, true, true, true, true>, args(B, D, vb16, out, 4), 4, false);
  add("bf16->bf16 tickets", spectre_mix_regtile64p<5, 3, false, true, true, true, true, true>, args(B, D, vb16, out, 4), 4, true);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 40; ++i) vs[0].go();
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (size_t k = 0; k < vs.size(); ++k) {
      V& x = vs[(k + r) % vs.size()];
      for (int i = 0; i < 8; ++i) x.go();
      CK(hipEventRecord(e0));
      for (int i = 0; i < 16; ++i) x.go();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      x.ms.push_back(ms / 16);
    }
  printf("\n%-22s   min     median\n", "variant");
  for (size_t k = 0; k < vs.size(); ++k) {
    auto m = vs[k].ms; std::sort(m.begin(), m.end());
    size_t kb = k; while (kb > 0 && !strstr(vs[kb].name, "static")) --kb;
    auto b0 = vs[kb].ms; std::sort(b0.begin(), b0.end());
    printf("%-22s %7.4f %7.4f (%+5.1f%% vs its static form)\n", vs[k].name, m[0], m[m.size() / 2], 100.0 * (m[m.size() / 2] / b0[b0.size() / 2] - 1.0));
  }
  printf("%s\n", failures ? "FAILED" : "all checks passed");
  return failures ? 1 : 0;
}
