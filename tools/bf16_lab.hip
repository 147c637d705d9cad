// bf16_lab — round 6: (SPLIT, PF) of the bf16-rows-in forms of the 4096 kernel UNDER THE TICKET ORDER (round 4 chose (5, 3) on the static map):
// the library's own kernel template instantiated directly, interleaved timing in one process, outputs compared with (5, 3).
// usage: bf16_lab [rounds] [f32out|bf16out]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-inline-asm tools/bf16_lab.hip -o tools/bf16_lab
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
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace sfft;

__global__ void to_bf16(const float* src, uint16_t* dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint16_t)sfft::f32_to_bf16_rne(src[i]);
}
__global__ void count_diff(const uint32_t* a, const uint32_t* b, size_t n, unsigned long long* out) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}
struct Variant { std::string name; std::function<void()> launch; std::vector<float> ms; };

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 5;
  const bool bfo = argc > 2 && !strcmp(argv[2], "bf16out");
  const int B = 256, N = 4096, D = 768, G = 4, F = N / 2 + 1;
  float *vf, *out, *out_ref; float2 *gate, *tw; uint16_t* v;
  const size_t ob = (size_t)B * N * D * (bfo ? 2 : 4);
  CK(hipMalloc(&vf, (size_t)B * N * D * 4)); CK(hipMalloc(&v, (size_t)B * N * D * 2)); CK(hipMalloc(&out, ob)); CK(hipMalloc(&out_ref, ob));
  CK(hipMalloc(&gate, (size_t)B * G * F * 8)); CK(hipMalloc(&tw, N * 8));
  {
    std::vector<float> hr(1 << 24);
    uint32_t st = 12345u;
    for (auto& x : hr) { st = st * 1664525u + 1013904223u; x = ((st >> 8) & 0xffff) / 32768.0f - 1.0f; }
    for (size_t off = 0; off < (size_t)B * N * D; off += hr.size()) CK(hipMemcpy(vf + off, hr.data(), std::min(hr.size(), (size_t)B * N * D - off) * 4, hipMemcpyHostToDevice));
    for (size_t off = 0; off < (size_t)B * G * F * 2; off += hr.size()) CK(hipMemcpy((float*)gate + off, hr.data(), std::min(hr.size(), (size_t)B * G * F * 2 - off) * 4, hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(to_bf16, dim3(4096), dim3(256), 0, 0, vf, v, (size_t)B * N * D);
  CK(hipDeviceSynchronize()); CK(hipFree(vf));
  std::vector<float2> h(N);
  for (int m = 0; m < N; ++m) h[m] = make_float2((float)cos(2 * M_PI * m / N), (float)-sin(2 * M_PI * m / N));
  CK(hipMemcpy(tw, h.data(), N * 8, hipMemcpyHostToDevice));
  RegtileArgs la{};
  la.v = v; la.gate = gate; la.mem = nullptr; la.out = out; la.tw = tw;
  la.B = B; la.N_in = N; la.D = D; la.G = G; la.d_g = D / G; la.F = F; la.rows_in = la.rows_out = N;
  la.v_sb = (long long)N * D; la.v_sn = D; la.out_sb = (long long)N * D; la.out_sn = D;
  la.tiles_per_row = D / 16; la.n_tiles = B * (D / 16);
  unsigned* cnt_uc = nullptr;
  if (hipExtMallocWithFlags((void**)&cnt_uc, 65536, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); CK(hipMalloc(&cnt_uc, 65536)); }
  auto mk = [&](auto kern, RegtileArgs a, bool tickets) {
    a.tpw = 48; a.n_wg = 4 * ((a.n_tiles + 4 * a.tpw - 1) / (4 * a.tpw));
    const int lds = tickets ? kP64LdsTotalT : kP64LdsTotal;
    if (tickets) a.tickets = cnt_uc;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return std::function<void()>([=] { if (tickets) CK(hipMemsetAsync(cnt_uc, 0, 65536, 0)); hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), lds, 0, a); });
  };
  std::vector<Variant> vs;
  auto add = [&](const char* name, std::function<void()> f) { vs.push_back({name, f, {}}); };
#define BOTH(S, P) \
  if (bfo) add("tickets (" #S "," #P ") bf16 -> bf16", mk(spectre_mix_regtile64p<S, P, false, true, true, true, true, 1>, la, true)); \
  else add("tickets (" #S "," #P ") bf16 -> fp32", mk(spectre_mix_regtile64p<S, P, false, true, false, true, true, 1>, la, true));
  BOTH(5, 3)
  if (bfo) add("static (5,3) bf16 -> bf16", mk(spectre_mix_regtile64p<5, 3, false, true, true, true, true, 0>, la, false));
  else add("static (5,3) bf16 -> fp32", mk(spectre_mix_regtile64p<5, 3, false, true, false, true, true, 0>, la, false));
#ifndef BF16_LAB_SHORT
  BOTH(4, 4) BOTH(6, 2) BOTH(4, 3) BOTH(5, 2) BOTH(7, 1) BOTH(8, 0) BOTH(3, 4)
#endif
  // correctness: every variant must give the bits of (5, 3)
  { RegtileArgs r = la; r.out = out_ref; CK(hipMemset(out_ref, 0xff, ob));
    if (bfo) mk(spectre_mix_regtile64p<5, 3, false, true, true, true, true, 0>, r, false)(); else mk(spectre_mix_regtile64p<5, 3, false, true, false, true, true, 0>, r, false)();
    CK(hipDeviceSynchronize());
    unsigned long long* dc; CK(hipMalloc(&dc, 8));
    for (auto& x : vs) {
      CK(hipMemset(out, 0xff, ob)); CK(hipMemset(dc, 0, 8));
      x.launch();
      hipLaunchKernelGGL(count_diff, dim3(4096), dim3(256), 0, 0, (const uint32_t*)out, (const uint32_t*)out_ref, ob / 4, dc);
      unsigned long long hd = 0; CK(hipMemcpy(&hd, dc, 8, hipMemcpyDeviceToHost));
      printf("check %-36s differing dwords vs static (5,3): %llu\n", x.name.c_str(), hd);
    } }
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
  printf("\n%-40s   min     median   | per round\n", "variant");
  const float base = [&] { auto m = vs[0].ms; std::sort(m.begin(), m.end()); return m[m.size() / 2]; }();
  for (auto& x : vs) {
    auto m = x.ms; std::sort(m.begin(), m.end());
    printf("%-40s %7.4f %7.4f (%+5.1f%%) |", x.name.c_str(), m[0], m[m.size() / 2], 100.0 * (m[m.size() / 2] / base - 1.0));
    for (float t : x.ms) printf(" %.4f", t);
    printf("\n");
  }
  return 0;
}
