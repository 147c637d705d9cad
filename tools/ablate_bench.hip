// ablate_bench — where does the register-tile kernel's time go?  Times spectre_mix_regtile<64> with parts
// switched off (results are then wrong; timing only): bit0 no HBM I/O, bit1 no math, bit2 no LDS exchange.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#include <cstdint>
#include "../fft_amd/csrc/kernel_regtile.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace sfft;

template <int ABL, bool BF, int XV = 1>
void run(const char* name, RegtileArgs a) {
  constexpr int PC = 8;
  auto kern = spectre_mix_regtile<64, 64, BF, BF, 0, ABL, XV>;
  const size_t lds = regtile_lds_total<64, 64, XV>();
  a.tiles_per_row = a.D / (2 * PC); a.n_tiles = a.B * a.tiles_per_row;
  if (a.tpw < 1) a.tpw = 1;
  a.n_wg = 2 * ((a.n_tiles + 2 * a.tpw - 1) / (2 * a.tpw));
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(PC * 64), lds, 0, a);
  CK(hipDeviceSynchronize());
  const int iters = 10;
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(PC * 64), lds, 0, a);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
  printf("%-44s %s XV=%d PC=%d : %7.3f ms   %6.2f us/(16-channel tile)/CU\n", name, BF ? "bf16" : "f32 ", XV, PC, ms, ms * 1e3 * 256 / (a.B * a.D / 16));
}

int main() {
  const int B = 256, N = 4096, D = 768, G = 4, F = N / 2 + 1;
  float *v, *out; float2 *gate, *tw;
  CK(hipMalloc(&v, (size_t)B * N * D * 4)); CK(hipMalloc(&out, (size_t)B * N * D * 4));
  CK(hipMalloc(&gate, (size_t)B * G * F * 8)); CK(hipMalloc(&tw, N * 8));
  {  // random data: constant fills run ~10 % faster (DVFS), which is not the number to chase
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
  a.B = B; a.N_in = N; a.D = D; a.G = G; a.d_g = D / G; a.F = F; a.tiles_per_row = D / 16; a.n_tiles = B * (D / 16);
  a.v_sb = (long long)N * D; a.v_sn = D; a.out_sb = (long long)N * D; a.out_sn = D;
  for (int rep = 0; rep < 3; ++rep) {
    run<0, false, 0>("full kernel", a);
    run<0, false, 1>("full kernel", a);
  }
  run<1, false, 0>("no HBM I/O (math + LDS)", a);
  run<1, false, 1>("no HBM I/O (math + LDS)", a);
  run<3, false, 0>("LDS exchange only", a);
  run<3, false, 1>("LDS exchange only", a);
  run<0, true, 0>("full kernel", a);
  run<0, true, 1>("full kernel", a);
  run<6, false>("I/O only", a);
  run<7, false>("empty", a);
  return 0;
}
