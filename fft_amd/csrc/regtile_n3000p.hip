// n_fft = 3000 (= 60 x 50), persistent software-pipelined kernel with deferred row blocks (kernel_regtile_mixedp.h)
#include "kernel_regtile_mixedp.h"
#include <atomic>
#include <cstdlib>
namespace sfft {
template <>
hipError_t launch_regtile_mixedp<60, 50>(const RegtileArgs& a, hipStream_t stream) {
  // P = deferred row blocks (of 60): interleaved runs on one box, (256,3000,768): one tile per workgroup (kernel_regtile_mixed.h)
  // 1.663 ms; P = 0 1.766, 16 1.661, 24 1.507, 26 1.506, 28 1.561 (253 VGPRs), 30 1.628 (spills).  SPECTRE_MIXEDP_P overrides (0, 16, 24, 26).
  static const int pp = [] { const char* e = getenv("SPECTRE_MIXEDP_P"); return e ? atoi(e) : 24; }();
  auto kern = pp == 0 ? spectre_mix_regtile_mixedp<60, 50, 0> : pp == 16 ? spectre_mix_regtile_mixedp<60, 50, 16>
              : pp == 26 ? spectre_mix_regtile_mixedp<60, 50, 26> : spectre_mix_regtile_mixedp<60, 50, 24>;
  const int variant = pp == 0 ? 0 : pp == 16 ? 1 : pp == 26 ? 3 : 2;
  static std::atomic<bool> lds_opt_in[16][4];
  const size_t lds = mixed_lds_total<60, 50>();
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !lds_opt_in[dev][variant]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) lds_opt_in[dev][variant] = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(mixed_threads<60, 50>()), lds, stream, a);
  return hipGetLastError();
}
}  // namespace sfft
