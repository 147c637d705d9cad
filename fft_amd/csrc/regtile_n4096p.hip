// n_fft = 4096, persistent software-pipelined kernel (kernel_regtile64p.h)
#include "kernel_regtile64p.h"
#include <cstdlib>
namespace sfft {
hipError_t launch_regtile64p(const RegtileArgs& a, hipStream_t stream) {
  static const int split = [] { const char* e = getenv("SPECTRE_P64_SPLIT"); return e ? atoi(e) : 4; }();   // tuning aid
  static bool lds_opt_in[16][2] = {};
  auto go = [&](auto kern, int variant) -> hipError_t {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !lds_opt_in[dev][variant]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kP64LdsTotal);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 16) lds_opt_in[dev][variant] = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), kP64LdsTotal, stream, a);
    return hipGetLastError();
  };
  return split > 0 ? go(spectre_mix_regtile64p<4>, 1) : go(spectre_mix_regtile64p<0>, 0);
}
}  // namespace sfft
