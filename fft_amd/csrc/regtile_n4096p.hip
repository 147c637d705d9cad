// n_fft = 4096, persistent software-pipelined kernel (kernel_regtile64p.h)
#include "kernel_regtile64p.h"
#include <atomic>
#include <cstdlib>
namespace sfft {
hipError_t launch_regtile64p(const RegtileArgs& a, hipStream_t stream) {
  static std::atomic<bool> lds_opt_in[16];   // [device]: > 64 KiB of dynamic LDS needs a one-time opt-in (idempotent; the flag only saves the call)
  auto kern = spectre_mix_regtile64p<4>;   // 4 of the 8 row groups of the next tile travel through the exchange image
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !lds_opt_in[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kP64LdsTotal);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) lds_opt_in[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), kP64LdsTotal, stream, a);
  return hipGetLastError();
}
}  // namespace sfft
