// n_fft = 4096, persistent software-pipelined kernel (kernel_regtile64p.h)
#include "kernel_regtile64p.h"
#include <atomic>
namespace sfft {
hipError_t launch_regtile64p(const RegtileArgs& a, bool in_bf16, bool out_bf16, hipStream_t stream) {
  const bool with_mem = a.mem != nullptr;
  static std::atomic<bool> lds_opt_in[16][4];   // [device]: > 64 KiB of dynamic LDS needs a one-time opt-in (idempotent; the flag only saves the call)
  // <SPLIT, PF>: SPLIT row groups of the next tile travel through LDS (LDS-DMA, requested before the stores), PF row groups have their
  // stores / loads moved out of the store/load burst to the end of F1 (16 registers each); the other 8 - SPLIT - PF groups are reloaded
  // behind their own stores.  Interleaved A/B on one box (profiles/r02_p64_ab_waits.log): (4,1) 1.591 ms, (4,2) 1.563, (4,3) 1.548,
  // (3,3) 1.510-1.517, (4,4) 1.651 (spills); round 3 (real plane written by the producer, profiles/r03_p64x_*.log): (3,3) 1.62 -> 1.45.
  auto kern = spectre_mix_regtile64p<3, 3>;
  if (with_mem) kern = spectre_mix_regtile64p<4, 1, true>;                            // + memory_fft (spectre.py:548-549)
  if (in_bf16) kern = spectre_mix_regtile64p<3, 3, false, true>;                      // bf16 rows in
  if (in_bf16 && out_bf16) kern = spectre_mix_regtile64p<3, 3, false, true, true>;    // bf16 rows in and out
  const int variant = in_bf16 ? (out_bf16 ? 3 : 2) : with_mem ? 1 : 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !lds_opt_in[dev][variant]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kP64LdsTotal);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) lds_opt_in[dev][variant] = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), kP64LdsTotal, stream, a);
  return hipGetLastError();
}
}  // namespace sfft
