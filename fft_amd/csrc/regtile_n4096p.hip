// n_fft = 4096, persistent software-pipelined kernel (kernel_regtile64p.h)
#include "kernel_regtile64p.h"
#include <atomic>
#include <cstdlib>
namespace sfft {
hipError_t launch_regtile64p(const RegtileArgs& a, bool in_bf16, hipStream_t stream) {
  const bool with_mem = a.mem != nullptr;
  static std::atomic<bool> lds_opt_in[16][3];   // [device]: > 64 KiB of dynamic LDS needs a one-time opt-in (idempotent; the flag only saves the call)
  // PF = row groups whose stores / loads are moved out of the store/load burst to the end of F1 (16 registers each).  Interleaved A/B on
  // one box (tools/p64_ab_bench.hip, profiles/r02_p64_ab_lds_twiddles.log): round-1 kernel 1.859 ms, PF = 1 1.680, PF = 2 1.637,
  // PF = 3 1.621 (248 VGPRs), PF = 4 1.882 (spills).  Before the twiddle vectors moved into LDS (216 instead of 244 VGPRs at PF = 1)
  // only PF = 1 fitted.  SPECTRE_P64_PF overrides (0 .. 3; -1 = PF 0 with scheduling fences).
  static const int pf = [] { const char* e = getenv("SPECTRE_P64_PF"); return e ? atoi(e) : 3; }();
  auto kern = pf == 3 ? spectre_mix_regtile64p<4, 3> : pf == 2 ? spectre_mix_regtile64p<4, 2> : pf == 1 ? spectre_mix_regtile64p<4, 1>
              : pf == -1 ? spectre_mix_regtile64p<4, 0, 0, true> : spectre_mix_regtile64p<4, 0>;
  if (with_mem) kern = spectre_mix_regtile64p<4, 1, 0, true, true>;   // + memory_fft (spectre.py:548-549)
  if (in_bf16) kern = pf == 2 ? spectre_mix_regtile64p<4, 2, 0, true, false, true> : spectre_mix_regtile64p<4, 3, 0, true, false, true>;   // bf16 rows in
  const int variant = in_bf16 ? 2 : with_mem ? 1 : 0;
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
