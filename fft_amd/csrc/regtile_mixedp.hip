// Persistent software-pipelined mixed-radix kernels with deferred row blocks (kernel_regtile_mixedp.h): n_fft = 3000 and the larger
// smooth lengths whose tiles leave a CU to one workgroup (2400, 2560, 3072, 3600, 3840).
#include "kernel_regtile_mixedp.h"
#include <atomic>
#include <cstdlib>
namespace sfft {
// P = deferred row blocks.  n_fft = 3000 (60 x 50), interleaved runs on one box at (256,3000,768): one tile per workgroup
// (kernel_regtile_mixed.h) 1.663 ms; P = 0 1.766, 16 1.661, 24 1.507, 26 1.506, 28 1.561 (253 VGPRs), 30 1.628 (spills).
// Round 3, after the exchanges were tidied (fewer live registers; profiles/r03_mixedp_p_sweep.log, one box): 60 x 50 P = 20 1.498,
// 24 1.472, 26 1.441, 28 1.423 (252 VGPRs, no spill), 30 1.428; 64 x 40 P = 16 1.329, 20 1.292, 22 1.291, 24 1.297, 26 1.306;
// 60 x 40 P = 20 1.446, 24 1.438, 26 1.444, 28 1.449, 30 1.507.
#define SFFT_DEFINE_MIXEDP_LAUNCHER(RF_, RS_, P_, S_, XP_)                                                                      \
  template <>                                                                                                          \
  hipError_t launch_regtile_mixedp<RF_, RS_>(const RegtileArgs& a, hipStream_t stream) {                                \
    void (*kern)(const RegtileArgs) = spectre_mix_regtile_mixedp<RF_, RS_, P_, false, S_, XP_>;                         \
    static std::atomic<bool> lds_opt_in[16][2];                                                                         \
    size_t lds = mixedp_lds_total<RF_, RS_, S_>();                                                                      \
    bool tickets = false;                                                                                               \
    if constexpr (((XP_) & 8) != 0 && (S_) > 0) {      /* round 5: dynamic tile order (kernel_tickets.h) where spectre_hip.hip hands over a ticket slice */ \
      if (a.tickets) { kern = spectre_mix_regtile_mixedp<RF_, RS_, P_, false, S_, XP_, true>; lds = mixedp_lds_bytes<RF_, RS_, S_, true>(); tickets = true; } \
    }                                                                                                                   \
    if (a.tickets && !tickets) return hipErrorInvalidValue;   /* (the host only offers tickets to the lengths built with them) */ \
    int dev = 0;                                                                                                        \
    (void)hipGetDevice(&dev);                                                                                           \
    if (dev < 0 || dev >= 16 || !lds_opt_in[dev][tickets]) {                                                            \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return e;                                                                                    \
      if ((e = mixed_check_lds_layout(reinterpret_cast<const void*>(kern))) != hipSuccess) return e;                    \
      if (dev >= 0 && dev < 16) lds_opt_in[dev][tickets] = true;                                                        \
    }                                                                                                                   \
    hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(mixedp_launch_threads<RF_, RS_>()), lds, stream, a);                            \
    return hipGetLastError();                                                                                           \
  }
// Same-box sweep at (256, n, 768), persistent vs one tile per workgroup (tools/mixedp_sweep.py, profiles/r02_mixedp_sweep.log):
// 3000 1.511 vs 1.649 ms, 2560 1.268 vs 1.386, 2400 1.207 vs 1.343; no gain at 3072 (1.599 both), 3600 (1.961 vs 1.917) and 3840 (1.974 vs
// 1.958) in round 2 (round 3: see below).
// Round 4: S = 16 row blocks of the next tile staged by LDS-DMA before the stores + twiddle bases in LDS (kernel_regtile_mixedp.h): (256, 3000, 768)
// 1.42 -> 1.25-1.27 ms; grids of (P, S) on three boxes: profiles/r04_mixedp_stage_*.log.  XP = 8: the deferred loads behind the nine barriers
// of E1 and E2 instead of in one burst in front of E1 (library against library, same box: 3000 -3.0 %, 3600 -3.8 %, 3840 -1.7 %, 3072 -0.4 %;
// 2400 +0.9 % and 2560 +2.8 %: those keep the burst), which also makes room for P = 28 at 60 x 50
SFFT_DEFINE_MIXEDP_LAUNCHER(60, 50, 28, 16, 8)
SFFT_DEFINE_MIXEDP_LAUNCHER(64, 40, 20, 16, 0)
SFFT_DEFINE_MIXEDP_LAUNCHER(60, 40, 24, 16, 0)
// Round 3, with the tidied exchanges the persistent kernel also wins at the three lengths round 2 left to kernel_regtile_mixed.h (one box,
// B = 768000 / n; one tile per workgroup / persistent at P = 12, 16, 20, 24, 28, 32): 3072 = 64 x 48 1.531 / 1.377 1.351 1.297 1.264 1.308
// 1.739 (spills from 28); 3600 = 60 x 60 1.574 / 1.500 1.493 1.449 1.408 1.357 1.584; 3840 = 64 x 60 1.520 / 1.462 1.459 1.409 1.410
// 1.350 1.573.
SFFT_DEFINE_MIXEDP_LAUNCHER(64, 48, 22, 16, 8)
SFFT_DEFINE_MIXEDP_LAUNCHER(60, 60, 28, 16, 8)
SFFT_DEFINE_MIXEDP_LAUNCHER(64, 60, 22, 16, 8)
}  // namespace sfft
