// n_fft = 4096, persistent software-pipelined kernel (kernel_regtile64p.h)
#include "kernel_regtile64p.h"
#include <atomic>
#include <cstdlib>
namespace sfft {
hipError_t launch_regtile64p(const RegtileArgs& a, bool in_bf16, bool out_bf16, bool burst, hipStream_t stream) {
  const bool with_mem = a.mem != nullptr;
  static std::atomic<bool> lds_opt_in[16][16];   // [device]: > 64 KiB of dynamic LDS needs a one-time opt-in (idempotent; the flag only saves the call)
  // Shipped forms (round 4: the stores of a tile as one burst behind I2's last butterfly and a workgroup barrier, phased I/O, every load request
  // and the deferred stores spread over the arithmetic — profiles/r04_burst_ab.log, r04_p64v_ab_19_23_spread.log, r04_p64v_bf16_spread.log,
  // r04_p64v_mem_spread.log):
  //   fp32 rows                 (SPLIT, PF) = (3, 3)                  -6 ... -7.5 % against the round-3 order
  //   fp32 rows + memory_fft    (4, 1)                                -11.5 %   (spectre.py:548-549)
  //   bf16 rows in              (5, 3): a row group is 16 KiB, nothing is reloaded behind its store; bf16 -> fp32 -5.2 %, bf16 -> bf16 -10 %
  auto kern = spectre_mix_regtile64p<3, 3, false, false, false, true, true>;
  if (with_mem) kern = spectre_mix_regtile64p<4, 1, true, false, false, true, true>;
  if (in_bf16 && !out_bf16) kern = spectre_mix_regtile64p<5, 3, false, true, false, true, true>;
  if (in_bf16 && out_bf16) kern = spectre_mix_regtile64p<5, 3, false, true, true, true, true>;
#ifdef SPECTRE_P64_LEGACY
  // The round-3 order (no burst, no spreads) for A/B runs: built only with -DSPECTRE_P64_LEGACY (tools/build_variant.sh) and selected by
  // SPECTRE_TUNING=1 SPECTRE_P64_BURST=0.  (SPLIT, PF) history: profiles/r02_p64_ab_waits.log, r03_p64x_ab_12.log ... _14.log.
  if (!burst) {
    kern = spectre_mix_regtile64p<4, 2>;
    if (with_mem) kern = spectre_mix_regtile64p<4, 1, true>;
    if (in_bf16) kern = spectre_mix_regtile64p<3, 3, false, true>;
    if (in_bf16 && out_bf16) kern = spectre_mix_regtile64p<3, 3, false, true, true>;
  }
#else
  burst = true;
#endif

  // round 5: DYNAMIC tile tickets per gang (a.tickets = this launch's zeroed slice of the plan's ticket ring; spectre_hip.hip decides):
  // the chip-wide window of open rows shrinks from every batch element to a few (kernel_regtile64p.h, TICKETS)
  const bool tickets = a.tickets != nullptr && burst && !(with_mem && in_bf16);
  if (tickets && !in_bf16) kern = spectre_mix_regtile64p<3, 3, false, false, false, true, true, true>;
  if (tickets && !in_bf16 && with_mem) kern = spectre_mix_regtile64p<4, 1, true, false, false, true, true, true>;   // (round 6: memory_fft on the ticket order as well)
  if (tickets && in_bf16 && !out_bf16) kern = spectre_mix_regtile64p<5, 3, false, true, false, true, true, true>;
  if (tickets && in_bf16 && out_bf16) kern = spectre_mix_regtile64p<5, 3, false, true, true, true, true, true>;

  const int variant = (in_bf16 ? (out_bf16 ? (burst && !with_mem ? 6 : 3) : burst && !with_mem ? 5 : 2) : with_mem ? (burst ? 7 : 1) : burst ? 4 : 0) + (tickets ? 8 : 0);
  const int lds = tickets ? kP64LdsTotalT : kP64LdsTotal;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !lds_opt_in[dev][variant]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 16) lds_opt_in[dev][variant] = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.n_wg), dim3(512), lds, stream, a);
  return hipGetLastError();
}
}  // namespace sfft
