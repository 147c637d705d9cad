// n_fft = 4096, persistent software-pipelined kernel (kernel_regtile64p.h)
#include "kernel_regtile64p.h"
#include <atomic>
#include <cstdlib>
namespace sfft {
hipError_t launch_regtile64p(const RegtileArgs& a, bool in_bf16, bool out_bf16, bool burst, hipStream_t stream) {
  const bool with_mem = a.mem != nullptr;
  static std::atomic<bool> lds_opt_in[16][16];   // [device]: > 64 KiB of dynamic LDS needs a one-time opt-in (idempotent; the flag only saves the call)
  // <SPLIT, PF>: SPLIT row groups of the next tile travel through LDS (LDS-DMA, requested before the stores), PF row groups have their
  // stores / loads moved out of the store/load burst to the end of F1 (16 registers each); the other 8 - SPLIT - PF groups are reloaded
  // behind their own stores.  Interleaved A/B on one box (profiles/r02_p64_ab_waits.log): (4,1) 1.591 ms, (4,2) 1.563, (4,3) 1.548,
  // (3,3) 1.510-1.517, (4,4) 1.651 (spills).  Round 3 (real plane written by the producer, twiddles in scaled form), two boxes
  // (profiles/r03_p64x_ab_12.log, _13.log): (3,3) 1.595 / 1.457, (4,3) 1.591 / 1.455, (4,2) 1.544 / 1.402, (3,2) 1.609 / 1.464, (4,1) 1.612 / 1.466,
  // (2,3) 1.618 / 1.471 — the round-2 kernel on the same boxes 1.602 / -.  bf16 rows: (8,0) = every load through the image.
  auto kern = spectre_mix_regtile64p<4, 2>;                                           // fp32: 4 groups by LDS-DMA, 2 deferred, 2 behind their stores
  // round 4: the stores of a tile as one burst behind I2's last butterfly and a workgroup barrier: -4.0 ... -4.6 % for fp32 rows
  // (profiles/r04_burst_ab.log); bf16 rows +-0.3 %, memory_fft +0.5 %: those keep the round-3 order
  // ... and every load request and the deferred stores spread over the arithmetic, (SPLIT, PF) = (3, 3): another -5.7 ... -7.5 % on three boxes
  // (tools/p64v_bench.hip batches 19-23, profiles/r04_p64v_ab_19_23_spread.log)
  if (burst && !with_mem && !in_bf16) kern = spectre_mix_regtile64p<3, 3, false, false, false, true, true>;
  if (with_mem) kern = spectre_mix_regtile64p<4, 1, true>;                            // + memory_fft (spectre.py:548-549)
  if (with_mem && burst && !in_bf16) kern = spectre_mix_regtile64p<4, 1, true, false, false, true, true>;   // ... phased order + spread requests: -11.5 %
  // bf16 rows in: a row group is 16 KiB, so up to all eight groups of the next tile fit the image — measured on one box
  // (profiles/r03_p64x_ab_14.log, bf16 -> f32 / bf16 -> bf16): (3,3) 1.476 / 1.329 ms, (4,2) 1.481 / 1.329, (6,2) 1.485 / 1.361,
  // (7,1) 1.509 / 1.383, (6,0) 1.528 / 1.369, (8,0) 1.566 / 1.435: requesting everything early does NOT pay, the deferred stores do
  if (in_bf16) kern = spectre_mix_regtile64p<3, 3, false, true>;
  if (in_bf16 && out_bf16) kern = spectre_mix_regtile64p<3, 3, false, true, true>;
  // bf16 rows in / fp32 rows out (BASELINE config 2 read literally): phased I/O, pairs instead of gangs of four, 2 groups through LDS and
  // 3 deferred: -7.1 ... -7.6 % against the round-3 form (tools/p64v_bench.hip ... bf16, profiles/r04_p64v_bf16_in_ab.log)
  if (burst && in_bf16 && !out_bf16) kern = spectre_mix_regtile64p<2, 3, false, true, false, true>;
  // ... third session of round 4: with every load request and the deferred stores spread over the arithmetic, five groups through LDS and
  // three deferred (nothing reloaded behind its store): bf16 -> fp32 -5.2 %, bf16 -> bf16 (phased order, gangs of four) -10 %
  if (burst && !with_mem && in_bf16 && !out_bf16) kern = spectre_mix_regtile64p<5, 3, false, true, false, true, true>;
  if (burst && !with_mem && in_bf16 && out_bf16) kern = spectre_mix_regtile64p<5, 3, false, true, true, true, true>;

  // round 5: DYNAMIC tile tickets per gang (a.tickets = this launch's zeroed slice of the plan's ticket ring; spectre_hip.hip decides):
  // the chip-wide window of open rows shrinks from every batch element to a few (kernel_regtile64p.h, TICKETS)
  const bool tickets = a.tickets != nullptr && burst && !with_mem;
  if (tickets && !in_bf16) kern = spectre_mix_regtile64p<3, 3, false, false, false, true, true, true>;
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
