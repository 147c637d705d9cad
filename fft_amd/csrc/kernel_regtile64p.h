// kernel_regtile64p.h — persistent, software-pipelined spectral mix for n_fft = 4096 = 64 x 64 on gfx950 (every 16-channel tile
// inside one gate group, 16-byte aligned fp32 rows or 8-byte aligned bf16 rows in, fp32 or (with bf16 in) bf16 rows out; optional memory_fft; ANY sequence length: rows beyond N_in are the buffer
// instructions' out-of-range case — loads return 0 = rfft's zero padding (spectre.py:506), stores are dropped (spectre.py:553) —
// so a padded sequence costs exactly what a full one costs, without a single predicate).
//
// Same mathematics and the same register-tile plan as kernel_regtile.h (one workgroup owns 16 channels x 4096 rows; F1 ->
// twiddle -> E1 -> F2 -> gate -> I1 -> E2 -> conj twiddle -> I2; replaces /root/reference/spectre.py:506 + :542-553), built
// around what the measurements showed (profiles/r02_*, profiles/r03_p64x_*):
//
//  * with one workgroup per tile all 256 CUs run in lock-step — everybody loads, then everybody computes — and that state is an
//    attractor, so HBM idles while the chip computes.  Loads of tile t+1 therefore have to be in flight while tile t is still being
//    computed on the SAME CU (persistent workgroups; they do drift apart);
//  * a wave's stores and loads retire through one in-order counter (vmcnt): loads issued behind the stores of the previous
//    tile cannot be consumed before those stores are acknowledged.  The next tile's first SPLIT row groups are therefore requested
//    BEFORE the stores, as LDS-DMA (buffer_load_dwordx4 ... lds, no VGPR needed) into the exchange image, which is idle between the
//    last exchange of tile t and the first exchange of tile t+1.  Every lane reads back exactly the 16 bytes it requested, so the
//    staging needs no barrier of its own;
//  * PF row groups are deferred: their results stay in 16 PF registers through F1 of the next tile, are stored at the end of F1, and
//    the same registers then prefetch those groups of the tile after — that traffic travels while the CU exchanges and multiplies;
//    the remaining 8 - SPLIT - PF groups are loaded straight into the registers that the stores of I2 have just released;
//  * for the same vmcnt reason nothing that is needed "now" may be a global load: the twiddle vectors live in LDS (written once per
//    workgroup), the gate bins are requested a tile ahead and committed to LDS as late as possible;
//  * 16-byte global accesses: a lane moves the 4 channels (2 packed sequences) of one row; v_permlane16_swap hands the
//    second sequence to the partner lane (lane ^ 16) and receives the partner's row of this lane's sequence, so a lane still
//    owns ONE sequence.  The lane <-> (sequence, row class) map keeps the LDS image layout conflict-free (write banks 4p + rc,
//    b128 read groups distinct: SQ_LDS_BANK_CONFLICT = 0);
//  * (round 3) the exchanges move one float plane at a time through the 136-KiB image: write re | barrier | read re | barrier |
//    write im | barrier | read im.  The LDS accepts ~80 B/clk of scattered writes, so the two write phases are ~1700 cycles each during
//    which no wave computes (per-wave s_memtime stamps, profiles/r03_p64x_timeline_loads_stores.log).  The REAL plane is now written by the producer,
//    column by column, while it computes the next column (the last butterfly stage of F1 / of the middle phase yields 8 finished
//    positions at a time), and the barrier that frees the image sits in front of that stage instead of behind the previous read:
//    1.62 -> 1.45 ms on one box;
//  * requests are issued as early and as bunched as the registers allow, and pairs of workgroups walk through adjacent tiles: a
//    64-byte row segment is half an L2 line, the L2 fetches whole lines, and the neighbour's request a few microseconds later hits;
//  * (round 4) STORE BURST.  The L2 takes a half-line store at two thirds of the rate of a whole-line one — unless the two halves of a
//    line arrive within about a microsecond of each other (tools/store_lab.hip: both halves from one CU back to back 0.66 ms, two
//    workgroups free-running 0.94 / 0.68, the same two meeting once per tile 0.58, whole lines 0.61 / 0.49).  So in the fp32 kernel every
//    butterfly of I2's last stage runs first, the eight waves meet at an LDS-only barrier, and all stores of the tile leave back to back
//    (BURST): the pair's two 192-KiB bursts overlap in time far more often than stores that trickle out between butterflies.  Same
//    box, interleaved, through the library (tools/burst_ab.py, profiles/r04_burst_ab.log): 1.551 -> 1.480 ms and 1.455 -> 1.392 ms
//    (-4.0 ... -4.6 %).  PHASED I/O on top: one more LDS-only barrier between the burst and the reloads behind it, and one between the
//    deferred stores and the deferred loads of the quiet part, so that a CU's eight waves never mix the two directions inside a burst:
//    -6.3 ... -6.5 % in all against the round-3 order (tools/p64v_bench.hip batches 10-12: burst alone -2.2 %, barrier alone -0.7 %,
//    barrier + burst -4.0 ... -4.6 %, + barrier in front of the reloads -5.4 ... -6.0 %, + barrier between the deferred stores and loads
//    -6.3 ... -6.4 %; a barrier in front of the deferred stores or of the gate fetch, the LDS-DMA requests moved in front of the burst,
//    an s_sleep between burst and reloads, other (SPLIT, PF): all worse).  Letting the two workgroups of a pair MEET in front of the burst as well (device-scope counter, scalar polling;
//    tools/p64v.h SYNCP) adds nothing measurable on top in the library (-0.3 ... +0.2 %) and is not shipped; bf16 rows and memory_fft do
//    not gain from the burst (+-0.3 %, +0.5 %) and keep the round-3 order.
//
// Thread <-> data:  lane = (pp = lane & 3, rcl = (lane >> 2) & 3, h = (lane >> 4) & 1, rch = lane >> 5);
//   sequence p = 2 pp + h, team index u = rcl + 4 rch + 8 wave  (n2 in F1 / I2, k1 in the middle phase);
//   register position j = 8 g + e  <->  row n1 = g + 8 e  (both when loading and when storing).
//
// The s_waitcnt vmcnt(N) that guards the LDS-DMA landing slots is counted by hand (p64_younger below); tools/isa_lint.py recounts
// it in the ISA of every shipped instantiation at build time (fft_amd/build.py) and fails the build on a mismatch.
#pragma once
#include "kernel_regtile.h"

namespace sfft {

constexpr int kP64ImageBytes = regtile_image_bytes<64, 64, 1>();
// LDS: exchange image | half-spectrum gate | the two twiddle vectors of every team index u (W^(u j), W^(8 u j), j = 1..7): 64 x 14 x 8 B.
// The twiddles are read twice per tile; as global loads they would queue (one in-order vmcnt) behind the LDS-DMA requests of the next
// tile and behind the last stores.  From LDS they cost 7 ds_read_b128 and no vmcnt.
constexpr int kP64TwOff = (regtile_lds_total<64, 64, 1>() + 15) & ~15;
constexpr int kP64LdsTotal = kP64TwOff + 64 * 14 * 8;
static_assert(kP64LdsTotal <= 160 * 1024, "LDS budget");

// TICKETS (round 5): the launch's slice of the plan's ticket ring (RegtileArgs::tickets; UNCACHED device memory — the counter is driven
// by scalar atomics, which carry no scope bits: only memory the L2 does not keep is coherent between XCDs for them), in 32-bit words:
//   [0]                      the chip-wide ticket counter
//   [1]                      how many tiles have been claimed so far (a workgroup that runs out of tickets and finds n_tiles here is done
//                            without looking at the claim bits: 9 us -> 3 us at the end of every launch)
//   [kP64TkBox + 8 g ...]    mailbox of gang g: 8 slots, (tag << 24) | ticket, tag = sequence % 255 + 1 (never 0 = the reset state)
//   [kP64TkClaim ...]        one claim bit per TILE (tile index = GANG * ticket + member)
// and two LDS words behind everything else (the next-but-one tile on its way from wave 0 to the other waves).
constexpr int kP64TkBox = 16, kP64TkClaim = 2048, kP64TkSliceWords = 16384;
constexpr unsigned kP64TkEnd = 0xffffffu;         // "no more tickets" in a mailbox slot
constexpr int kP64LdsTotalT = kP64LdsTotal + 16;
static_assert(kP64LdsTotalT <= 160 * 1024, "LDS budget");
constexpr int p64_ticket_capacity() { return (kP64TkSliceWords - kP64TkClaim) * 32; }      // tiles one slice can keep claim bits for

typedef unsigned int p64_u32x4 __attribute__((ext_vector_type(4)));
constexpr int kP64RsrcFlags = 0x00020000;        // raw buffer, 32-bit data format (gfx90a / gfx942 / gfx950 dword 3)

__device__ __forceinline__ void lane16_swap(float& x, float& y) {   // x of the odd 16-lane rows <-> y of the even rows
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}

// The 8 x 8 in-register transforms of fft_regs.h with a scheduling fence after every radix-8 butterfly: hipcc otherwise interleaves
// all eight butterflies of a stage (up to 120 temporaries on top of the 128 data registers), and with the deferred-result registers
// on top it parks those in scratch.  pin8<BASE, STRIDE>: the eight values z[BASE + STRIDE j] have to exist in registers HERE.
// sched_barrier only fences the machine scheduler; the IR-level sinking pass still splits a butterfly and leaves half-finished sums
// alive until their first use hundreds of instructions later (that, not the schedule, is where 250-register peaks come from).
template <int BASE, int STRIDE>
__device__ __forceinline__ void pin8(float2 (&z)[64]) {
  asm volatile("" : "+v"(z[BASE].x), "+v"(z[BASE].y), "+v"(z[BASE + STRIDE].x), "+v"(z[BASE + STRIDE].y),
                    "+v"(z[BASE + 2 * STRIDE].x), "+v"(z[BASE + 2 * STRIDE].y), "+v"(z[BASE + 3 * STRIDE].x), "+v"(z[BASE + 3 * STRIDE].y),
                    "+v"(z[BASE + 4 * STRIDE].x), "+v"(z[BASE + 4 * STRIDE].y), "+v"(z[BASE + 5 * STRIDE].x), "+v"(z[BASE + 5 * STRIDE].y),
                    "+v"(z[BASE + 6 * STRIDE].x), "+v"(z[BASE + 6 * STRIDE].y), "+v"(z[BASE + 7 * STRIDE].x), "+v"(z[BASE + 7 * STRIDE].y));
}

template <bool INV>
__device__ __forceinline__ void p64_stageA1(float2 (&z)[64]) {      // type A stage 1: radix-8 over q1 (positions 8 q1 + q0); W_64^(q0 ka) is stage 2's
  static_for<0, 8>([&](auto q0c) {
    constexpr int q0 = decltype(q0c)::value;
    bfly_plain<8, INV, q0, 8, 64>(z);
    pin8<q0, 8>(z);
    __builtin_amdgcn_sched_barrier(0);
  });
}

template <bool INV, class CB>
__device__ __forceinline__ void p64_stageA1_cb(float2 (&z)[64], CB cb) {   // ... with a call-back behind every butterfly (SPREAD: one share of the LDS-DMA requests)
  static_for<0, 8>([&](auto q0c) {
    constexpr int q0 = decltype(q0c)::value;
    bfly_plain<8, INV, q0, 8, 64>(z);
    pin8<q0, 8>(z);
    __builtin_amdgcn_sched_barrier(0);
    cb(q0c);
    __builtin_amdgcn_sched_barrier(0);
  });
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier, and hipcc implements the
// fence with s_waitcnt vmcnt(0): every barrier of the exchanges would drain the deferred stores and the prefetches that are meant to
// travel DURING the exchanges.  Nothing that crosses waves goes through global memory here (a lane reads back only what its own wave
// requested by LDS-DMA, after its own vmcnt wait), so the LDS counter is all a barrier has to wait for.
__device__ __forceinline__ void p64_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// workgroups that walk through adjacent tiles in step (launch: n_wg is a multiple of it): the 2 that share a 128-byte line of fp32 rows,
// the 4 that share a line of bf16 rows.  bf16 rows in / fp32 rows out with the round-4 phased I/O: 2 — the pair that matters is the one
// whose half-line STORES meet (gang 4: 1.466 ms, gang 2: 1.435 on one box; profiles/r04_p64v_bf16_in_ab.log) — until the requests were spread
// over the arithmetic: with (5, 3) + SPREAD the gang of four is ahead again (1.352 against 1.374 ms, profiles/r04_p64v_bf16_spread.log), so
// every bf16 variant walks in fours
constexpr int p64_gang(bool in_bf16, bool out_bf16, bool burst) { (void)burst; return in_bf16 || out_bf16 ? 4 : 2; }

// One exchange = position j of thread (p, u) -> image row j, column (p, u); thread (p, u) then reads row u, slots 0..63.  One float
// plane at a time (the tile is 256 KiB, the image 136 KiB).  The scattered dword writes are ds_write2st64_b32 (the LDS takes a store's
// address and data registers at 2 cycles per dword: 6 cycles for two dwords instead of 8 for two ds_write_b32); the instruction's two
// 8-bit offsets count units of 256 bytes, and rows 8 apart are 8 * 2176 = 68 * 256 bytes apart.
//
// p64_write_col<KA, IM>: the real (imaginary) parts of positions KA + 8 kb, kb = 0..7 -> rows KA + 8 kb: four ds_write2st64_b32 with two
// opaque base addresses (kb < 4, kb >= 4: the offsets reach 3 * 68 units).  Called by the PRODUCER for the real plane right after the
// butterfly that finished those eight positions, so the writes travel while the next butterfly runs.
template <int KA, bool IM>
__device__ __forceinline__ void p64_write_col(float2 (&z)[64], float* img, int p, int u) {
  constexpr int RW = 8 * 68, PS = 68;              // image row / column strides in floats (16-byte layout of kernel_regtile.h)
  typedef __attribute__((address_space(3))) float lds_float;
  lds_float* lo = (lds_float*)(img + p * PS + u) + KA * RW;
  lds_float* hi = lo + 32 * RW;
  asm volatile("" : "+v"(lo), "+v"(hi));           // keep them apart: base + 16-bit offset would fold them back into one ds_write_b32 each
  if constexpr (IM) {
    lo[0] = z[KA].y;            lo[8 * RW] = z[KA + 8].y;
    lo[16 * RW] = z[KA + 16].y; lo[24 * RW] = z[KA + 24].y;
    hi[0] = z[KA + 32].y;       hi[8 * RW] = z[KA + 40].y;
    hi[16 * RW] = z[KA + 48].y; hi[24 * RW] = z[KA + 56].y;
  } else {
    lo[0] = z[KA].x;            lo[8 * RW] = z[KA + 8].x;
    lo[16 * RW] = z[KA + 16].x; lo[24 * RW] = z[KA + 24].x;
    hi[0] = z[KA + 32].x;       hi[8 * RW] = z[KA + 40].x;
    hi[16 * RW] = z[KA + 48].x; hi[24 * RW] = z[KA + 56].x;
  }
}
// The rest of an exchange, entered with the real plane already written by every wave's producer code:
//   barrier | read re | barrier | write im | barrier | read im [| barrier].
// The chunk order of the reads (slots 0-3, 8-11, ..., then 4-7, 12-15, ...) is the order in which the next stage's first butterflies
// consume them.  LAST_BARRIER = false leaves the image busy: the caller puts the barrier in front of its next write.
// CB: called behind each of the three barriers (SPREAD: a share of the deferred loads — a wave that has just passed a barrier of an exchange
// waits for the LDS anyway, so the time a request spends in the issue stage there is free)
struct P64NoCb { template <class T> __device__ __forceinline__ void operator()(T) const {} };
template <bool LAST_BARRIER, class CB = P64NoCb>
__device__ __forceinline__ void p64_exchange_rest(float2 (&z)[64], float* img, int p, int u, CB cb = CB{}) {
  constexpr int RW = 8 * 68, PS = 68;
  const float* rd = img + u * RW + p * PS;
  auto read_plane = [&](auto is_im) {
    static_for<0, 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value, m = 4 * ((i / 8) + 2 * (i % 8));
      const float4 v = *reinterpret_cast<const float4*>(rd + m);
      if constexpr (decltype(is_im)::value) { z[m].y = v.x; z[m + 1].y = v.y; z[m + 2].y = v.z; z[m + 3].y = v.w; }
      else { z[m].x = v.x; z[m + 1].x = v.y; z[m + 2].x = v.z; z[m + 3].x = v.w; }
    });
  };
  p64_barrier();
  cb(std::integral_constant<int, 0>{});
  read_plane(std::false_type{});
  p64_barrier();
  cb(std::integral_constant<int, 1>{});
  static_for<0, 8>([&](auto cc) { p64_write_col<decltype(cc)::value, true>(z, img, p, u); });
  p64_barrier();
  cb(std::integral_constant<int, 2>{});
  read_plane(std::true_type{});
  if constexpr (LAST_BARRIER) p64_barrier();       // image free again
}

// VMEM instructions a wave issues between its last LDS-DMA request of a burst and the s_waitcnt that guards the landing slots at the
// top of the next tile (completion is in order, so vmcnt(N) with N = that count means "everything up to and including the LDS-DMA has
// landed").  Steady state: per reloaded group 4 stores + 4 loads, per LDS-staged group 4 stores, the 5 gate loads.  First tile (the
// prologue): the 4 loads of every register-loaded group and the 5 gate loads.  tools/isa_lint.py checks both against the ISA.
template <int SPLIT, int PF> constexpr int p64_younger() { return 8 * (8 - PF - SPLIT) + 4 * SPLIT + 5; }
template <int SPLIT> constexpr int p64_younger_first() { return 4 * (8 - SPLIT) + 5; }

// SPLIT = row groups (of 8) of the next tile that travel through LDS.
// PF    = row groups whose I/O is moved out of the store/load burst into the exchange / middle phase, when this CU has no other
//         memory traffic in flight (3 in the library: 245 VGPRs; 4 spills).
// IN_BF16 = bf16 rows in (spectre.py's activations under autocast), fp32 arithmetic: a lane still moves the 4 channels of a row —
//         8 bytes, two packed dwords = its two sequences — so the lane map, the swap and everything after it are the fp32 kernel's;
//         only the staging differs: a DMA instruction fetches 32 whole 32-byte row segments (16 bytes per lane, lane = (row, half)), and
//         every lane reads its 8 bytes back out of its wave's slot (a permutation of 512 contiguous bytes: conflict-free).  A bf16 row
//         group is 16 KiB, so ALL EIGHT groups of the next tile fit the image (SPLIT = 8, PF = 0): every load of a tile is requested
//         right behind E2 of the previous one, none behind its stores (round 3: the fp32 kernel's late groups cost a memory latency
//         per tile — same-box ablation: loads only 1.11 ms, stores only 0.90, neither 0.84).
// OUT_BF16 = bf16 rows out (round to nearest even, like every other kernel here): a lane stores its 4 channels as 8 bytes.
// BURST   = the stores of a tile leave back to back behind I2's last butterfly and a workgroup barrier (header comment)
// SPREAD  = (round 4, with BURST) every LOAD request and the deferred stores are issued one at a time between butterflies instead
//         of in bursts: a wave sits in the issue stage of a load until the memory pipeline has taken it (~40 clocks per 1-KiB request with the
//         chip's read rate saturated: phase times in profiles/r04_p64v_phase_times.log), and a burst of 16 keeps it there while its butterflies
//         wait.  The LDS-DMA requests go behind the eight twiddle rows, the eight butterflies of I2's first stage and the eight groups of its
//         last stage (24 shares); the deferred loads behind the three barriers of E1 (half of them: -1.5 ... -2.9 % on three boxes; ALL of
//         them there +1 %, the gaps of E1 and E2 -0.7 %) and the eight groups of the middle phase; the deferred stores behind the eight
//         columns of F1's second stage.  The store burst stays a burst (its two halves have to meet in the L2) and the reloads stay behind it.
//         Same box, one process (tools/p64v_bench.hip batches 19-23, three boxes): deferred loads spread -2.8 %, LDS-DMA spread -1.7 %,
//         both -4.8 %, + deferred stores -5.3 %, (SPLIT, PF) = (3, 3) instead of (4, 2) on top: -5.7 ... -7.5 % against the phased order.
//         (Stores and loads mixed in the middle phase: +4 %; groups prefetched into spare registers instead of reloaded: +0.4 ... +2 %.)
//         bf16 rows in (a row group is 16 KiB): (SPLIT, PF) = (5, 3) — nothing is reloaded behind its store — with the phased order and the
//         three spreads: bf16 -> fp32 1.385 -> 1.312 ms (-5.2 %; (2,3) spread -2.6 %), bf16 -> bf16 1.319 -> 1.187 ms (-10 %) on one box
//         (profiles/r04_p64v_bf16_spread.log).  fp32 rows + memory_fft, (4, 1): phased order -7.7 %, + the three spreads -11.5 % (2.025 -> 1.793 ms,
//         profiles/r04_p64v_mem_spread.log).
// TICKETS = (round 5) DYNAMIC tile order.  With the static map every gang owns its own region of the tensor, so at any time the chip has
//         all 256 batch elements open (2 x 3.2 GB); a pure copy in this tile shape runs 40 % faster (64-byte tiles; 32-byte tiles 43 %, 128-byte
//         tiles 10 %) when the gangs instead take adjacent tiles from ONE counter in address order — the chip-wide window is then a few
//         batch elements (tools/window_lab.hip, profiles/r05_window_lab_tile_maps.log) — provided the workgroups that share a line keep
//         walking in step: one ticket per GANG (= GANG adjacent tiles = every piece of a row's 128-byte line), not per workgroup (that
//         was round 4's attempt: +0.6 ... +17 %).  The gang's LEADER (member 0) draws the ticket two tiles ahead with a scalar atomic
//         (lgkmcnt: no place in the in-order vmcnt the hand-counted waits look at), publishes it in the gang's mailbox and every member
//         claims its own tile of the ticket by setting the tile's claim bit (atomic or, scalar as well); the requests are issued at the
//         barriers the tile has anyway (each waits for lgkmcnt(0)), one step of the little state machine per barrier, so nothing waits
//         for a round trip.  Correctness never depends on the members of a gang being resident together: a tile is processed by whoever
//         set its claim bit (exactly one workgroup can), a follower whose leader does not publish in time stops following, and every
//         workgroup that runs out of tickets SWEEPS the claim bits for tiles nobody has claimed and processes them one at a time.
//         A ticket whose tile this member did not get (claimed by a sweeper, or beyond the last tile) is a PHANTOM tile: the same
//         instruction stream over empty buffer ranges, so that the gang stays in step.  Harness, same box, one process: -3.3 % with one
//         counter per XCD (profiles/r05_p64v_ab_31_pair_tickets.log).
#ifndef SPECTRE_P64_EARLY1
#define SPECTRE_P64_EARLY1 11
#endif
template <int SPLIT, int PF, bool WITH_MEM = false, bool IN_BF16 = false, bool OUT_BF16 = false, bool BURST = false, bool SPREAD = false, int TICKETS = 0>
__global__ void __launch_bounds__(512, 2) spectre_mix_regtile64p(const RegtileArgs a) {
  constexpr int ESI = IN_BF16 ? 2 : 4, ESO = OUT_BF16 ? 2 : 4;   // bytes per input / output element
  // EARLY1 (round 5, last session): stage 1 of F1 for the DEFERRED groups of the next tile — their rows trade places with this tile's
  // results inside the store burst — runs right behind the burst, in front of the reload / gate requests, instead of at the top of the
  // next tile: arithmetic while the memory pipeline drains the stores.  Why it was tried: phase times of wave 0 (tools/p64v_phases.hip,
  // profiles/r05_p64v_phases_shipped_forms.log) show the bf16 kernels bound by their own instruction stream (1.14 ms with 0.85 ms worth of
  // traffic), and the burst, the request phases and the back edge are where a wave sits in the issue stage.  Same bits.  Harness, static map,
  // three boxes: bf16 -> bf16 -2.2 ... -2.8 %, bf16 -> fp32 -1.9 ... -2.4 %, fp32 -0.2 ... -1.2 %; under the ticket order fp32 +1.4 ... +1.6 %.
  // Through the library (tools/early1_ab.py, profiles/r05_early1_ab_library.log): bf16 -> bf16 static -1.1 %, tickets -0.4 %; bf16 -> fp32
  // static -0.9 % but tickets +1.1 %.  So: bf16 -> bf16 in both orders, fp32 rows out only on the static map.
  // SPECTRE_P64_EARLY1 (compile time, A/B through tools/build_variant.sh): bit 0 = bf16 -> bf16 (both orders), bit 1 = fp32 rows on the static
  // map, bit 2 = fp32 rows on tickets, bit 3 = bf16 -> fp32 on the static map, bit 4 = bf16 -> fp32 on tickets.  Default 11: the static forms
  // take it (whichever order measures faster on a tensor pair still wins, choose_tile_order in spectre_hip.hip), the ticket forms with fp32
  // rows out do not.
  constexpr bool EARLY1 = BURST && SPREAD && !WITH_MEM && PF > 0 &&
                          (IN_BF16 ? (OUT_BF16 ? (SPECTRE_P64_EARLY1 & 1) != 0 : TICKETS ? (SPECTRE_P64_EARLY1 & 16) != 0 : (SPECTRE_P64_EARLY1 & 8) != 0)
                                   : TICKETS ? (SPECTRE_P64_EARLY1 & 4) != 0 : (SPECTRE_P64_EARLY1 & 2) != 0);
  constexpr float inv_n = 1.0f / 4096.0f;
  constexpr int GROUP_SLOT = IN_BF16 ? 2 * 1024 : 4 * 1024;     // bytes of one row group in a wave's landing slots
  static_assert(SPLIT >= 1 && SPLIT <= 8 && SPLIT * GROUP_SLOT * 8 <= kP64ImageBytes, "staging lives in the exchange image");
  static_assert(PF >= 0 && SPLIT + PF <= 8, "row groups: SPLIT through LDS, PF deferred / prefetched in registers, the rest behind their stores");
  static_assert(!SPREAD || BURST, "SPREAD: with the phased order");
  constexpr int GP = 8 - PF;                       // first deferred / prefetched group
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);
  float2* glds = reinterpret_cast<float2*>(smem + kP64ImageBytes);
  float2* twl = reinterpret_cast<float2*>(smem + kP64TwOff);
  for (int i = threadIdx.x; i < 64 * 14; i += 512) {
    const int uu = i / 14, e = i - 14 * uu, j = (e % 7) + 1;
    twl[i] = a.tw[e < 7 ? uu * j : uu * 8 * j];
  }
  __syncthreads();
  auto load_twiddles = [&](float2 (&wa)[8], float2 (&wb)[8], int uu) {
    const float4* t = reinterpret_cast<const float4*>(twl + 14 * uu);
    const float4 q0 = t[0], q1 = t[1], q2 = t[2], q3 = t[3], q4 = t[4], q5 = t[5], q6 = t[6];
    wa[1] = make_float2(q0.x, q0.y); wa[2] = make_float2(q0.z, q0.w); wa[3] = make_float2(q1.x, q1.y); wa[4] = make_float2(q1.z, q1.w);
    wa[5] = make_float2(q2.x, q2.y); wa[6] = make_float2(q2.z, q2.w); wa[7] = make_float2(q3.x, q3.y); wb[1] = make_float2(q3.z, q3.w);
    wb[2] = make_float2(q4.x, q4.y); wb[3] = make_float2(q4.z, q4.w); wb[4] = make_float2(q5.x, q5.y); wb[5] = make_float2(q5.z, q5.w);
    wb[6] = make_float2(q6.x, q6.y); wb[7] = make_float2(q6.z, q6.w);
  };

  // Only threadIdx.x stays live across the tile loop; the lane coordinates are re-derived from an opaque copy per tile
  // (otherwise LICM hoists every per-lane address out of the loop and the allocator spills them).
  const int tid0 = threadIdx.x;
  int lane, pp, h, p, u;
  char* slot;
  auto coords = [&]() {
    int t = tid0;
    asm volatile("" : "+v"(t));
    lane = t & 63;
    pp = lane & 3; h = (lane >> 4) & 1;
    p = 2 * pp + h;
    u = ((lane >> 2) & 3) + 4 * (lane >> 5) + 8 * (t >> 6);
    slot = smem + __builtin_amdgcn_readfirstlane(t >> 6) * (SPLIT * GROUP_SLOT);   // this wave's landing slots
  };
  coords();

  // GANG neighbouring workgroups (same L2) walk through GANG adjacent tiles in step = one 128-byte line per row: the L2 fetches a
  // line once and the neighbours' requests hit (fp32: two 64-byte halves; bf16: four 32-byte quarters)
  constexpr int GANG = p64_gang(IN_BF16, OUT_BF16, BURST);
  const int wg_lin = xcd_contiguous(blockIdx.x, a.n_wg);
  const int pair_base = (wg_lin / GANG) * a.tpw * GANG + (wg_lin % GANG);
  if constexpr (!TICKETS) { if (pair_base >= a.n_tiles) return; }
  // ---- TICKETS: tiles are GANG * ticket + member; a tile variable holds a tile index, -1 (phantom) or -2 (no more tickets)
  [[maybe_unused]] const int tk_member = wg_lin % GANG;
  [[maybe_unused]] unsigned* const tk_cnt = a.tickets;
  [[maybe_unused]] unsigned* const tk_box = a.tickets + kP64TkBox + 8 * (wg_lin / GANG);
  [[maybe_unused]] unsigned* const tk_claim = a.tickets + kP64TkClaim;
  [[maybe_unused]] volatile int* const tk_lds = reinterpret_cast<volatile int*>(smem + kP64LdsTotal);
  [[maybe_unused]] const unsigned tk_total = (unsigned)((a.n_tiles + GANG - 1) / GANG);
  [[maybe_unused]] const bool tk_w0 = __builtin_amdgcn_readfirstlane(tid0 >> 6) == 0;
  [[maybe_unused]] auto tk_tag = [](int seq) -> unsigned { return (unsigned)(seq % 255) + 1u; };
  [[maybe_unused]] int cur_tile = -2, nxt_tile = -2;
  [[maybe_unused]] int tk_seq = 2;                 // sequence number (within the gang's stream) of the ticket being acquired
  [[maybe_unused]] bool tk_follow = true;          // false: the leader stopped publishing in time (or the stream ended): go and sweep
  [[maybe_unused]] bool tk_grace = false;          // sweep: the grace period behind the end of the stream is over

  float2 z[64];
  float4 dfr[PF > 0 ? 4 * PF : 1];                 // deferred results of the previous tile / prefetched rows of the next one
  static_for<0, 4 * PF>([&](auto ic) { dfr[decltype(ic)::value] = make_float4(0.f, 0.f, 0.f, 0.f); });   // (stored into an empty range before the first tile)
  char* obp = nullptr;                             // output tile of the deferred results
  float2 gstage[5];      // the next tile's gate bins on their way to LDS (4097 bins / 512 threads, rounded up; + 1 for the last)

  // lane offset of an LDS-DMA request: fp32 = the lane's own 16 bytes (the register-load offset); bf16 = lane l fetches the 16-byte half
  // l & 1 of the 32-byte segment of row  (l >> 1 & 7) + 8 wave + 512 (l >> 4 & 1) + 1024 (l >> 5)   [+ 64 g + 2048 (m >> 1) per instruction]
  auto dma_voff = [&](uint32_t voff, long long sn) -> uint32_t {
    if constexpr (IN_BF16) return (uint32_t)(((long long)(((lane >> 1) & 7) + 8 * (u >> 3) + 512 * ((lane >> 4) & 1) + 1024 * (lane >> 5)) * sn) * ESI + (lane & 1) * 16);
    else return voff;
  };
  auto tile_ptrs = [&](int tile, const char*& vb, char*& ob, const float2*& gp) {
    const int b = tile / a.tiles_per_row, ct = tile - b * a.tiles_per_row;
    vb = reinterpret_cast<const char*>(a.v) + ((size_t)b * a.v_sb + (size_t)ct * 16) * ESI;
    ob = reinterpret_cast<char*>(a.out) + ((size_t)b * a.out_sb + (size_t)ct * 16) * ESO;
    gp = a.gate + ((size_t)b * a.G + (ct * 16) / a.d_g) * a.F;
  };
  // row of this lane in load / store instruction (g, m):  u + 512 h + 64 g + 1024 m; addresses = workgroup-uniform base of the
  // instruction (SGPRs) + one 32-bit lane offset (spectre_hip.hip bounds 4095 * row stride * 4 + 64 below 2^31)
  // Buffer resources: base = the tile's first row, num_records = the bytes of its rows that exist (a.rows_in input rows are read, the
  // rest are rfft's zero padding; a.rows_out output rows are written).  The range check covers the VGPR offset (lane offset + row-block
  // offset; the SGPR offset operand is not checked on gfx9), so both go there.
  // live = false: an empty range.  Every request of the tile loop is issued UNCONDITIONALLY — after the last tile (and, for the deferred
  // stores, before the first) with an empty range, which costs nothing: hipcc computes its s_waitcnt vmcnt(N) from the requests that are
  // GUARANTEED to be younger than the one waited for, so a request inside `if (more)` does not count, N comes out too small, and a wait
  // for a prefetched register early in I2 turned into a wait for the LDS-DMA issued just before it (a full HBM round trip per tile).
  auto rsrc_in = [&](const char* vb, long long sn, bool live = true) {
    const int rows = live ? a.rows_in : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vb), 0, (int)((long long)rows * sn * ESI), kP64RsrcFlags);
  };
  auto rsrc_out = [&](char* ob, long long sn, bool live = true) {
    const int rows = live ? a.rows_out : 0;
    return __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)((long long)rows * sn * ESO), kP64RsrcFlags);
  };
  auto unpack_lo = [](uint32_t d) { return make_float2(__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)); };   // two bf16 -> (re, im)
  auto load_group = [&](__amdgpu_buffer_rsrc_t rs, uint32_t voff, long long sn, auto gc) {       // straight into the registers of group g
    constexpr int g = decltype(gc)::value;
    static_for<0, 4>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if constexpr (IN_BF16) {
        const rt_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + (uint32_t)((64 * g + 1024 * m) * sn * ESI), 0, 0);
        z[8 * g + 2 * m] = unpack_lo(t.x);
        z[8 * g + 2 * m + 1] = unpack_lo(t.y);
      } else {
        const p64_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (uint32_t)((64 * g + 1024 * m) * sn * 4), 0, 0);
        z[8 * g + 2 * m] = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
        z[8 * g + 2 * m + 1] = make_float2(__uint_as_float(t.z), __uint_as_float(t.w));
      }
    });
  };
  // 1 KiB per instruction, lane l's 16 bytes at slot + 16 l.  fp32: four instructions per group (m = 0..3); bf16: two (m >> 1 = 0, 1),
  // each with the rows of both h and of m & 1 (dma_voff).
  auto dma_group = [&](__amdgpu_buffer_rsrc_t rs, uint32_t voff, long long sn, auto gc) {        // into this wave's LDS slots
    constexpr int g = decltype(gc)::value;
    if constexpr (IN_BF16) {
      static_for<0, 2>([&](auto mc) {
        constexpr int mh = decltype(mc)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(slot + (2 * g + mh) * 1024), 16,
                                                 voff + (uint32_t)((64 * g + 2048 * mh) * sn * ESI), 0, 0, 0);
      });
    } else {
      static_for<0, 4>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(slot + (4 * g + m) * 1024), 16,
                                                 voff + (uint32_t)((64 * g + 1024 * m) * sn * 4), 0, 0, 0);
      });
    }
  };
  auto store16 = [&](__amdgpu_buffer_rsrc_t rs, uint32_t off, const float4 v) {   // this lane's 4 channels of one row
    if constexpr (OUT_BF16) {
      rt_u32x2 t;
      t.x = f32x2_to_bf16x2_rne(v.x, v.y); t.y = f32x2_to_bf16x2_rne(v.z, v.w);
      __builtin_amdgcn_raw_buffer_store_b64(t, rs, off, 0, 0);
    } else {
      p64_u32x4 t;
      t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y); t.z = __float_as_uint(v.z); t.w = __float_as_uint(v.w);
      __builtin_amdgcn_raw_buffer_store_b128(t, rs, off, 0, 0);
    }
  };
  auto read_group = [&](auto gc) {                                       // this lane's bytes back out of the slot
    constexpr int g = decltype(gc)::value;
    static_for<0, 4>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if constexpr (IN_BF16) {
        // DMA lane 2 (rcl + 4 rch + 8 h + 16 (m & 1)) + (pp >> 1) of instruction (g, m >> 1), half pp & 1 of its 16 bytes
        const rt_u32x2 t = *reinterpret_cast<const rt_u32x2*>(slot + (2 * g + (m >> 1)) * 1024 + ((((lane >> 2) & 3) + 4 * (lane >> 5)) + 8 * h + 16 * (m & 1)) * 32 + pp * 8);
        z[8 * g + 2 * m] = unpack_lo(t.x);
        z[8 * g + 2 * m + 1] = unpack_lo(t.y);
      } else {
        const float4 t = *reinterpret_cast<const float4*>(slot + (4 * g + m) * 1024 + lane * 16);
        z[8 * g + 2 * m] = make_float2(t.x, t.y);
        z[8 * g + 2 * m + 1] = make_float2(t.z, t.w);
      }
    });
  };
  auto swap_group = [&](auto gc) {      // rows (g + 16 m, g + 16 m + 8) of sequences (2pp, 2pp+1)  <->  both rows of sequence p
    constexpr int g = decltype(gc)::value;
    static_for<0, 4>([&](auto mc) {
      constexpr int j = 8 * g + 2 * decltype(mc)::value;
      lane16_swap(z[j].x, z[j + 1].x);
      lane16_swap(z[j].y, z[j + 1].y);
    });
  };
  // gate_fetch only REQUESTS the bins: the staging registers cross the loop's back edge, and anything computed from them before it
  // (the edge rule, the conj, the 1/N scale) would have to wait for the loads right there, at the end of the burst — i.e. for every
  // store of the tile (one in-order vmcnt).  All arithmetic happens in gate_commit, a phase and a half later.
  auto gate_fetch = [&](const float2* gp) {
    static_for<0, 5>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int k = lane + 64 * (u >> 3) + 512 * i;
      // (every lane loads — the lanes beyond bin 2048 re-read it and gate_commit ignores them: a predicated fifth load becomes a
      //  branch with `s_waitcnt vmcnt(0)` behind it, and the waves that skip it would have one request less in flight than the
      //  vmcnt() at the top of the loop counts on)
      gstage[i] = gp[i < 4 ? k : (k <= 2048 ? k : 2048)];
    });
  };
  auto gate_commit = [&]() {
    static_for<0, 5>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int k = lane + 64 * (u >> 3) + 512 * i;
      float2 g = gstage[i];
      asm volatile("" : "+v"(g.x), "+v"(g.y));     // consumed HERE by every wave: the fifth bin's write below is lane-predicated, and a wave
                                                   // that branches around it would carry the pending load into the exchange, where hipcc then
                                                   // protects a reused register with s_waitcnt vmcnt(0) — behind the deferred requests
      if (k == 0 || k == 2048) g.y = 0.f;          // irfft ignores Im(DC), Im(Nyquist) (spectre.py:551)
      if (a.conj_gate) g.y = -g.y;
      if (i < 4 || k <= 2048) glds[k] = make_float2(g.x * inv_n, g.y * inv_n);
    });
  };

  static_assert(!TICKETS || (SPREAD && BURST), "TICKETS: built on the phased order with spread requests");
  // TICKETS: round 0 follows the gang's tickets; every later round processes ONE tile that nobody has claimed (the sweep).  Static map: one round.
  for (int round = 0;; ++round) {
  if constexpr (TICKETS) {
    if (round == 0) {
      // the first two tickets of the gang, synchronously (nothing is in flight yet): the leader draws and publishes, the others read
      if (tid0 == 0) {
        for (int sq = 0; sq < 2; ++sq) {
          unsigned t = kP64TkEnd;
          if (tk_member == 0) {
            t = __hip_atomic_fetch_add(tk_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t >= tk_total) t = kP64TkEnd;
            __hip_atomic_store(tk_box + sq, (tk_tag(sq) << 24) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            for (int i = 0; i < (1 << 18); ++i) {      // (bounded: a leader that is not resident yet must not hang this workgroup)
              const unsigned w = __hip_atomic_load(tk_box + sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if ((w >> 24) == tk_tag(sq)) { t = w & 0xffffffu; break; }
              __builtin_amdgcn_s_sleep(8);
            }
          }
          int tl = -2;
          if (t != kP64TkEnd) {
            const unsigned ti = t * GANG + tk_member;
            tl = -1;
            if (ti < (unsigned)a.n_tiles && !(__hip_atomic_fetch_or(tk_claim + (ti >> 5), 1u << (ti & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (1u << (ti & 31)))) tl = (int)ti;
          }
          if (tl >= 0) __hip_atomic_fetch_add(tk_cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tk_lds[sq] = tl;
          if (tl == -2) { tk_lds[1] = -2; break; }       // the stream has ended (or the leader is silent): claim nothing behind it
        }
      }
      __syncthreads();
      cur_tile = __builtin_amdgcn_readfirstlane(tk_lds[0]); nxt_tile = __builtin_amdgcn_readfirstlane(tk_lds[1]);   // (uniform: SGPRs)
      __syncthreads();
      tk_follow = nxt_tile != -2;
      if (cur_tile == -2) nxt_tile = -2;
    } else {
      if constexpr (TICKETS == 2) break;               // (measurement only, tools/tickets_lab.hip: what the look at the claim bits costs at the end of a launch)
      // sweep: the lowest tile whose claim bit is still clear (every thread looks at its share of the bit words; LDS minimum)
      asm volatile("s_waitcnt vmcnt(0) ; lint: drain" ::: "memory");
      __syncthreads();
      if (tid0 == 0) { tk_lds[0] = 0x7fffffff; tk_lds[1] = (int)__hip_atomic_load(tk_cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(tk_lds[1]) >= a.n_tiles) break;      // every tile has an owner (the usual end of a launch)
      __syncthreads();
      // Only tiles that are certainly nobody's any more: a ticket stays unclaimed for a moment between the leader's draw and each member's
      // claim, and a gang has at most three tickets on their way (this tile, the next, the one after), so while the stream is running the
      // sweep looks below  counter - 3 * gangs.  Once the counter has passed the end, one grace period (~ a tile) later everything counts.
      if (tid0 == 0) tk_lds[1] = (int)__hip_atomic_load(tk_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ONE reading for the whole workgroup:
      __syncthreads();                                                                                           // every wave must take the same way out
      const unsigned tk_now = (unsigned)__builtin_amdgcn_readfirstlane(tk_lds[1]);
      __syncthreads();
      const unsigned n_gangs = (unsigned)(a.n_wg / GANG);
      const bool tk_all = tk_now >= tk_total && tk_grace;
      const long long tk_old = tk_all ? (long long)tk_total : (long long)tk_now - 3LL * n_gangs;
      const int lim = (int)(tk_old <= 0 ? 0 : tk_old * GANG > a.n_tiles ? a.n_tiles : tk_old * GANG);      // tiles below this are fair game
      const int n_words = (lim + 31) >> 5, all_words = (a.n_tiles + 31) >> 5;
      const int sw = n_words ? (int)((long long)wg_lin * n_words / a.n_wg) : 0;   // every workgroup starts looking somewhere else (fewer collisions when many sweep)
      if (tid0 == 0) tk_lds[1] = 0;
      __syncthreads();
      for (int w = tid0; w < all_words; w += 512) {
        unsigned fr = ~__hip_atomic_load(tk_claim + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (32 * w + 32 > a.n_tiles) fr &= (1u << (a.n_tiles - 32 * w)) - 1u;
        if (fr) tk_lds[1] = 1;                       // something is unclaimed somewhere (the usual end of a launch: nothing is)
        if (32 * w + 32 > lim) fr &= 32 * w >= lim ? 0u : (1u << (lim - 32 * w)) - 1u;
        if (fr) atomicMin(const_cast<int*>(tk_lds), 32 * (w >= sw ? w - sw : w - sw + n_words) + (int)__builtin_ctz(fr));
      }
      __syncthreads();
      const int key = __builtin_amdgcn_readfirstlane(tk_lds[0]), any_free = __builtin_amdgcn_readfirstlane(tk_lds[1]);
      __syncthreads();
      if (key == 0x7fffffff) {
        if (tk_all || !any_free) break;              // every tile has an owner: done
        for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(127);       // ~ 25 us: let the members that are on their way claim what is theirs
        if (tk_now >= tk_total) tk_grace = true;
        continue;
      }
      const int cand = 32 * (((key >> 5) + sw) % n_words) + (key & 31);
      if (tid0 == 0) tk_lds[1] = (__hip_atomic_fetch_or(tk_claim + (cand >> 5), 1u << (cand & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (cand & 31)) & 1u;
      __syncthreads();
      const int lost = __builtin_amdgcn_readfirstlane(tk_lds[1]);
      __syncthreads();
      if (lost) continue;                            // somebody else took it in the meantime: look again
      if (tid0 == 0) __hip_atomic_fetch_add(tk_cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cur_tile = cand; nxt_tile = -2;
    }
    if (cur_tile == -2) continue;                    // (round 0 without a single ticket: straight to the sweep)
  }
  [[maybe_unused]] bool prev_live = false;         // TICKETS: dfr holds the deferred results of a real previous tile (they are stored in F1)
  // ---- prologue: request tile 0 the same way every later tile is requested --------------------------------------------
  {
    const char* vb; char* ob; const float2* gp;
    tile_ptrs(TICKETS ? (cur_tile >= 0 ? cur_tile : 0) : pair_base, vb, ob, gp);
    const uint32_t voff = (uint32_t)(((long long)(u + 512 * h) * a.v_sn + 4 * pp) * ESI);
    const __amdgpu_buffer_rsrc_t rs = rsrc_in(vb, a.v_sn, TICKETS ? cur_tile >= 0 : true);
    static_for<0, SPLIT>([&](auto gc) { dma_group(rs, dma_voff(voff, a.v_sn), a.v_sn, gc); });
    asm volatile("" ::: "memory");
    static_for<SPLIT, 8>([&](auto gc) { load_group(rs, voff, a.v_sn, gc); });
    gate_fetch(gp);
  }

  for (int it = 0; TICKETS || it < a.tpw; ++it) {
    const int tile = TICKETS ? (cur_tile >= 0 ? cur_tile : 0) : pair_base + GANG * it;
    if (TICKETS ? cur_tile == -2 : tile >= a.n_tiles) break;                  // workgroup-uniform
    const bool more = TICKETS ? nxt_tile >= 0 : (it + 1 < a.tpw) && (tile + GANG < a.n_tiles);
    [[maybe_unused]] const bool cur_live = TICKETS ? cur_tile >= 0 : true;     // false: a phantom tile (empty ranges: loads return 0, stores are dropped)
    coords();
    // ---- TICKETS: the tile after next.  Wave 0 moves a small state machine one step at each of the barriers the tile has anyway (they wait
    //      for lgkmcnt(0), which is what a scalar request returns through):
    //        leader:   H0 draw (s_atomic_add) | H1 publish (s_atomic_swap into the mailbox) + claim (s_atomic_or) | H2 result -> LDS
    //        follower: H3 read the mailbox (s_load_dword glc) | H4 check the tag (a late leader is waited for, bounded) + claim | H5 result -> LDS
    [[maybe_unused]] unsigned tk_a = 1, tk_b = 0, tk_bit = 0;
    [[maybe_unused]] int tk_tile = -2;
    [[maybe_unused]] unsigned* const tk_slot = tk_box + (tk_seq & 7);
    [[maybe_unused]] const bool tk_lead = tk_w0 && tk_member == 0 && round == 0 && tk_follow, tk_foll = tk_w0 && tk_member != 0 && round == 0 && tk_follow;
    [[maybe_unused]] auto tk_claim_issue = [&](unsigned t) {      // t < tk_total: ask for this member's tile of ticket t
      const unsigned ti = t * GANG + tk_member;
      tk_tile = -1;
      if (ti < (unsigned)a.n_tiles) {
        tk_tile = (int)ti; tk_bit = 1u << (ti & 31); tk_b = tk_bit;
        unsigned* wp = tk_claim + (ti >> 5);
        asm volatile("s_atomic_or %0, %1, 0x0 glc" : "+s"(tk_b) : "s"(wp) : "memory");
      }
    };
    [[maybe_unused]] auto tk_claim_result = [&]() {                 // (behind a wait for lgkmcnt(0)) -> the LDS word the other waves read
      if (tk_tile >= 0) { asm volatile("" : "+s"(tk_b)); if (tk_b & tk_bit) tk_tile = -1; }
      if (tk_tile >= 0) { const unsigned one = 1u; unsigned* cp = tk_cnt + 1; asm volatile("s_atomic_add %0, %1, 0x0" :: "s"(one), "s"(cp) : "memory"); }   // (no return value)
      if (lane == 0) tk_lds[0] = tk_tile;
    };
    if constexpr (TICKETS) { if (tk_lead) asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(tk_a) : "s"(tk_cnt) : "memory"); }   // H0
    long long v_sn = a.v_sn, out_sn = a.out_sn;
    asm volatile("" : "+s"(v_sn), "+s"(out_sn));
    const char* vb; char* ob; const float2* gp;
    tile_ptrs(tile, vb, ob, gp);
    const char* vbn = vb; char* obn = ob; const float2* gpn = gp;
    if (more) tile_ptrs(TICKETS ? nxt_tile : tile + GANG, vbn, obn, gpn);
    const __amdgpu_buffer_rsrc_t rs_next = rsrc_in(vbn, v_sn, more), rs_out = rsrc_out(ob, out_sn, cur_live);

    [[maybe_unused]] const uint32_t pf_ooff = (uint32_t)(((long long)(u + 512 * h) * out_sn + 4 * pp) * ESO);
    [[maybe_unused]] const uint32_t pf_voff = (uint32_t)(((long long)(u + 512 * h) * v_sn + 4 * pp) * ESI);
    [[maybe_unused]] auto pf_store = [&](auto ic) {
      constexpr int g = GP + decltype(ic)::value / 4, m = decltype(ic)::value % 4;
      store16(rsrc_out(obp, out_sn, TICKETS ? prev_live : it > 0), pf_ooff + (uint32_t)((64 * g + 1024 * m) * out_sn * ESO), dfr[decltype(ic)::value]);
    };
    [[maybe_unused]] auto pf_load = [&](auto ic) {
      constexpr int g = GP + decltype(ic)::value / 4, m = decltype(ic)::value % 4;
      if constexpr (IN_BF16) {                       // stays packed (two dwords) until it trades places with the results in I2
        const rt_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs_next, pf_voff + (uint32_t)((64 * g + 1024 * m) * v_sn * ESI), 0, 0);
        dfr[decltype(ic)::value].x = __uint_as_float(t.x); dfr[decltype(ic)::value].y = __uint_as_float(t.y);
      } else {
        const p64_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_next, pf_voff + (uint32_t)((64 * g + 1024 * m) * v_sn * 4), 0, 0);
        dfr[decltype(ic)::value] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
      }
    };

    // ---- F1: 64-point forward transform over n1 (register position 8g + e holds row g + 8e), then W_N^(u*k1) ------------
    //      Stage 1 works group by group, in the order the groups arrive: the deferred groups (prefetched a tile ago) first, then the
    //      LDS-staged ones — only now does the wave wait for the LDS-DMA of the burst it has just left — and the groups reloaded
    //      behind the stores last.
    static_for<0, 8>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int g = i < PF ? GP + i : i - PF;                            // [GP, 8), [0, SPLIT), [SPLIT, GP)
      if constexpr (i == PF) {
        // the tile arrives: completion is in order, so once everything but the requests younger than the last LDS-DMA has retired the
        // staged groups are in the slots (p64_younger; checked against the ISA by tools/isa_lint.py)
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0) ; lint: first" :: "n"(p64_younger_first<SPLIT>()) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) ; lint: steady" :: "n"(p64_younger<SPLIT, PF>()) : "memory");
        static_for<0, SPLIT>([&](auto gc) { read_group(gc); });
      }
      if (!EARLY1 || i >= PF || it == 0) {         // (EARLY1: done behind the previous tile's burst — except for the tile the prologue loaded)
        swap_group(std::integral_constant<int, g>{});
        bfly_plain<8, false, 8 * g, 1, 64>(z);     // over e -> ka at position 8g + ka (W_64^(g ka) is applied by the column butterflies below)
        pin8<8 * g, 1>(z);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    {
      float2 wa[8], wb[8];
      __builtin_amdgcn_sched_barrier(0);           // keep the base loads (and their registers) out of stage 1
      load_twiddles(wa, wb, u);
      // column by column: butterfly over g -> kb at position 8 kb + ka (k1 = position), both twiddle factors W^(u ka) W^(8 u kb), and the
      // real parts straight into the image while the next column is computed
      static_for<0, 8>([&](auto kac) {
        constexpr int ka = decltype(kac)::value;
        bfly_tw<8, false, ka, 8, 1, ka, 64>(z);    // input g still needs W_64^(g ka): scaled form (fft_regs.h)
        static_for<0, 8>([&](auto kbc) {
          constexpr int kb = decltype(kbc)::value, j = 8 * kb + ka;
          if constexpr (ka > 0) z[j] = cmul(z[j], wa[ka]);
          if constexpr (kb > 0) z[j] = cmul(z[j], wb[kb]);
        });
        pin8<ka, 8>(z);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SPREAD) { static_for<ka * (4 * PF) / 8, (ka + 1) * (4 * PF) / 8>([&](auto ic) { pf_store(ic); }); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (ka == 0) p64_barrier();      // every wave has emptied its landing slots (and finished E2's reads of the previous
                                                   // tile): the image may be written
        if constexpr (ka == 0 && TICKETS) {          // H1 (the barrier waited for lgkmcnt(0): the ticket is here)
          if (tk_lead) {
            asm volatile("" : "+s"(tk_a));
            const unsigned t = tk_a < tk_total ? tk_a : kP64TkEnd;
            const unsigned pub = (tk_tag(tk_seq) << 24) | t;
            asm volatile("s_atomic_swap %0, %1, 0x0" :: "s"(pub), "s"(tk_slot) : "memory");      // (no return value: fire and forget)
            if (t != kP64TkEnd) tk_claim_issue(t);
          }
        }
        p64_write_col<ka, false>(z, img, p, u);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    // this tile's gate bins -> LDS.  They were requested behind the previous tile's last stores, so waiting for them means waiting
    // for every store of that tile to be acknowledged: as late as possible (the bins are first read after E1's barriers) — but BEFORE
    // the deferred requests below: hipcc waits for registers that were loaded before the loop's back edge with vmcnt(0), which behind
    // those requests would mean a full HBM round trip at the end of every F1.
    gate_commit();
    __builtin_amdgcn_sched_barrier(0);
    // ---- the quiet part of the tile starts: the deferred results of the previous tile leave, the same row groups of the next tile
    //      are requested into the registers they vacate
    if constexpr (!SPREAD) {
      static_for<0, 4 * PF>([&](auto ic) { pf_store(ic); });
      if constexpr (BURST) p64_barrier();          // PHASED I/O: every wave has issued its deferred stores before any wave requests the next rows
      static_for<0, 4 * PF>([&](auto ic) { pf_load(ic); });
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- E1: position k1 -> image row k1, column (p, u); thread (p, s = u) reads row u, slot n2.  No barrier behind the last read:
    //      the image stays busy until the barrier in front of the middle phase's last stage.
    if constexpr (SPREAD) p64_exchange_rest<false>(z, img, p, u, [&](auto kc) {       // half of the deferred loads in E1's three gaps ...
      constexpr int k = decltype(kc)::value, H = (4 * PF) / 2;
      if constexpr (TICKETS && k == 0) { if (tk_lead) tk_claim_result(); }             // H2
      static_for<k * H / 3, (k + 1) * H / 3>([&](auto ic) { pf_load(ic); });
    });
    else p64_exchange_rest<false>(z, img, p, u);
    if constexpr (TICKETS) { if (tk_foll) asm volatile("s_load_dword %0, %1, 0x0 glc" : "=s"(tk_a) : "s"(tk_slot) : "memory"); }   // H3

    // ---- middle: F2 -> gate -> I1 (kernel_regtile.h; bin of register (ka, kb): k = k1 + 64 k2, k2 = ka + 8 kb) -------------
    {
      const int k1 = u;
      p64_stageA1<false>(z);
      auto fetch_gate = [&](int k2, bool upper) -> float2 {
        float2 g = glds[upper ? 64 * (64 - k2) - k1 : k1 + 64 * k2];      // scaled by 1/N, edges fixed, conj applied
        if (upper) g.y = -g.y;                                          // Hermitian extension above N/2
        return g;
      };
      // the gate bins of one register group are fetched right where they are used (16 registers)
      float2 gcur[8];
      // memory_fft (spectre.py:548-549): row k of the (F, D) complex buffer, this lane's two channels = 16 bytes; one register
      // group (8 bins) at a time, requested right after the previous group has been consumed (L2 / Infinity-Cache resident:
      // 12.6 MB at the headline shape, re-read by every batch element).  These loads sit in the exchange / middle phase, when the
      // CU has no other memory traffic in flight.
      [[maybe_unused]] float4 mcur[WITH_MEM ? 8 : 1];
      [[maybe_unused]] const float* mbase = nullptr;
      if constexpr (WITH_MEM) mbase = a.mem + (size_t)((tile - (tile / a.tiles_per_row) * a.tiles_per_row) * 16 + 2 * p) * 2;
      auto fetch_mem = [&](int k2, bool upper) -> float4 {
        return *reinterpret_cast<const float4*>(mbase + (size_t)(upper ? 64 * (64 - k2) - k1 : k1 + 64 * k2) * a.D * 2);
      };
      static_for<0, 8>([&](auto kbc) {
        constexpr int k2 = 8 * decltype(kbc)::value;
        gcur[decltype(kbc)::value] = fetch_gate(k2, k2 >= 32);
        if constexpr (WITH_MEM) mcur[decltype(kbc)::value] = fetch_mem(k2, k2 >= 32);
      });
      static_for<0, 8>([&](auto kac) {
        constexpr int ka = decltype(kac)::value;
        if constexpr (SPREAD) {                      // ... the other half behind the eight groups of the middle phase
          constexpr int H = (4 * PF) / 2, R = 4 * PF - H;
          static_for<H + ka * R / 8, H + (ka + 1) * R / 8>([&](auto ic) { pf_load(ic); });
          __builtin_amdgcn_sched_barrier(0);
        }
        fftA_stage2_group<8, 8, false, ka>(z);
        static_for<0, 8>([&](auto kbc) {
          constexpr int kb = decltype(kbc)::value, j = 8 * ka + kb, k2 = ka + 8 * kb;
          z[j] = cmul(z[j], gcur[kb]);                                   // spectre.py:545
          if constexpr (WITH_MEM) {                                      // Mf[k] = mem_c[k] + i mem_{c+1}[k] below N/2, conj(mem_c[N-k]) + i conj(mem_{c+1}[N-k]) above
            const float4 m = mcur[kb];
            float2 add;
            if ((k2 == 0 || k2 == 32) && k1 == 0) add = make_float2(m.x, m.z);          // DC, Nyquist: real parts only
            else if (k2 >= 32)                    add = make_float2(m.x + m.w, m.z - m.y);
            else                                  add = make_float2(m.x - m.w, m.y + m.z);
            z[j].x += add.x * inv_n; z[j].y += add.y * inv_n;
          }
        });
        pin8<8 * ka, 1>(z);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ka + 1 < 8)
          static_for<0, 8>([&](auto kbc) {
            constexpr int k2n = ka + 1 + 8 * decltype(kbc)::value;
            gcur[decltype(kbc)::value] = fetch_gate(k2n, k2n >= 32);
            if constexpr (WITH_MEM) mcur[decltype(kbc)::value] = fetch_mem(k2n, k2n >= 32);
          });
        fftB_stage1_group<8, 8, true, ka>(z);
        pin8<8 * ka, 1>(z);
        __builtin_amdgcn_sched_barrier(0);         // keep the gate prefetch one group deep (register budget)
      });
      // type B stage 2: radix-8 over ka (positions 8 ka + n_lo) -> natural order, position n2 = n_lo + 8 n_hi.  Every wave has long
      // finished E1's reads; behind this barrier the image is written again, column by column like in F1.
      if constexpr (TICKETS) {                       // H4: the slot must carry this sequence number; a leader that is behind is waited for
        if (tk_foll) {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tk_a) :: "memory");
          for (int i = 0; i < (1 << 16) && (tk_a >> 24) != tk_tag(tk_seq); ++i) {
            __builtin_amdgcn_s_sleep(8);
            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk_a) : "s"(tk_slot) : "memory");
          }
          if ((tk_a >> 24) == tk_tag(tk_seq) && (tk_a & 0xffffffu) != kP64TkEnd) tk_claim_issue(tk_a & 0xffffffu);
          // (no ticket in time, or the end of the stream: tk_tile stays -2 and this workgroup goes on to the sweep)
        }
      }
      p64_barrier();
      static_for<0, 8>([&](auto nc) {
        constexpr int nlo = decltype(nc)::value;
        fftB_stage2_group<8, 8, true, nlo>(z);
        pin8<nlo, 8>(z);
        __builtin_amdgcn_sched_barrier(0);
        p64_write_col<nlo, false>(z, img, p, u);
        __builtin_amdgcn_sched_barrier(0);
      });
    }

    // ---- E2: position n2 -> image row n2, column (p, k1 = u); thread (p, u) reads row u, slot k1.  The barrier behind the last read
    //      frees the image for the LDS-DMA below.
    if constexpr (TICKETS) p64_exchange_rest<true>(z, img, p, u, [&](auto kc) { if constexpr (decltype(kc)::value == 0) { if (tk_foll) tk_claim_result(); } });   // H5
    else p64_exchange_rest<true>(z, img, p, u);

    // ---- the image is idle until the next F1: let the first row groups of the next tile land in it, and fetch its gate -----
    const uint32_t voff = (uint32_t)(((long long)(u + 512 * h) * v_sn + 4 * pp) * ESI);
    // (the twiddles of I2 are read BEFORE the LDS-DMA is issued: hipcc orders every LDS read behind a pending LDS-DMA with
    //  s_waitcnt vmcnt(0) — the whole HBM round trip of the requests below, at the start of every burst)
    float2 wa[8], wb[8];
    load_twiddles(wa, wb, u);
    [[maybe_unused]] int fut_tile = -2;
    if constexpr (TICKETS) { if (round == 0 && tk_follow) fut_tile = __builtin_amdgcn_readfirstlane(tk_lds[0]); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // SPREAD: request q = 4 g + m of the burst (fp32 rows: group g, instruction m), issued in 24 shares
    constexpr int NDMA = (IN_BF16 ? 2 : 4) * SPLIT, NSHARE = 24;   // (bf16 rows: two requests per group, dma_group)
    [[maybe_unused]] const uint32_t dvo = dma_voff(voff, v_sn);
    [[maybe_unused]] auto dma_one = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if constexpr (IN_BF16) {
        constexpr int g = q / 2, mh = q % 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_next, (__attribute__((address_space(3))) void*)(slot + (2 * g + mh) * 1024), 16,
                                                 dvo + (uint32_t)((64 * g + 2048 * mh) * v_sn * ESI), 0, 0, 0);
      } else {
        constexpr int g = q / 4, m = q % 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_next, (__attribute__((address_space(3))) void*)(slot + (4 * g + m) * 1024), 16,
                                                 voff + (uint32_t)((64 * g + 1024 * m) * v_sn * 4), 0, 0, 0);
      }
    };
    [[maybe_unused]] auto dma_share = [&](auto sc) {
      constexpr int sl = decltype(sc)::value;
      static_for<sl * NDMA / NSHARE, (sl + 1) * NDMA / NSHARE>([&](auto qc) { dma_one(qc); });
    };
    if constexpr (!SPREAD) static_for<0, SPLIT>([&](auto gc) { dma_group(rs_next, dma_voff(voff, v_sn), v_sn, gc); });
    asm volatile("" ::: "memory");                 // the vmcnt() at the top of the loop counts on these being older than every store below

    // ---- conj twiddle, I2, stores (spectre.py:553) interleaved with the loads that refill the released registers -----------
    static_for<1, 64>([&](auto jc) {
      constexpr int j = decltype(jc)::value, ja = j % 8, jb = j / 8;     // position j carries k1 = j
      if constexpr (ja > 0) z[j] = cmulc(z[j], wa[ja]);
      if constexpr (jb > 0) z[j] = cmulc(z[j], wb[jb]);
      if constexpr (ja == 7) {
        pin8<8 * jb, 1>(z); __builtin_amdgcn_sched_barrier(0);   // one wb at a time
        if constexpr (SPREAD) { dma_share(std::integral_constant<int, jb>{}); __builtin_amdgcn_sched_barrier(0); }
      }
    });
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SPREAD) p64_stageA1_cb<true>(z, [&](auto q0c) { dma_share(std::integral_constant<int, 8 + decltype(q0c)::value>{}); });
    else p64_stageA1<true>(z);
    asm volatile("" ::: "memory");
    {
      const uint32_t ooff = (uint32_t)(((long long)(u + 512 * h) * out_sn + 4 * pp) * ESO);
      if constexpr (BURST) {
        static_for<0, 8>([&](auto ic) {
          constexpr int g = (decltype(ic)::value + SPLIT) % 8;
          fftA_stage2_group<8, 8, true, g>(z);                            // rows g + 8e at positions 8g + e
          pin8<8 * g, 1>(z);
          swap_group(std::integral_constant<int, g>{});
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (SPREAD) { dma_share(std::integral_constant<int, 16 + decltype(ic)::value>{}); __builtin_amdgcn_sched_barrier(0); }
        });
        asm volatile("" ::: "memory");                  // (SPREAD: the last LDS-DMA request is older than every store below)
        p64_barrier();                                  // the eight waves start the burst together
        static_for<0, 8>([&](auto ic) {
          constexpr int g = (decltype(ic)::value + SPLIT) % 8;
          static_for<0, 4>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const float4 res = make_float4(z[8 * g + 2 * m].x, z[8 * g + 2 * m].y, z[8 * g + 2 * m + 1].x, z[8 * g + 2 * m + 1].y);
            if constexpr (g >= GP) {
              if (more) {                            // trade places: results wait for the next quiet part, the prefetched rows move in
                const float4 nx = dfr[4 * (g - GP) + m];
                dfr[4 * (g - GP) + m] = res;
                if constexpr (IN_BF16) {
                  z[8 * g + 2 * m] = unpack_lo(__float_as_uint(nx.x));
                  z[8 * g + 2 * m + 1] = unpack_lo(__float_as_uint(nx.y));
                } else {
                  z[8 * g + 2 * m] = make_float2(nx.x, nx.y);
                  z[8 * g + 2 * m + 1] = make_float2(nx.z, nx.w);
                }
              } else {
                store16(rs_out, ooff + (uint32_t)((64 * g + 1024 * m) * out_sn * ESO), res);
              }
            } else {
              store16(rs_out, ooff + (uint32_t)((64 * g + 1024 * m) * out_sn * ESO), res);
            }
          });
          __builtin_amdgcn_sched_barrier(0);
        });
        p64_barrier();                                  // ... and the reloads start when every wave's stores are out
        if constexpr (EARLY1) {
          if (more) {                                   // the rows that have just traded places with the results: stage 1 of the next tile's F1
            static_for<0, PF>([&](auto ic) {
              constexpr int g = GP + decltype(ic)::value;
              swap_group(std::integral_constant<int, g>{});
              bfly_plain<8, false, 8 * g, 1, 64>(z);
              pin8<8 * g, 1>(z);
              __builtin_amdgcn_sched_barrier(0);
            });
          }
        }
        static_for<SPLIT, GP>([&](auto gc) { load_group(rs_next, voff, v_sn, gc); });
        __builtin_amdgcn_sched_barrier(0);
      } else {
      static_for<0, 8>([&](auto ic) {
        constexpr int g = (decltype(ic)::value + SPLIT) % 8;             // register-loaded groups first: their reloads start earliest
        fftA_stage2_group<8, 8, true, g>(z);                              // rows g + 8e at positions 8g + e
        pin8<8 * g, 1>(z);
        swap_group(std::integral_constant<int, g>{});
        static_for<0, 4>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          const float4 res = make_float4(z[8 * g + 2 * m].x, z[8 * g + 2 * m].y, z[8 * g + 2 * m + 1].x, z[8 * g + 2 * m + 1].y);
          if constexpr (g >= GP) {
            if (more) {                              // trade places: results wait for the next quiet part, the prefetched rows move in
              const float4 nx = dfr[4 * (g - GP) + m];
              dfr[4 * (g - GP) + m] = res;
              if constexpr (IN_BF16) {
                z[8 * g + 2 * m] = unpack_lo(__float_as_uint(nx.x));
                z[8 * g + 2 * m + 1] = unpack_lo(__float_as_uint(nx.y));
              } else {
                z[8 * g + 2 * m] = make_float2(nx.x, nx.y);
                z[8 * g + 2 * m + 1] = make_float2(nx.z, nx.w);
              }
            } else {
              store16(rs_out, ooff + (uint32_t)((64 * g + 1024 * m) * out_sn * ESO), res);
            }
          } else {
            store16(rs_out, ooff + (uint32_t)((64 * g + 1024 * m) * out_sn * ESO), res);
          }
        });
        if constexpr (g >= SPLIT && g < GP) load_group(rs_next, voff, v_sn, std::integral_constant<int, g>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    }
      }
    obp = ob;
    gate_fetch(gpn);     // committed to LDS at the end of the next tile's F1 (after the last tile: a harmless re-read of this tile's bins)
    if constexpr (TICKETS) {
      prev_live = cur_live && more;                  // (only then did this tile's deferred results go into dfr; without a next tile they were stored above)
      cur_tile = nxt_tile; nxt_tile = fut_tile; ++tk_seq;
      if (fut_tile == -2) tk_follow = false;         // the stream has ended (or the leader fell silent): no more requests
    }
  }  // tile loop
  if constexpr (!TICKETS) break;
  }  // rounds
}

hipError_t launch_regtile64p(const RegtileArgs& a, bool in_bf16, bool out_bf16, bool burst, hipStream_t stream);

}  // namespace sfft
