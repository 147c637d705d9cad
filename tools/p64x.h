// kernel_regtile64p.h — persistent, software-pipelined spectral mix for n_fft = 4096 = 64 x 64 on gfx950 (every 16-channel tile
// inside one gate group, 16-byte aligned fp32 rows or 8-byte aligned bf16 rows in, fp32 or (with bf16 in) bf16 rows out; optional memory_fft; ANY sequence length: rows beyond N_in are the buffer
// instructions' out-of-range case — loads return 0 = rfft's zero padding (spectre.py:506), stores are dropped (spectre.py:553) —
// so a padded sequence costs exactly what a full one costs, without a single predicate).
//
// Same mathematics and the same register-tile plan as kernel_regtile.h (one workgroup owns 16 channels x 4096 rows; F1 ->
// twiddle -> E1 -> F2 -> gate -> I1 -> E2 -> conj twiddle -> I2; replaces /root/reference/spectre.py:506 + :542-553), rebuilt
// around what the round-2 measurements showed (tools/trace_bench.hip, tools/iolab.hip, profiles/r02_*):
//
//  * with one workgroup per tile all 256 CUs run in lock-step — everybody loads, then everybody computes — and that state is an
//    attractor (a CU that loads while the others compute finishes early and drifts back into the pack), so HBM idles while the
//    chip computes.  Loads of tile t+1 therefore have to be in flight while tile t is still being computed on the SAME CU
//    (the persistent workgroups of this kernel do drift apart: 115 +- 14 of 256 are in their store/load burst at any instant);
//  * a wave's stores and loads retire through one in-order counter (vmcnt): loads issued behind the stores of the previous
//    tile cannot be consumed before those stores are acknowledged.  The next tile's first half is therefore requested BEFORE
//    the stores, as LDS-DMA (buffer_load_dwordx4 ... lds, no VGPR needed) into the exchange image, which is idle between the last
//    exchange of tile t and the first exchange of tile t+1.  Every lane reads back exactly the 16 bytes it requested, so the
//    staging needs no barrier of its own;
//  * (library: SPLIT = 3 row groups through LDS-DMA, PF = 3 deferred, 2 reloaded behind their own stores)
//  * three of the other row groups (PF = 3) are deferred: their results stay in 48 registers through F1 of the next tile, are
//    stored at the end of F1, and the same registers then prefetch those groups of the tile after — 3/8 of the traffic travels while
//    the CU exchanges and multiplies; the last group is loaded straight into the registers that the stores of I2 have just released;
//  * for the same vmcnt reason nothing that is needed "now" may be a global load: the twiddle vectors live in LDS (written once per
//    workgroup), the gate bins are requested a tile ahead and committed to LDS as late as possible;
//  * 16-byte global accesses: a lane moves the 4 channels (2 packed sequences) of one row; v_permlane16_swap hands the
//    second sequence to the partner lane (lane ^ 16) and receives the partner's row of this lane's sequence, so a lane still
//    owns ONE sequence.  The lane <-> (sequence, row class) map is chosen so that the existing conflict-free LDS image layout
//    stays conflict-free (checked with the bank model of MI355X_MICROARCH.md: write banks 4p + rc, b128 read groups distinct);
//  * requests are issued as early and as bunched as the registers allow (spreading them over the arithmetic was slower), and pairs
//    of workgroups walk through adjacent tiles: a 64-byte row segment is half an L2 line, the L2 fetches whole lines, and the
//    neighbour's request a few microseconds later hits.  Larger gangs of neighbours collide on DRAM channels and were slower.
// Where it stands: the same instruction stream without memory traffic takes 0.94 ms, with every request answered
// by the L2 1.0 ms, the product 1.51-1.62 ms; the difference is HBM latency beyond what 128 KiB of staging + 48 registers cover and
// the DRAM efficiency of half-line segments (tools/iolab3.hip).
//
// Thread <-> data:  lane = (pp = lane & 3, rcl = (lane >> 2) & 3, h = (lane >> 4) & 1, rch = lane >> 5);
//   sequence p = 2 pp + h, team index u = rcl + 4 rch + 8 wave  (n2 in F1 / I2, k1 in the middle phase);
//   register position j = 8 g + e  <->  row n1 = g + 8 e  (both when loading and when storing).
#pragma once
#include "../fft_amd/csrc/kernel_regtile.h"

namespace sfft {

// the round-2 forms of the two-factor transforms (twiddle as a complex multiplication on stage 1's outputs): fft_regs.h has since moved
// the twiddles into stage 2's butterflies in scaled form, this frozen kernel keeps the arithmetic it was measured with
template <int RA, int RB, bool INV, int KA, int OFF = 0, int NTOT = RA * RB>
__device__ __forceinline__ void xfftA_stage2_group(float2 (&z)[NTOT]) { bfly<RB, INV, OFF + RB * KA, 1, NTOT>(z); }
template <int RA, int RB, bool INV, int KA, int OFF = 0, int NTOT = RA * RB>
__device__ __forceinline__ void xfftB_stage1_group(float2 (&z)[NTOT]) {
  constexpr int R = RA * RB, U = 64 / R;
  bfly<RB, INV, OFF + RB * KA, 1, NTOT>(z);
  if constexpr (KA > 0)
    static_for<1, RB>([&](auto nc) { constexpr int nlo = decltype(nc)::value; z[OFF + RB * KA + nlo] = twid64<U * KA * nlo, INV>(z[OFF + RB * KA + nlo]); });
}

// the argument block of the round-2 kernel (the library's RegtileArgs has since lost the experiment fields)
struct XRegtileArgs {
  const void* v; const float2* gate; const float* mem; void* out; const float2* tw;
  int B, N_in, D, G, d_g, F;
  int tiles_per_row, n_tiles;
  long long v_sb, v_sn, out_sb, out_sn;
  int tpw, n_wg, conj_gate;
  unsigned long long* trace;   // ABL bit4: 8 words per (workgroup, tile); XP bit0: 8 x 16 uint32 per (workgroup, tile)
  int pf_dist;                 // ABL bit13
  unsigned* gang_cnt;          // ABL bit5: zeroed rendezvous counters, 4 words per wave pair
};

constexpr int xkP64ImageBytes = regtile_image_bytes<64, 64, 1>();
// LDS: exchange image | half-spectrum gate | the two twiddle vectors of every team index u (W^(u j), W^(8 u j), j = 1..7): 64 x 14 x 8 B.
// The twiddles are read twice per tile; as global loads they queue (one in-order vmcnt) behind the LDS-DMA requests of the next tile at
// the start of the store/load burst and behind the last stores at the start of F1, i.e. every tile paid a full HBM round trip for
// 112 bytes that never change.  From LDS they cost 7 ds_read_b128 and no vmcnt.
constexpr int xkP64TwOff = (regtile_lds_total<64, 64, 1>() + 15) & ~15;
constexpr int xkP64LdsTotal = xkP64TwOff + 64 * 14 * 8;
static_assert(xkP64LdsTotal <= 160 * 1024, "LDS budget");

typedef unsigned int xp64_u32x4 __attribute__((ext_vector_type(4)));
constexpr int xkP64RsrcFlags = 0x00020000;        // raw buffer, 32-bit data format (gfx90a / gfx942 / gfx950 dword 3)

__device__ __forceinline__ void xlane16_swap(float& x, float& y) {   // x of the odd 16-lane rows <-> y of the even rows
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}

// The 8 x 8 in-register transforms of fft_regs.h with a scheduling fence after every radix-8 butterfly: hipcc otherwise interleaves
// all eight butterflies of a stage (up to 120 temporaries on top of the 128 data registers: the middle phase and I2's first stage
// peak at 236 and 250 live VGPRs), and with the deferred-result registers of PF > 0 on top it parks those in scratch instead.
// pin<BASE, STRIDE>: the eight values z[BASE + STRIDE j] have to exist in registers HERE.  sched_barrier only fences the machine
// scheduler; the IR-level sinking pass still splits a butterfly and leaves half-finished sums alive until their first use
// hundreds of instructions later (that, not the schedule, is where the 250-register peaks come from).
template <int BASE, int STRIDE>
__device__ __forceinline__ void xpin8(float2 (&z)[64]) {
  asm volatile("" : "+v"(z[BASE].x), "+v"(z[BASE].y), "+v"(z[BASE + STRIDE].x), "+v"(z[BASE + STRIDE].y),
                    "+v"(z[BASE + 2 * STRIDE].x), "+v"(z[BASE + 2 * STRIDE].y), "+v"(z[BASE + 3 * STRIDE].x), "+v"(z[BASE + 3 * STRIDE].y),
                    "+v"(z[BASE + 4 * STRIDE].x), "+v"(z[BASE + 4 * STRIDE].y), "+v"(z[BASE + 5 * STRIDE].x), "+v"(z[BASE + 5 * STRIDE].y),
                    "+v"(z[BASE + 6 * STRIDE].x), "+v"(z[BASE + 6 * STRIDE].y), "+v"(z[BASE + 7 * STRIDE].x), "+v"(z[BASE + 7 * STRIDE].y));
}

template <bool INV, bool FENCE>
__device__ __forceinline__ void xp64_stageA1(float2 (&z)[64]) {      // type A stage 1: radix-8 over q1 (positions 8 q1 + q0), * W_64^(q0 ka)
  static_for<0, 8>([&](auto q0c) {
    constexpr int q0 = decltype(q0c)::value;
    bfly<8, INV, q0, 8, 64>(z);
    static_for<1, 8>([&](auto kac) { constexpr int ka = decltype(kac)::value; z[8 * ka + q0] = twid64<q0 * ka, INV>(z[8 * ka + q0]); });
    if constexpr (FENCE) { xpin8<q0, 8>(z); __builtin_amdgcn_sched_barrier(0); }
  });
}
template <bool INV, bool FENCE>
__device__ __forceinline__ void xp64_stageB2(float2 (&z)[64]) {      // type B stage 2: radix-8 over ka (positions 8 ka + n_lo)
  static_for<0, 8>([&](auto nc) {
    bfly<8, INV, decltype(nc)::value, 8, 64>(z);
    if constexpr (FENCE) { xpin8<decltype(nc)::value, 8>(z); __builtin_amdgcn_sched_barrier(0); }
  });
}

// The two exchanges of this kernel (kernel_regtile.h exchange_planes_b128 with wr(j) = j * RW + p * PS + u, rd4(m) = u * RW + p * PS + m),
// with the 64 scattered dword writes of a plane issued as 32 ds_write2st64_b32: the LDS takes a store's address and data registers at
// 2 cycles per dword, so one instruction with two data dwords (6 cycles) beats two ds_write_b32 (8 cycles), and the writes are 80 % of
// an exchange's LDS time.  Rows j and j + 2 are 2 * 2176 = 17 * 256 bytes apart — a multiple of the instruction's 256-byte offset unit;
// its 8-bit offsets reach 30 rows, hence four opaque base addresses (even / odd rows below and above 32) instead of one.
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier, and hipcc implements the
// fence with s_waitcnt vmcnt(0): every barrier of the exchanges would drain the deferred stores and the prefetches that are meant to
// travel DURING the exchanges.  Nothing that crosses waves goes through global memory here (a lane reads back only what its own wave
// requested by LDS-DMA, after its own vmcnt wait), so the LDS counter is all a barrier has to wait for.
__device__ __forceinline__ void xp64_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool IN_BF16> constexpr int xkP64Gang = IN_BF16 ? 4 : 2;   // workgroups per 128-byte line (launch: n_wg is a multiple of it)

// early Wre: the real parts of positions KA + 8 kb (kb = 0..7) -> image rows KA + 8 kb, column (p, u); rows 8 apart are 68 x 256 bytes
// apart, so four ds_write2st64_b32 with two opaque bases (kb < 4, kb >= 4) do it
template <int KA>
__device__ __forceinline__ void xp64_write_im_col(float2 (&z)[64], float* img, int p, int u) {
  constexpr int RW = 8 * 68, PS = 68;
  typedef __attribute__((address_space(3))) float lds_float;
  lds_float* lo = (lds_float*)(img + p * PS + u) + KA * RW;
  lds_float* hi = lo + 32 * RW;
  asm volatile("" : "+v"(lo), "+v"(hi));
  lo[0] = z[KA].y;            lo[8 * RW] = z[KA + 8].y;
  lo[16 * RW] = z[KA + 16].y; lo[24 * RW] = z[KA + 24].y;
  hi[0] = z[KA + 32].y;       hi[8 * RW] = z[KA + 40].y;
  hi[16 * RW] = z[KA + 48].y; hi[24 * RW] = z[KA + 56].y;
}
template <int KA>
__device__ __forceinline__ void xp64_write_re_col(float2 (&z)[64], float* img, int p, int u) {
  constexpr int RW = 8 * 68, PS = 68;
  typedef __attribute__((address_space(3))) float lds_float;
  lds_float* lo = (lds_float*)(img + p * PS + u) + KA * RW;
  lds_float* hi = lo + 32 * RW;
  asm volatile("" : "+v"(lo), "+v"(hi));
  lo[0] = z[KA].x;            lo[8 * RW] = z[KA + 8].x;
  lo[16 * RW] = z[KA + 16].x; lo[24 * RW] = z[KA + 24].x;
  hi[0] = z[KA + 32].x;       hi[8 * RW] = z[KA + 40].x;
  hi[16 * RW] = z[KA + 48].x; hi[24 * RW] = z[KA + 56].x;
}

// XF: bit0 = the real plane has already been written by the producer (early Wre), bit1 = no LDS traffic (barriers only), bit2 = no barrier after the last read
template <bool LAST_BARRIER, int XF = 0, class STAMP>
__device__ __forceinline__ void xp64_exchange(float2 (&z)[64], float* img, int p, int u, STAMP stamp) {
  constexpr int RW = 8 * 68, PS = 68;
  typedef __attribute__((address_space(3))) float lds_float;
  lds_float* w0 = (lds_float*)(img + p * PS + u);
  lds_float *w1 = w0 + RW, *w2 = w0 + 32 * RW, *w3 = w0 + 33 * RW;
  asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));       // keep them apart: base + 16-bit offset would fold them back together
  const float* rd = img + u * RW + p * PS;
  auto write_plane = [&](auto is_im) {
    static_for<0, 16>([&](auto ic) {                                  // rows (4i, 4i + 2) and (4i + 1, 4i + 3)
      constexpr int j = 4 * decltype(ic)::value, jw = j % 32;
      lds_float* we = j < 32 ? w0 : w2;
      lds_float* wo = j < 32 ? w1 : w3;
      if constexpr (decltype(is_im)::value) {
        we[jw * RW] = z[j].y; we[(jw + 2) * RW] = z[j + 2].y;
        wo[jw * RW] = z[j + 1].y; wo[(jw + 2) * RW] = z[j + 3].y;
      } else {
        we[jw * RW] = z[j].x; we[(jw + 2) * RW] = z[j + 2].x;
        wo[jw * RW] = z[j + 1].x; wo[(jw + 2) * RW] = z[j + 3].x;
      }
    });
  };
  auto read_plane = [&](auto is_im) {
    static_for<0, 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value, m = 4 * ((i / 8) + 2 * (i % 8));   // the chunk order of exchange_planes_b128<64, 8, 8>
      const float4 v = *reinterpret_cast<const float4*>(rd + m);
      if constexpr (decltype(is_im)::value) { z[m].y = v.x; z[m + 1].y = v.y; z[m + 2].y = v.z; z[m + 3].y = v.w; }
      else { z[m].x = v.x; z[m + 1].x = v.y; z[m + 2].x = v.z; z[m + 3].x = v.w; }
    });
  };
  constexpr bool LDS = (XF & 2) == 0;
  if constexpr (LDS && (XF & 1) == 0) write_plane(std::false_type{});
  xp64_barrier();
  stamp(0);
  if constexpr (LDS) read_plane(std::false_type{});
  xp64_barrier();
  stamp(1);
  if constexpr (LDS && (XF & 8) != 0) static_for<0, 8>([&](auto cc) { xp64_write_im_col<decltype(cc)::value>(z, img, p, u); });
  else if constexpr (LDS) write_plane(std::true_type{});
  xp64_barrier();
  stamp(2);
  if constexpr (LDS) read_plane(std::true_type{});
  if constexpr (LAST_BARRIER && (XF & 4) == 0) xp64_barrier();     // image free again
}

// SPLIT = row groups (of 8) of the next tile that travel through LDS (0: everything is loaded behind the stores).
// PF    = row groups whose I/O is moved out of the store/load burst into the exchange / middle phase, when this CU has no other
//         memory traffic in flight: the results of the last PF groups of tile t stay in 16 PF registers through F1 of tile t+1 and
//         are stored right before E1; the same registers then receive those groups of tile t+2, which trade places with the next
//         results at the end of I2.  (3 in the library: 250 VGPRs; 4 spills.)
// ABL   = experiments of tools/p64_ab_bench.hip / trace64p_bench.hip, 0 in the library: bit4 phase timestamps, bit5 wave-pair
//         rendezvous, bits 8-11 traffic switched off or kept inside the L2, bit12 chip-wide sweep, bit13 rotated pair ranges.
// IN_BF16 = bf16 rows in (spectre.py's activations under autocast), fp32 arithmetic and fp32 rows out: a lane still moves the 4
//         channels of a row — 8 bytes, two packed dwords = its two sequences — so the lane map, the swap and everything after it are
//         the fp32 kernel's; only the staging differs (8-byte LDS-DMA does not exist: a DMA instruction fetches 8 whole 32-byte row
//         segments, lane = (row, dword), and every lane reads its 8 bytes back out of its wave's slot).
// OUT_BF16 = bf16 rows out (round to nearest even, like every other kernel here): a lane stores its 4 channels as 8 bytes.
template <int SPLIT, int PF = 0, int ABL = 0, bool FEN = (PF > 0), bool WITH_MEM = false, bool IN_BF16 = false, bool OUT_BF16 = false, int XP = 0, int KB0 = 0>
__global__ void __launch_bounds__(512, 2) spectre_mix_p64x(const XRegtileArgs a) {
  constexpr int RW = 8 * 68, PS = 68;              // image row / column strides in floats (kernel_regtile.h, 16-byte layout)
  constexpr int ESI = IN_BF16 ? 2 : 4, ESO = OUT_BF16 ? 2 : 4;   // bytes per input / output element
  constexpr float inv_n = 1.0f / 4096.0f;
  static_assert(SPLIT >= 0 && SPLIT <= 4 && SPLIT * 4 * 1024 * 8 <= xkP64ImageBytes, "staging lives in the exchange image");
  static_assert(PF >= 0 && SPLIT + PF <= 8, "row groups: SPLIT through LDS, PF deferred / prefetched in registers, the rest behind their stores");
  constexpr int GP = 8 - PF;                       // first deferred / prefetched group
  static_assert(!WITH_MEM || FEN, "memory_fft: one register group of gate bins and memory rows at a time");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);
  float2* glds = reinterpret_cast<float2*>(smem + xkP64ImageBytes);
  float2* twl = reinterpret_cast<float2*>(smem + xkP64TwOff);
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
    slot = smem + __builtin_amdgcn_readfirstlane(t >> 6) * (SPLIT * 4 * 1024);   // this wave's landing slots
  };
  coords();

  const int wg_lin = xcd_contiguous(blockIdx.x, a.n_wg);
  // ABL bit12 (tools/p64_ab_bench.hip): the whole chip sweeps the tiles front to back (tile = workgroup + n_wg * iteration, every XCD
  // on 32 adjacent tiles) instead of every pair of workgroups walking through its own 2 * tpw tiles
  constexpr bool SWEEP = (ABL & 4096) != 0;
  // GANG neighbouring workgroups (same L2) walk through GANG adjacent tiles in step = one 128-byte line per row: the L2 fetches a
  // line once and the neighbours' requests hit (fp32: two 64-byte halves; bf16: four 32-byte quarters)
  constexpr int GANG = xkP64Gang<(IN_BF16 || OUT_BF16)>;
  const int tile_step = SWEEP ? a.n_wg : GANG;
  const int pair_base = SWEEP ? wg_lin : (wg_lin / GANG) * a.tpw * GANG + (wg_lin % GANG);
  if (pair_base >= a.n_tiles) return;

  // ABL bit5 (tools/p64_ab_bench.hip only; measured and NOT shipped: profiles/r02_p64_ab_rendezvous.log).  A 64-byte row segment is
  // half an L2 line.  Reads: the L2 fetches the whole line on a miss and the neighbour workgroup's request a few microseconds later
  // hits (FETCH_SIZE = the algorithmic bytes).  Writes: in a pure copy the two halves of a line leave the L2 as one DRAM burst only if
  // they were written within about a microsecond of each other (tools/iolab3.hip: 64-byte segments 3.4 TB/s, halves together 5.4).
  // Here wave w of workgroup 2m and wave w of workgroup 2m+1 — the owners of the two halves of the same lines — meet before every group
  // of 4 store instructions: s_atomic_add to arrive (before the group's butterflies), s_load_dword glc to poll; scalar unit only, no
  // VGPR, no vmcnt, no workgroup barrier; a wave whose partner does not show up within 24 polls drops out for the rest of the launch.
  // Result: the rendezvous works (all 1024 wave pairs stay together, ~0.4 failed polls per meeting) and the kernel gets 2-4 % SLOWER.
  [[maybe_unused]] unsigned* gcnt = nullptr;
  [[maybe_unused]] bool gang_live = false;
  [[maybe_unused]] unsigned gang_done = 0, gang_polls = 0;
  if constexpr ((ABL & 32) != 0) {
    gcnt = a.gang_cnt + ((wg_lin >> 1) * 8 + __builtin_amdgcn_readfirstlane(tid0 >> 6)) * 4;
    gang_live = true;
  }
  auto gang_arrive = [&]() {
    if constexpr ((ABL & 32) != 0) {
      if (gang_live) { const unsigned one = 1; asm volatile("s_atomic_add %0, %1, 0x0" :: "s"(one), "s"(gcnt) : "memory"); }
    }
  };
  auto gang_await = [&]([[maybe_unused]] unsigned members) {   // members = workgroups of the pair that have a tile in this iteration
    if constexpr ((ABL & 32) != 0) {
      gang_done += members;
      if (gang_live && members > 1) {
        int spin = 0;
        for (;;) {
          unsigned now;
          asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(now) : "s"(gcnt) : "memory");
          if ((int)(now - gang_done) >= 0) break;
          ++gang_polls;
          if (++spin > 24) { gang_live = false; break; }
        }
      }
    }
  };

  // ABL (tools/trace64p_bench.hip only; 0 in the library): bit4 = phase timestamps (100 MHz) into a.trace, 8 per (workgroup, tile)
  [[maybe_unused]] auto stamp = [&](int it, int slot) {
    if constexpr ((ABL & 16) != 0) { if (threadIdx.x == 0) a.trace[((size_t)blockIdx.x * a.tpw + it) * 8 + slot] = wall_clock64(); }
  };
  // XP bit0: per-WAVE phase stamps (low 32 bits of s_memtime = shader cycles), 16 slots per wave and tile, parked in the last 512 bytes
  // of the LDS allocation and flushed to a.trace (as uint32) at the end of every tile
  [[maybe_unused]] unsigned* wst = reinterpret_cast<unsigned*>(smem + xkP64LdsTotal) + (tid0 >> 6) * 16;
  [[maybe_unused]] auto wstamp = [&](int slot) {
    if constexpr ((XP & 1) != 0) { if ((tid0 & 63) == 0) wst[slot] = (unsigned)__builtin_amdgcn_s_memtime(); }
  };
  constexpr bool VALU = (XP & 4) == 0;             // XP bit2: no butterflies / twiddles / gate multiply (LDS + barriers only)
  constexpr int XF_LDS = (XP & 2) != 0 ? 2 : 0;    // XP bit1: exchanges without LDS traffic (barriers only)
  constexpr bool EARLY = (XP & 8) != 0;            // XP bit3: real plane written by the producer, column by column
  constexpr bool LATEB = (XP & 16) != 0;           // XP bit4 (with bit3): E1's "image free" barrier moves to the end of the middle phase
  float2 z[64];
  float4 dfr[PF > 0 ? 4 * PF : 1];                 // deferred results of the previous tile / prefetched rows of the next one
  static_for<0, (PF > 0 ? 4 * PF : 1)>([&](auto ic) { dfr[decltype(ic)::value] = make_float4(0.f, 0.f, 0.f, 0.f); });   // (stored into an empty range before the first tile)
  char* obp = nullptr;                             // output tile of the deferred results
  float2 gstage[5];      // the next tile's gate bins on their way to LDS (4097 bins / 512 threads, rounded up; + 1 for the last)

  // lane offset of an LDS-DMA request: fp32 = the lane's own 16 bytes (the register-load offset); bf16 = (row l >> 3, dword l & 7)
  auto dma_voff = [&](uint32_t voff, long long sn) -> uint32_t {
    if constexpr (IN_BF16) return (uint32_t)(((long long)((lane >> 3) + 8 * (u >> 3)) * sn) * ESI + (lane & 7) * 4);
    else return voff;
  };
  auto tile_ptrs = [&](int tile, const char*& vb, char*& ob, const float2*& gp) {
    const int b = tile / a.tiles_per_row, ct = tile - b * a.tiles_per_row;
    vb = reinterpret_cast<const char*>(a.v) + ((size_t)b * a.v_sb + (size_t)ct * 16) * ESI;
    ob = reinterpret_cast<char*>(a.out) + ((size_t)b * a.out_sb + (size_t)ct * 16) * ESO;
    // ABL bit10 / bit11 (tools/p64_ab_bench.hip): every workgroup of an XCD stores to / loads from ONE dense 256-KiB tile (row stride
    // 64 bytes, see v_sn / out_sn in the tile loop) — real requests and acknowledgements that never leave the L2
    if constexpr ((ABL & 1024) != 0) ob = reinterpret_cast<char*>(a.out) + (size_t)(blockIdx.x % 8) * (4096 * 64);
    if constexpr ((ABL & 2048) != 0) vb = reinterpret_cast<const char*>(a.v) + (size_t)(blockIdx.x % 8) * (4096 * 64);
    gp = a.gate + ((size_t)b * a.G + (ct * 16) / a.d_g) * a.F;
  };
  // row of this lane in load / store instruction (g, m):  u + 512 h + 64 g + 1024 m; addresses = workgroup-uniform base of the
  // instruction (SGPRs) + one 32-bit lane offset (spectre_hip.hip bounds 4095 * row stride * 4 + 64 below 2^31)
  // Buffer resources: base = the tile's first row, num_records = the bytes of its rows below N_in.  The range check covers the
  // VGPR offset (lane offset + row-block offset; the SGPR offset operand is not checked on gfx9), so both go there.
  // ABL bit8 / bit9 (tools/p64_ab_bench.hip): an empty range for the stores / the loads — the same instruction stream without the
  // memory traffic (out-of-range stores are dropped, out-of-range loads return 0 before they leave the CU)
  // live = false: an empty range.  Every request of the tile loop is issued UNCONDITIONALLY — after the last tile (and, for the deferred
  // stores, before the first) with an empty range, which costs nothing: hipcc computes its s_waitcnt vmcnt(N) from the requests that are
  // GUARANTEED to be younger than the one waited for, so a request inside `if (more)` does not count, N comes out too small, and a wait
  // for a prefetched register early in I2 turned into a wait for the LDS-DMA issued just before it (a full HBM round trip per tile).
  auto rsrc_in = [&](const char* vb, long long sn, bool live = true) {
    const int rows = (ABL & 512) != 0 || !live ? 0 : a.N_in < 4096 ? a.N_in : 4096;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vb), 0, (int)((long long)rows * sn * ESI), xkP64RsrcFlags);
  };
  auto rsrc_out = [&](char* ob, long long sn, bool live = true) {
    const int rows = (ABL & 256) != 0 || !live ? 0 : a.N_in < 4096 ? a.N_in : 4096;
    return __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)((long long)rows * sn * ESO), xkP64RsrcFlags);
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
        const xp64_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (uint32_t)((64 * g + 1024 * m) * sn * 4), 0, 0);
        z[8 * g + 2 * m] = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
        z[8 * g + 2 * m + 1] = make_float2(__uint_as_float(t.z), __uint_as_float(t.w));
      }
    });
  };
  // fp32: 1 KiB per instruction, lane l's 16 bytes at slot + 16 l.  bf16: 256 B per instruction = the 8 rows (rcl + 4 rch) of one h,
  // lane l = (row l >> 3, dword l & 7); dvoff = that lane's offset inside row block 0 (computed per tile by the caller).
  auto dma_group = [&](__amdgpu_buffer_rsrc_t rs, uint32_t voff, long long sn, auto gc) {        // into this wave's LDS slots
    constexpr int g = decltype(gc)::value;
    static_for<0, 4>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if constexpr (IN_BF16) {
        static_for<0, 2>([&](auto hc) {
          constexpr int hh = decltype(hc)::value;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(slot + ((4 * g + m) * 2 + hh) * 256), 4,
                                                   voff + (uint32_t)((64 * g + 1024 * m + 512 * hh) * sn * ESI), 0, 0, 0);
        });
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(slot + (4 * g + m) * 1024), 16,
                                                 voff + (uint32_t)((64 * g + 1024 * m) * sn * 4), 0, 0, 0);
      }
    });
  };
  auto store16 = [&](__amdgpu_buffer_rsrc_t rs, uint32_t off, const float4 v) {   // this lane's 4 channels of one row
    if constexpr (OUT_BF16) {
      rt_u32x2 t;
      t.x = f32_to_bf16_rne(v.x) | (f32_to_bf16_rne(v.y) << 16); t.y = f32_to_bf16_rne(v.z) | (f32_to_bf16_rne(v.w) << 16);
      __builtin_amdgcn_raw_buffer_store_b64(t, rs, off, 0, 0);
    } else {
      xp64_u32x4 t;
      t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y); t.z = __float_as_uint(v.z); t.w = __float_as_uint(v.w);
      __builtin_amdgcn_raw_buffer_store_b128(t, rs, off, 0, 0);
    }
  };
  auto read_group = [&](auto gc) {                                       // this lane's bytes back out of the slot
    constexpr int g = decltype(gc)::value;
    static_for<0, 4>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if constexpr (IN_BF16) {
        const rt_u32x2 t = *reinterpret_cast<const rt_u32x2*>(slot + ((4 * g + m) * 2 + h) * 256 + (((lane >> 2) & 3) + 4 * (lane >> 5)) * 32 + pp * 8);
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
      xlane16_swap(z[j].x, z[j + 1].x);
      xlane16_swap(z[j].y, z[j + 1].y);
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

  // ---- prologue: request tile 0 the same way every later tile is requested --------------------------------------------
  {
    const char* vb; char* ob; const float2* gp;
    tile_ptrs(pair_base, vb, ob, gp);
    const uint32_t voff = (uint32_t)(((long long)(u + 512 * h) * a.v_sn + 4 * pp) * ESI);
    const __amdgpu_buffer_rsrc_t rs = rsrc_in(vb, a.v_sn);
    static_for<0, SPLIT>([&](auto gc) { dma_group(rs, dma_voff(voff, a.v_sn), a.v_sn, gc); });
    asm volatile("" ::: "memory");
    static_for<SPLIT, 8>([&](auto gc) { load_group(rs, voff, a.v_sn, gc); });
    gate_fetch(gp);
  }

  for (int it = 0; it < a.tpw; ++it) {
    // ABL bit13 (tools/p64_ab_bench.hip; shapes whose tile count is a multiple of 2 * tpw): every pair starts its walk at a different
    // position of its range, so that pairs that run in step are never at the same offset of their (25 MB-aligned) ranges
    [[maybe_unused]] const int rot = (ABL & 8192) != 0 ? ((wg_lin >> 1) * a.pf_dist) % a.tpw : 0;
    const int tile = (ABL & 8192) != 0 ? pair_base + 2 * ((it + rot) % a.tpw) : pair_base + tile_step * it;
    if (tile >= a.n_tiles) break;                  // workgroup-uniform
    const bool more = (it + 1 < a.tpw) && (tile + tile_step < a.n_tiles);
    [[maybe_unused]] const unsigned members = (unsigned)min(2, a.n_tiles - (tile & ~1));   // workgroups of the pair that have a tile in this iteration
    coords();
    long long v_sn = (ABL & 2048) != 0 ? 16 : a.v_sn, out_sn = (ABL & 1024) != 0 ? 16 : a.out_sn;
    asm volatile("" : "+s"(v_sn), "+s"(out_sn));
    const char* vb; char* ob; const float2* gp;
    tile_ptrs(tile, vb, ob, gp);
    const char* vbn = vb; char* obn = ob; const float2* gpn = gp;
    if (more) tile_ptrs((ABL & 8192) != 0 ? pair_base + 2 * ((it + 1 + rot) % a.tpw) : tile + tile_step, vbn, obn, gpn);
    const __amdgpu_buffer_rsrc_t rs_next = rsrc_in(vbn, v_sn, more), rs_out = rsrc_out(ob, out_sn);

    stamp(it, 0);
    wstamp(0);
    // ---- the tile arrives.  The LDS-staged groups were requested before the previous tile's stores and completion is in order, so
    //      once everything but the 16 youngest stores and the 5 gate loads has retired (vmcnt(21) below) they are in the slots.

    [[maybe_unused]] const uint32_t pf_ooff = (uint32_t)(((long long)(u + 512 * h) * out_sn + 4 * pp) * ESO);
    [[maybe_unused]] const uint32_t pf_voff = (uint32_t)(((long long)(u + 512 * h) * v_sn + 4 * pp) * ESI);
    [[maybe_unused]] auto pf_store = [&](auto ic) {
      constexpr int g = GP + decltype(ic)::value / 4, m = decltype(ic)::value % 4;
      store16(rsrc_out(obp, out_sn, it > 0), pf_ooff + (uint32_t)((64 * g + 1024 * m) * out_sn * ESO), dfr[decltype(ic)::value]);
    };
    [[maybe_unused]] auto pf_load = [&](auto ic) {
      constexpr int g = GP + decltype(ic)::value / 4, m = decltype(ic)::value % 4;
      if constexpr (IN_BF16) {                       // stays packed (two dwords) until it trades places with the results in I2
        const rt_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs_next, pf_voff + (uint32_t)((64 * g + 1024 * m) * v_sn * ESI), 0, 0);
        dfr[decltype(ic)::value].x = __uint_as_float(t.x); dfr[decltype(ic)::value].y = __uint_as_float(t.y);
      } else {
        const xp64_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_next, pf_voff + (uint32_t)((64 * g + 1024 * m) * v_sn * 4), 0, 0);
        dfr[decltype(ic)::value] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
      }
    };
    [[maybe_unused]] auto pf_block = [&]() {
      if constexpr ((ABL & 32) != 0) gang_await(members);   // (rendezvous experiment: the deferred stores leave an otherwise idle request queue)
      static_for<0, 4 * PF>([&](auto ic) { pf_store(ic); });
      static_for<0, 4 * PF>([&](auto ic) { pf_load(ic); });
    };
    constexpr bool PF_TOP = (ABL & 131072) != 0;   // ABL bit17 (p64_ab_bench): the deferred requests at the top of the tile instead of the end of F1
    if constexpr (PF > 0 && PF_TOP) { pf_block(); asm volatile("" ::: "memory"); }
    stamp(it, 1);
    // ---- F1: 64-point forward transform over n1 (register position 8g + e holds row g + 8e), then W_N^(u*k1) ------------
    //      Stage 1 works group by group, in the order the groups arrive: the deferred groups (prefetched a tile ago) first, then the
    //      LDS-staged ones — only now does the wave wait for the LDS-DMA of the burst it has just left — and the group reloaded
    //      behind the stores last.
    static_for<0, 8>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr bool ARRIVAL = (ABL & 16384) == 0;                           // ABL bit14 (p64_ab_bench): groups 0..7 in index order
      constexpr int g = !ARRIVAL ? i : i < PF ? GP + i : i - PF;             // [GP, 8), [0, SPLIT), [SPLIT, GP)
      if constexpr (SPLIT > 0 && i == (ARRIVAL ? PF : 0)) {
        // younger than the last LDS-DMA request: the reloaded groups' 4 stores + 4 loads each, the 4 SPLIT stores of the staged
        // groups, the 5 gate loads (first tile: the prologue's 4 (8 - SPLIT) loads and the 5 gate loads)
        constexpr int YOUNGER = 8 * (GP - SPLIT) + 4 * SPLIT + 5 + (PF_TOP ? 8 * PF : 0), YOUNGER0 = 4 * (8 - SPLIT) + 5 + (PF_TOP ? 8 * PF : 0);
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(YOUNGER0 < 63 ? YOUNGER0 : 63) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(YOUNGER < 63 ? YOUNGER : 63) : "memory");
        static_for<0, SPLIT>([&](auto gc) { read_group(gc); });
      }
      swap_group(std::integral_constant<int, g>{});
      if constexpr (VALU) {
        bfly<8, false, 8 * g, 1, 64>(z);             // over e -> ka at position 8g + ka
        static_for<1, 8>([&](auto kac) { constexpr int ka = decltype(kac)::value; z[8 * g + ka] = twid64<g * ka, false>(z[8 * g + ka]); });
      }
      if constexpr (FEN) { xpin8<8 * g, 1>(z); __builtin_amdgcn_sched_barrier(0); }
    });
    wstamp(1);
    if constexpr ((ABL & 32) != 0 && PF > 0) gang_arrive();
    {
      float2 wa[8], wb[8];
      __builtin_amdgcn_sched_barrier(0);           // keep the base loads (and their registers) out of stage 1
      load_twiddles(wa, wb, u);
      if constexpr (!EARLY) {
      static_for<0, 8>([&](auto kac) {                                                        // over g -> kb at position 8kb + ka: k1 = position
        if constexpr (VALU) bfly<8, false, decltype(kac)::value, 8, 64>(z);
        if constexpr (FEN) { xpin8<decltype(kac)::value, 8>(z); __builtin_amdgcn_sched_barrier(0); }
      });
      static_for<1, 64>([&](auto jc) {
        constexpr int j = decltype(jc)::value, ka = j % 8, kb = j / 8;
        if constexpr (VALU && ka > 0) z[j] = cmul(z[j], wa[ka]);
        if constexpr (VALU && kb > 0) z[j] = cmul(z[j], wb[kb]);
        if constexpr (FEN && ka == 7) { xpin8<8 * kb, 1>(z); __builtin_amdgcn_sched_barrier(0); }   // one wb at a time
      });
      } else {
      // column by column: butterfly over g, both twiddle factors, and (from column KB0 on, behind the barrier that says every wave has
      // emptied its landing slots) the real parts straight into the exchange image while the next column is computed
      static_for<0, 8>([&](auto kac) {
        constexpr int ka = decltype(kac)::value;
        if constexpr (VALU) {
          bfly<8, false, ka, 8, 64>(z);
          static_for<0, 8>([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value, j = 8 * kb + ka;
            if constexpr (ka > 0) z[j] = cmul(z[j], wa[ka]);
            if constexpr (kb > 0) z[j] = cmul(z[j], wb[kb]);
          });
        }
        if constexpr (FEN) { xpin8<ka, 8>(z); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (ka == KB0) {
          xp64_barrier();
          if constexpr (XF_LDS == 0) static_for<0, KB0>([&](auto cc) { xp64_write_re_col<decltype(cc)::value>(z, img, p, u); });
        }
        if constexpr (ka >= KB0 && XF_LDS == 0) xp64_write_re_col<ka>(z, img, p, u);
        if constexpr (FEN) __builtin_amdgcn_sched_barrier(0);
      });
      }
    }
    // this tile's gate bins -> LDS.  They were requested behind the previous tile's last stores, so waiting for them means waiting
    // for every store of that tile to be acknowledged: as late as possible (the bins are first read after E1's barriers) — but BEFORE
    // the deferred requests below: hipcc waits for registers that were loaded before the loop's back edge with vmcnt(0), which behind
    // those requests would mean a full HBM round trip at the end of every F1.
    if ((ABL & 32768) == 0 || it == 0) gate_commit();   // ABL bit15 (p64_ab_bench): the first tile's gate for every tile = no wait for the stores
    __builtin_amdgcn_sched_barrier(0);
    // ---- the quiet part of the tile starts: the deferred results of the previous tile leave, the same row groups of the next tile
    //      are requested into the registers they vacate
    if constexpr (PF > 0 && !PF_TOP) pf_block();
    __builtin_amdgcn_sched_barrier(0);
    // ---- E1: position k1 -> image row k1, column (p, u); thread (p, s = u) reads row u, slot n2 ---------------------------
    stamp(it, 2);
    wstamp(2);
    if constexpr (!EARLY) xp64_barrier();           // every wave has emptied its landing slots / finished E2's reads of the previous tile
    xp64_exchange<true, XF_LDS | (EARLY ? 1 : 0) | (EARLY && LATEB ? 4 : 0) | ((XP & 32) != 0 ? 8 : 0)>(z, img, p, u, [&](int k) { wstamp(3 + k); });

    stamp(it, 3);
    wstamp(6);
    // ---- middle: F2 -> gate -> I1 (kernel_regtile.h; bin of register (ka, kb): k = k1 + 64 k2, k2 = ka + 8 kb) -------------
    {
      const int k1 = u;
      if constexpr (VALU) xp64_stageA1<false, FEN>(z);
      auto fetch_gate = [&](int k2, bool upper) -> float2 {
        float2 g = glds[upper ? 64 * (64 - k2) - k1 : k1 + 64 * k2];      // scaled by 1/N, edges fixed, conj applied
        if (upper) g.y = -g.y;                                          // Hermitian extension above N/2
        return g;
      };
      // With deferred / prefetched groups (PF > 0) the gate bins of one register group are fetched right where they are used
      // (16 registers); otherwise the next group's bins are prefetched while this group's butterflies run (32 registers).
      float2 gcur[8], gnxt[FEN ? 1 : 8];
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
        if constexpr (ka + 1 < 8 && !FEN)
          static_for<0, 8>([&](auto kbc) { constexpr int k2n = ka + 1 + 8 * decltype(kbc)::value; gnxt[decltype(kbc)::value] = fetch_gate(k2n, k2n >= 32); });
        if constexpr (VALU) xfftA_stage2_group<8, 8, false, ka>(z);
        static_for<0, 8>([&](auto kbc) {
          constexpr int kb = decltype(kbc)::value, j = 8 * ka + kb, k2 = ka + 8 * kb;
          if constexpr (VALU) z[j] = cmul(z[j], gcur[kb]);                                   // spectre.py:545
          else { z[j].x += gcur[kb].x; z[j].y += gcur[kb].y; }
          if constexpr (WITH_MEM) {                                      // Mf[k] = mem_c[k] + i mem_{c+1}[k] below N/2, conj(mem_c[N-k]) + i conj(mem_{c+1}[N-k]) above
            const float4 m = mcur[kb];
            float2 add;
            if ((k2 == 0 || k2 == 32) && k1 == 0) add = make_float2(m.x, m.z);          // DC, Nyquist: real parts only
            else if (k2 >= 32)                    add = make_float2(m.x + m.w, m.z - m.y);
            else                                  add = make_float2(m.x - m.w, m.y + m.z);
            z[j].x += add.x * inv_n; z[j].y += add.y * inv_n;
          }
        });
        if constexpr (FEN) { xpin8<8 * ka, 1>(z); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (ka + 1 < 8 && FEN)
          static_for<0, 8>([&](auto kbc) {
            constexpr int k2n = ka + 1 + 8 * decltype(kbc)::value;
            gcur[decltype(kbc)::value] = fetch_gate(k2n, k2n >= 32);
            if constexpr (WITH_MEM) mcur[decltype(kbc)::value] = fetch_mem(k2n, k2n >= 32);
          });
        if constexpr (VALU) xfftB_stage1_group<8, 8, true, ka>(z);
        if constexpr (FEN) xpin8<8 * ka, 1>(z);
        if constexpr (ka + 1 < 8 && !FEN) static_for<0, 8>([&](auto kbc) { gcur[decltype(kbc)::value] = gnxt[decltype(kbc)::value]; });
        __builtin_amdgcn_sched_barrier(0);         // keep the gate prefetch one group deep (register budget)
      });
      wstamp(7);
      if constexpr (!EARLY) {
        if constexpr (VALU) xp64_stageB2<true, FEN>(z);              // natural order: position n2
      } else {
        if constexpr (LATEB) xp64_barrier();       // E1's reads are done everywhere: the image may be written again
        static_for<0, 8>([&](auto nc) {
          constexpr int nlo = decltype(nc)::value;
          if constexpr (VALU) bfly<8, true, nlo, 8, 64>(z);
          if constexpr (FEN) { xpin8<nlo, 8>(z); __builtin_amdgcn_sched_barrier(0); }
          if constexpr (XF_LDS == 0) xp64_write_re_col<nlo>(z, img, p, u);
          if constexpr (FEN) __builtin_amdgcn_sched_barrier(0);
        });
      }
    }
    wstamp(8);

    // ---- E2: position n2 -> image row n2, column (p, k1 = u); thread (p, u) reads row u, slot k1 --------------------------
    xp64_exchange<(SPLIT > 0), XF_LDS | (EARLY ? 1 : 0) | ((XP & 32) != 0 ? 8 : 0)>(z, img, p, u, [&](int k) { wstamp(9 + k); });

    stamp(it, 4);
    wstamp(12);
    // ---- the image is idle until the next E1: let the first row groups of the next tile land in it, and fetch its gate -----
    const uint32_t voff = (uint32_t)(((long long)(u + 512 * h) * v_sn + 4 * pp) * ESI);
    // (the twiddles of I2 are read BEFORE the LDS-DMA is issued: hipcc orders every LDS read behind a pending LDS-DMA with
    //  s_waitcnt vmcnt(0) — the whole HBM round trip of the requests below, at the start of every burst)
    float2 wa2[8], wb2[8];
    load_twiddles(wa2, wb2, u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    static_for<0, SPLIT>([&](auto gc) { dma_group(rs_next, dma_voff(voff, v_sn), v_sn, gc); });
    asm volatile("" ::: "memory");                 // the vmcnt() above counts on these being older than every store below
    // ABL bit16 (p64_ab_bench): the next tile's gate bins requested BEFORE the stores of this one (10 registers through I2), so that
    // gate_commit waits for requests older than the burst's stores — measured: no difference, the default stays behind the stores
    if constexpr ((ABL & 32768) == 0 && (ABL & 65536) != 0) gate_fetch(gpn);
    asm volatile("" ::: "memory");

    // ---- conj twiddle, I2, stores (spectre.py:553) interleaved with the loads that refill the released registers -----------
    {
      float2 (&wa)[8] = wa2, (&wb)[8] = wb2;
      static_for<1, 64>([&](auto jc) {
        constexpr int j = decltype(jc)::value, ja = j % 8, jb = j / 8;   // position j carries k1 = j
        if constexpr (VALU && ja > 0) z[j] = cmulc(z[j], wa[ja]);
        if constexpr (VALU && jb > 0) z[j] = cmulc(z[j], wb[jb]);
        if constexpr (FEN && ja == 7) { xpin8<8 * jb, 1>(z); __builtin_amdgcn_sched_barrier(0); }
      });
      if constexpr (FEN) __builtin_amdgcn_sched_barrier(0);
      if constexpr (VALU) xp64_stageA1<true, FEN>(z);
    }
    wstamp(13);
    {
      const uint32_t ooff = (uint32_t)(((long long)(u + 512 * h) * out_sn + 4 * pp) * ESO);
      static_for<0, 8>([&](auto ic) {
        constexpr int g = (decltype(ic)::value + SPLIT) % 8;             // register-loaded groups first: their reloads start earliest
        constexpr bool stores_now = g < GP && (ABL & 262144) == 0 && ((ABL & 64) == 0 || decltype(ic)::value % 2 == 0) && ((ABL & 128) == 0 || decltype(ic)::value % 4 == 0);                               // (deferred groups are stored before the next E1)
        if constexpr (stores_now) gang_arrive();
        if constexpr (VALU) xfftA_stage2_group<8, 8, true, g>(z);                              // rows g + 8e at positions 8g + e
        if constexpr (FEN) xpin8<8 * g, 1>(z);
        swap_group(std::integral_constant<int, g>{});
        if constexpr (stores_now) gang_await(members);
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
        if constexpr (FEN) __builtin_amdgcn_sched_barrier(0);
      });
    }
    obp = ob;
    if constexpr ((ABL & 32768) == 0 && (ABL & 65536) == 0) gate_fetch(gpn);   // committed to LDS at the end of the next tile's F1 (after the
                                                   // last tile: a harmless re-read of this tile's bins)
    stamp(it, 5);
    wstamp(14);
    if constexpr ((XP & 1) != 0) {                 // flush this wave's stamps (one more VMEM store younger than the LDS-DMA: the vmcnt() above only gets stricter)
      reinterpret_cast<unsigned*>(a.trace)[(((size_t)blockIdx.x * a.tpw + it) * 8 + (tid0 >> 6)) * 16 + (tid0 & 15)] = wst[tid0 & 15];
    }
  }  // tile loop
  if constexpr ((ABL & 32) != 0) {
    if (gcnt != nullptr && (tid0 & 63) == 0) { gcnt[1] = gang_polls; gcnt[2] = gang_live ? 1u : 0u; }
  }
}


}  // namespace sfft
