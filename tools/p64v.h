// p64v.h - round-4 experimental copy of fft_amd/csrc/kernel_regtile64p.h (same code, renamed symbols) with measurement knobs as template
// parameters, timed against the library kernel by tools/p64v_bench.hip.  Nothing in the product includes this file.
//   GANGX  workgroups that walk through adjacent tiles in step (library: 2 = one 128-byte line per row)
//   AUXD / AUXL / AUXS  cache-policy bits (gfx940 encoding: 1 = sc0, 2 = nt, 16 = sc1) of the LDS-DMA requests, the register loads, the stores
//   MAPX   1 = compact window: tile = it * n_wg + wg (all workgroups in adjacent tiles at any time)
//   SYNCP  the workgroups of a gang MEET once per tile through a device-scope counter (a.mem = counter buffer, one word per gang, zeroed before
//          the launch): 1 = behind E2 (in front of the LDS-DMA burst and the stores of I2), 2 = at the end of F1 (in front of the deferred stores)
//   EARLY1 (round 5) 1 = stage 1 of the DEFERRED groups of the next tile (their rows trade places with the results inside the store burst)
//          right behind the burst, in front of the reload requests — arithmetic while the memory pipeline drains the stores — instead of at the
//          top of the next tile; 2 = behind the reload requests (control: the same code motion without the drain)
//   SKEW   (round 5) wave w idles w * SKEW x `s_nop 7` at the start of the three phases that carry spread requests (behind the burst,
//          in front of the middle phase, in front of I2): the eight waves reach every request slot at the same clock and meet at the CU's one
//          memory path (16 clocks per 1-KiB request) — does a stagger of the whole instruction streams pay for itself?
//   PRIO   0 none; 1 = s_setprio 1 for waves 4..7 (static); 2 = s_setprio 1 for waves 0..3; 3 = (round 6) s_setprio 3 around every SPREAD request
//          (deferred stores / loads, LDS-DMA), 0 for the butterflies; 4 = (round 6) s_setprio 3 from the barrier in front of the store burst to the
//          back edge (burst, reloads, gate fetch), 0 elsewhere
#pragma once
#include "../fft_amd/csrc/kernel_regtile.h"

namespace sfft {

constexpr int kV64ImageBytes = regtile_image_bytes<64, 64, 1>();
// LDS: exchange image | half-spectrum gate | the two twiddle vectors of every team index u (W^(u j), W^(8 u j), j = 1..7): 64 x 14 x 8 B.
// The twiddles are read twice per tile; as global loads they would queue (one in-order vmcnt) behind the LDS-DMA requests of the next
// tile and behind the last stores.  From LDS they cost 7 ds_read_b128 and no vmcnt.
constexpr int kV64TwOff = (regtile_lds_total<64, 64, 1>() + 15) & ~15;
constexpr int kV64LdsTotal = kV64TwOff + 64 * 14 * 8;
static_assert(kV64LdsTotal <= 160 * 1024, "LDS budget");

typedef unsigned int pv_u32x4 __attribute__((ext_vector_type(4)));
typedef float pv_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kV64RsrcFlags = 0x00020000;        // raw buffer, 32-bit data format (gfx90a / gfx942 / gfx950 dword 3)

__device__ __forceinline__ void vlane16_swap(float& x, float& y) {   // x of the odd 16-lane rows <-> y of the even rows
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}

// The 8 x 8 in-register transforms of fft_regs.h with a scheduling fence after every radix-8 butterfly: hipcc otherwise interleaves
// all eight butterflies of a stage (up to 120 temporaries on top of the 128 data registers), and with the deferred-result registers
// on top it parks those in scratch.  vpin8<BASE, STRIDE>: the eight values z[BASE + STRIDE j] have to exist in registers HERE.
// sched_barrier only fences the machine scheduler; the IR-level sinking pass still splits a butterfly and leaves half-finished sums
// alive until their first use hundreds of instructions later (that, not the schedule, is where 250-register peaks come from).
template <int BASE, int STRIDE>
__device__ __forceinline__ void vpin8(float2 (&z)[64]) {
  asm volatile("" : "+v"(z[BASE].x), "+v"(z[BASE].y), "+v"(z[BASE + STRIDE].x), "+v"(z[BASE + STRIDE].y),
                    "+v"(z[BASE + 2 * STRIDE].x), "+v"(z[BASE + 2 * STRIDE].y), "+v"(z[BASE + 3 * STRIDE].x), "+v"(z[BASE + 3 * STRIDE].y),
                    "+v"(z[BASE + 4 * STRIDE].x), "+v"(z[BASE + 4 * STRIDE].y), "+v"(z[BASE + 5 * STRIDE].x), "+v"(z[BASE + 5 * STRIDE].y),
                    "+v"(z[BASE + 6 * STRIDE].x), "+v"(z[BASE + 6 * STRIDE].y), "+v"(z[BASE + 7 * STRIDE].x), "+v"(z[BASE + 7 * STRIDE].y));
}

__device__ __forceinline__ void vpin4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
template <bool INV, class CB>
__device__ __forceinline__ void p64v_stageA1_cb(float2 (&z)[64], CB cb) {
  static_for<0, 8>([&](auto q0c) {
    constexpr int q0 = decltype(q0c)::value;
    bfly_plain<8, INV, q0, 8, 64>(z);
    vpin8<q0, 8>(z);
    __builtin_amdgcn_sched_barrier(0);
    cb(q0c);
    __builtin_amdgcn_sched_barrier(0);
  });
}
template <bool INV>
__device__ __forceinline__ void p64v_stageA1(float2 (&z)[64]) {      // type A stage 1: radix-8 over q1 (positions 8 q1 + q0); W_64^(q0 ka) is stage 2's
  static_for<0, 8>([&](auto q0c) {
    constexpr int q0 = decltype(q0c)::value;
    bfly_plain<8, INV, q0, 8, 64>(z);
    vpin8<q0, 8>(z);
    __builtin_amdgcn_sched_barrier(0);
  });
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier, and hipcc implements the
// fence with s_waitcnt vmcnt(0): every barrier of the exchanges would drain the deferred stores and the prefetches that are meant to
// travel DURING the exchanges.  Nothing that crosses waves goes through global memory here (a lane reads back only what its own wave
// requested by LDS-DMA, after its own vmcnt wait), so the LDS counter is all a barrier has to wait for.
__device__ __forceinline__ void p64v_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool IN_BF16> constexpr int kV64Gang = IN_BF16 ? 4 : 2;   // workgroups per 128-byte line (launch: n_wg is a multiple of it)

// One exchange = position j of thread (p, u) -> image row j, column (p, u); thread (p, u) then reads row u, slots 0..63.  One float
// plane at a time (the tile is 256 KiB, the image 136 KiB).  The scattered dword writes are ds_write2st64_b32 (the LDS takes a store's
// address and data registers at 2 cycles per dword: 6 cycles for two dwords instead of 8 for two ds_write_b32); the instruction's two
// 8-bit offsets count units of 256 bytes, and rows 8 apart are 8 * 2176 = 68 * 256 bytes apart.
//
// p64v_write_col<KA, IM>: the real (imaginary) parts of positions KA + 8 kb, kb = 0..7 -> rows KA + 8 kb: four ds_write2st64_b32 with two
// opaque base addresses (kb < 4, kb >= 4: the offsets reach 3 * 68 units).  Called by the PRODUCER for the real plane right after the
// butterfly that finished those eight positions, so the writes travel while the next butterfly runs.
template <int KA, bool IM>
__device__ __forceinline__ void p64v_write_col(float2 (&z)[64], float* img, int p, int u) {
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
struct P64vNoCb { template <class T> __device__ __forceinline__ void operator()(T) const {} };
template <bool LAST_BARRIER, class CB = P64vNoCb>
__device__ __forceinline__ void p64v_exchange_rest(float2 (&z)[64], float* img, int p, int u, CB cb = CB{}) {
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
  p64v_barrier();
  cb(std::integral_constant<int, 0>{});
  read_plane(std::false_type{});
  p64v_barrier();
  cb(std::integral_constant<int, 1>{});
  static_for<0, 8>([&](auto cc) { p64v_write_col<decltype(cc)::value, true>(z, img, p, u); });
  p64v_barrier();
  cb(std::integral_constant<int, 2>{});
  read_plane(std::true_type{});
  if constexpr (LAST_BARRIER) p64v_barrier();       // image free again
}

// VMEM instructions a wave issues between its last LDS-DMA request of a burst and the s_waitcnt that guards the landing slots at the
// top of the next tile (completion is in order, so vmcnt(N) with N = that count means "everything up to and including the LDS-DMA has
// landed").  Steady state: per reloaded group 4 stores + 4 loads, per LDS-staged group 4 stores, the 5 gate loads.  First tile (the
// prologue): the 4 loads of every register-loaded group and the 5 gate loads.  tools/isa_lint.py checks both against the ISA.
template <int SPLIT, int PF, int LATE = 0, int PARK = 0> constexpr int p64v_younger() { return 8 * (8 - PF - SPLIT - LATE) + 4 * LATE + 4 * SPLIT + 5 - 4 * PARK; }
template <int SPLIT> constexpr int p64v_younger_first() { return 4 * (8 - SPLIT) + 5; }

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
template <int SPLIT, int PF, bool WITH_MEM = false, bool IN_BF16 = false, bool OUT_BF16 = false, int GANGX = 0, int AUXD = 0, int AUXL = 0, int AUXS = 0, int PRIO = 0, int MAPX = 0, int DSPREAD = 0, int SYNCP = 0, int TSTAMP = 0, int PFL2 = 0, int STAG = 0, int PFSP = 0, int LATE = 0, int RLF = 0, int PARK = 0, int ESPREAD = 1, int EARLY1 = 0, int SKEW = 0, int MEET = 0, int LAG = 0>
__global__ void __launch_bounds__(512, 2) spectre_mix_p64v(const RegtileArgs a) {
  constexpr int ESI = IN_BF16 ? 2 : 4, ESO = OUT_BF16 ? 2 : 4;   // bytes per input / output element
  constexpr float inv_n = 1.0f / 4096.0f;
  constexpr int GROUP_SLOT = IN_BF16 ? 2 * 1024 : 4 * 1024;     // bytes of one row group in a wave's landing slots
  static_assert(SPLIT >= 1 && SPLIT <= 8 && SPLIT * GROUP_SLOT * 8 <= kV64ImageBytes, "staging lives in the exchange image");
  static_assert(PF >= 0 && SPLIT + PF <= 8, "row groups: SPLIT through LDS, PF deferred / prefetched in registers, the rest behind their stores");
  constexpr int GP = 8 - PF;                       // first deferred / prefetched group
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);
  float2* glds = reinterpret_cast<float2*>(smem + kV64ImageBytes);
  float2* twl = reinterpret_cast<float2*>(smem + kV64TwOff);
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
  [[maybe_unused]] auto wskew = [&]() {
    if constexpr (SKEW > 0) {
      const int n = __builtin_amdgcn_readfirstlane(tid0 >> 6) * SKEW;
      for (int i = 0; i < n; ++i) asm volatile("s_nop 7");
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (PRIO == 1) { if (__builtin_amdgcn_readfirstlane(tid0 >> 6) >= 4) __builtin_amdgcn_s_setprio(1); }
  if constexpr (PRIO == 2) { if (__builtin_amdgcn_readfirstlane(tid0 >> 6) < 4) __builtin_amdgcn_s_setprio(1); }

  // GANG neighbouring workgroups (same L2) walk through GANG adjacent tiles in step = one 128-byte line per row: the L2 fetches a
  // line once and the neighbours' requests hit (fp32: two 64-byte halves; bf16: four 32-byte quarters)
  constexpr int GANG = GANGX > 0 ? GANGX : kV64Gang<(IN_BF16 || OUT_BF16)>;
  const int wg_lin = xcd_contiguous(blockIdx.x, a.n_wg);
  // MAPX = 1: compact window — iteration `it` of the whole grid covers tiles [it * n_wg, (it + 1) * n_wg)
  // MAPX = 2: compact window with neighbouring PAIRS on different XCDs (the two workgroups of a pair, blocks b and b + 8, share one)
  const int bx = blockIdx.x;
  // MAPX = 3: DYNAMIC tickets, one counter per XCD (a.mem = counters, 16 words apart, zeroed before the launch): XCD x owns tiles
  // [x * n_tiles / 8, (x + 1) * n_tiles / 8) and its workgroups take them in order, one ticket per tile, drawn a tile and a half ahead by wave 0
  // with a SCALAR atomic (lgkmcnt, not the in-order vmcnt) and handed to the other waves through one LDS word behind the tile's barriers.
  // Adjacent tickets = the two halves of a line go to the two workgroups of the XCD that arrive next to each other in time.
  [[maybe_unused]] const int per_xcd = a.n_tiles / 8, xcd_base = (bx % 8) * per_xcd;
  [[maybe_unused]] unsigned* tick_cnt = reinterpret_cast<unsigned*>(const_cast<float*>(a.mem)) + 16 * (bx % 8);
  [[maybe_unused]] volatile unsigned* tick_lds = reinterpret_cast<volatile unsigned*>(smem + kV64LdsTotal);
  [[maybe_unused]] int cur_t = 0, nxt_t = 0;
  [[maybe_unused]] unsigned long long t_start = 0;
  if constexpr (TSTAMP) t_start = __builtin_amdgcn_s_memrealtime();
  [[maybe_unused]] unsigned long long c_start = 0;
  if constexpr (TSTAMP) c_start = __builtin_readcyclecounter();
  // TSTAMP = 2: wave 0 adds up, over its tiles, the time between phase marks (s_memrealtime, 100 MHz; every mark waits for lgkmcnt(0))
  [[maybe_unused]] unsigned long long ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_last = 0;
  [[maybe_unused]] auto mark = [&](int k) {
    if constexpr (TSTAMP == 2) {
      const unsigned long long t = __builtin_amdgcn_s_memrealtime();
      ph_acc[k] += t - ph_last; ph_last = t;
    }
  };
  if constexpr (MAPX == 3) {
    if (tid0 == 0) { tick_lds[0] = atomicAdd(tick_cnt, 1u); tick_lds[1] = atomicAdd(tick_cnt, 1u); }
    __syncthreads();
    cur_t = __builtin_amdgcn_readfirstlane((int)tick_lds[0]); nxt_t = __builtin_amdgcn_readfirstlane((int)tick_lds[1]);
    __syncthreads();
  }
  // MAPX = 4 (round 5): DYNAMIC tickets per PAIR.  One ticket = the two adjacent tiles (2 T, 2 T + 1) of the XCD's eighth of the tensor = both
  // halves of every line; the pair's LEADER (even workgroup) draws it from the XCD's counter with a scalar atomic two tiles ahead and publishes
  // it in the pair's mailbox (a.mem + 10240 words, 8 tagged slots per pair: (sequence + 1) << 16 | ticket; scalar atomic swap, same L2), the
  // FOLLOWER reads the slot with a scalar load behind E1 and looks at it at the barrier of the middle phase (half a tile of slack).  The chip-wide
  // window is then 8 x 16 adjacent tile pairs instead of 128 regions 96 tiles apart, and the two halves of a line stay with two workgroups in step.
  [[maybe_unused]] const int member4 = wg_lin & 1;
  [[maybe_unused]] unsigned* mbox4 = reinterpret_cast<unsigned*>(const_cast<float*>(a.mem)) + 10240 + 8 * (wg_lin >> 1);   // (behind the TSTAMP areas)
  [[maybe_unused]] const int per_xcd4 = MAPX == 5 ? a.n_tiles / 2 : a.n_tiles / 16;   // MAPX = 5: ONE counter for the chip (a.mem must then be device-coherent for scalar atomics: uncached memory)
  [[maybe_unused]] const int base4 = MAPX == 5 ? 0 : xcd_base;
  if constexpr (MAPX == 5) tick_cnt = reinterpret_cast<unsigned*>(const_cast<float*>(a.mem));
  // ESPREAD (round 5, batch 34): consecutive tickets cycle over ESPREAD batch elements instead of walking through the columns of one
  // (1 = address order: ~5 elements open; 128 = every pair in its own element, like the static map but handed out dynamically)
  [[maybe_unused]] auto remap4 = [&](int t) -> int {
    if constexpr (ESPREAD <= 1) return t;
    const int cols = a.tiles_per_row / 2, per = cols * ESPREAD;
    const int blk = t / per, w = t - blk * per;
    return (blk * ESPREAD + w % ESPREAD) * cols + w / ESPREAD;
  };
  if constexpr (MAPX == 4 || MAPX == 5) {
    if (tid0 == 0) {
      unsigned t0, t1;
      if (member4 == 0) {
        t0 = atomicAdd(tick_cnt, 1u); t1 = atomicAdd(tick_cnt, 1u);
        __hip_atomic_store(mbox4 + 0, (1u << 16) | t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mbox4 + 1, (2u << 16) | t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        t0 = t1 = 0xffffu;
        for (int i = 0; i < (1 << 20); ++i) { const unsigned w = __hip_atomic_load(mbox4 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((w >> 16) == 1u) { t0 = w & 0xffffu; break; } __builtin_amdgcn_s_sleep(4); }
        for (int i = 0; i < (1 << 20); ++i) { const unsigned w = __hip_atomic_load(mbox4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((w >> 16) == 2u) { t1 = w & 0xffffu; break; } __builtin_amdgcn_s_sleep(4); }
      }
      tick_lds[0] = t0; tick_lds[1] = t1;
    }
    __syncthreads();
    cur_t = __builtin_amdgcn_readfirstlane((int)tick_lds[0]); nxt_t = __builtin_amdgcn_readfirstlane((int)tick_lds[1]);
    __syncthreads();
  }
  // LAG (round 6, third session): the pair's FOLLOWER (odd workgroup) starts LAG x 64 clocks late and stays that far behind for the whole launch — its
  // load requests then find their lines fetched by the leader's (one request per line in the L2's miss path instead of two, tools/fold_lab.hip), at the
  // price of its half-line stores arriving that much later than the leader's
  if constexpr (LAG > 0) { if (wg_lin & 1) __builtin_amdgcn_s_sleep(LAG); }
  const int pair_base = (MAPX == 4 || MAPX == 5) ? base4 + 2 * remap4(cur_t) + member4 : MAPX == 3 ? xcd_base + cur_t : MAPX == 2 ? ((bx / 16) * 8 + bx % 8) * 2 + (bx / 8) % 2 : MAPX ? wg_lin : (wg_lin / GANG) * a.tpw * GANG + (wg_lin % GANG);
  const int TS = MAPX ? a.n_wg : GANG;
  if ((MAPX == 4 || MAPX == 5) ? cur_t >= per_xcd4 : MAPX == 3 ? cur_t >= per_xcd : pair_base >= a.n_tiles) return;
  // SYNCP: wave 0 announces the workgroup (one atomic add, no return value: older than every request the hand-counted waits look at) and
  // polls the gang's counter with SCALAR loads (lgkmcnt, not the in-order vmcnt the stores sit in); the other waves wait at a barrier.
  // Performance only: the spin is bounded, and a gang whose partner never shows up stops waiting after the first time-out.
  [[maybe_unused]] unsigned nsync = 0;
  [[maybe_unused]] bool sync_on = true;
  [[maybe_unused]] auto gang_meet = [&]() {
    unsigned* cp = reinterpret_cast<unsigned*>(const_cast<float*>(a.mem)) + (MAPX >= 4 ? 12288 + 16 * (wg_lin >> 1) : MAPX ? 0 : wg_lin / GANG);   // (pair tickets: the pair is static, only its tiles are drawn)
    ++nsync;
    if (__builtin_amdgcn_readfirstlane(tid0 >> 6) == 0) {
      if (lane == 0) __hip_atomic_fetch_add(cp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (sync_on) {
        int i = 0;
        for (; i < 128; ++i) {
          unsigned val;
          asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(val) : "s"(cp) : "memory");
          if (val >= (unsigned)GANG * nsync) break;
          if constexpr (SYNCP != 7) __builtin_amdgcn_s_sleep(2);
        }
        if (i == 128) sync_on = false;
      }
    }
    p64v_barrier();
  };

  // MEET (round 6, third session): the pair's rendezvous in front of the STORE BURST under the ticket order — 1: arrive + poll there (gang_meet
  // in place of the burst's barrier); 2: arrive behind E2 (one fire-and-forget atomic, ~2 us earlier), poll in front of the burst
  [[maybe_unused]] auto gang_arrive = [&]() {
    unsigned* cp = reinterpret_cast<unsigned*>(const_cast<float*>(a.mem)) + 12288 + 16 * (wg_lin >> 1);
    ++nsync;
    if (__builtin_amdgcn_readfirstlane(tid0 >> 6) == 0 && lane == 0) __hip_atomic_fetch_add(cp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  [[maybe_unused]] auto gang_wait = [&]() {
    unsigned* cp = reinterpret_cast<unsigned*>(const_cast<float*>(a.mem)) + 12288 + 16 * (wg_lin >> 1);
    if (__builtin_amdgcn_readfirstlane(tid0 >> 6) == 0 && sync_on) {
      int i = 0;
      for (; i < 128; ++i) {
        unsigned val;
        asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(val) : "s"(cp) : "memory");
        if (val >= (unsigned)GANG * nsync) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (i == 128) sync_on = false;
    }
    p64v_barrier();
  };

  float2 z[64];
  float4 dfr[PF > 0 ? 4 * PF : 1];                 // deferred results of the previous tile / prefetched rows of the next one
  static_for<0, 4 * PF>([&](auto ic) { dfr[decltype(ic)::value] = make_float4(0.f, 0.f, 0.f, 0.f); });   // (stored into an empty range before the first tile)
  // LATE row groups [GP - LATE, GP): PREFETCHED like the deferred ones (spare registers, requested in the quiet middle of the tile) but
  // stored with the burst: no row group is reloaded behind its store any more when SPLIT + PF + LATE = 8
  static_assert(LATE >= 0 && SPLIT + PF + LATE <= 8, "row groups");
  float4 lat[LATE > 0 ? 4 * LATE : 1];
  static_for<0, (LATE > 0 ? 4 * LATE : 1)>([&](auto ic) { lat[decltype(ic)::value] = make_float4(0.f, 0.f, 0.f, 0.f); });
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
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vb), 0, (int)((long long)rows * sn * ESI), kV64RsrcFlags);
  };
  auto rsrc_out = [&](char* ob, long long sn, bool live = true) {
    const int rows = live ? a.rows_out : 0;
    return __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)((long long)rows * sn * ESO), kV64RsrcFlags);
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
        const pv_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (uint32_t)((64 * g + 1024 * m) * sn * 4), 0, AUXL);
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
                                                 voff + (uint32_t)((64 * g + 1024 * m) * sn * 4), 0, 0, AUXD);
      });
    }
  };
  auto store16 = [&](__amdgpu_buffer_rsrc_t rs, uint32_t off, const float4 v) {   // this lane's 4 channels of one row
    if constexpr (OUT_BF16) {
      rt_u32x2 t;
      t.x = f32_to_bf16_rne(v.x) | (f32_to_bf16_rne(v.y) << 16); t.y = f32_to_bf16_rne(v.z) | (f32_to_bf16_rne(v.w) << 16);
      __builtin_amdgcn_raw_buffer_store_b64(t, rs, off, 0, 0);
    } else {
      pv_u32x4 t;
      t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y); t.z = __float_as_uint(v.z); t.w = __float_as_uint(v.w);
      __builtin_amdgcn_raw_buffer_store_b128(t, rs, off, 0, AUXS);
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
      vlane16_swap(z[j].x, z[j + 1].x);
      vlane16_swap(z[j].y, z[j + 1].y);
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
    // RLF: the row groups that used to be reloaded behind the stores, and the gate bins, are requested at the START of their own tile, between
    // the butterflies of the deferred groups (which are already there) — not at the end of the previous one behind its store burst
    if constexpr (RLF) static_for<GP, 8>([&](auto gc) { load_group(rs, voff, a.v_sn, gc); });
    else {
      static_for<SPLIT, 8>([&](auto gc) { load_group(rs, voff, a.v_sn, gc); });
      gate_fetch(gp);
    }
  }

  for (int it = 0; MAPX == 3 || MAPX == 4 || MAPX == 5 || it < a.tpw; ++it) {
    const int tile = (MAPX == 4 || MAPX == 5) ? base4 + 2 * remap4(cur_t) + member4 : MAPX == 3 ? xcd_base + cur_t : pair_base + TS * it;
    if ((MAPX == 4 || MAPX == 5) ? cur_t >= per_xcd4 : MAPX == 3 ? cur_t >= per_xcd : tile >= a.n_tiles) break;                  // workgroup-uniform
    const bool more = (MAPX == 4 || MAPX == 5) ? nxt_t < per_xcd4 : MAPX == 3 ? nxt_t < per_xcd : (it + 1 < a.tpw) && (tile + TS < a.n_tiles);
    coords();
    if constexpr (TSTAMP == 2) { if (it == 0) ph_last = __builtin_amdgcn_s_memrealtime(); }
    if constexpr (PRIO == 4) __builtin_amdgcn_s_setprio(0);
    mark(0);                                       // (gate fetch of the previous iteration .. here)
    [[maybe_unused]] unsigned fut = 1;
    if constexpr (MAPX == 3) {                     // the ticket of the tile after next: requested now, in the LDS word behind F1's barrier
      if (__builtin_amdgcn_readfirstlane(tid0 >> 6) == 0) asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(fut) : "s"(tick_cnt) : "memory");
    }
    [[maybe_unused]] unsigned* slot4 = mbox4 + ((it + 2) & 7);
    if constexpr (MAPX == 4 || MAPX == 5) {        // the leader draws the pair's ticket of the tile after next
      if (__builtin_amdgcn_readfirstlane(tid0 >> 6) == 0 && member4 == 0) asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(fut) : "s"(tick_cnt) : "memory");
    }
    long long v_sn = a.v_sn, out_sn = a.out_sn;
    asm volatile("" : "+s"(v_sn), "+s"(out_sn));
    const char* vb; char* ob; const float2* gp;
    tile_ptrs(tile, vb, ob, gp);
    const char* vbn = vb; char* obn = ob; const float2* gpn = gp;
    if (more) tile_ptrs((MAPX == 4 || MAPX == 5) ? base4 + 2 * remap4(nxt_t) + member4 : MAPX == 3 ? xcd_base + nxt_t : tile + TS, vbn, obn, gpn);
    const __amdgpu_buffer_rsrc_t rs_next = rsrc_in(vbn, v_sn, more), rs_out = rsrc_out(ob, out_sn);
    // PFL2: touch the half lines of the row groups that will be RELOADED behind this tile's stores (one dword per row, LDS-DMA into a dump
    // word area: no register, tracked by vmcnt like every other request) so that the reloads find their lines in the L2.  Wave w covers the
    // 64 rows of its (h, m) = (w & 1, w >> 1) per group.  1 = right in front of the LDS-DMA burst, 2 = in the quiet part behind the deferred loads,
    // 3 = both row-group sets (also the LDS-staged groups) in the quiet part
    [[maybe_unused]] auto l2_touch = [&](auto gc) {
      constexpr int g = decltype(gc)::value;
      const int w = __builtin_amdgcn_readfirstlane(tid0 >> 6);
      const uint32_t off = (uint32_t)((long long)(64 * g + lane + 512 * (w & 1) + 1024 * (w >> 1)) * v_sn * ESI);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_next, (__attribute__((address_space(3))) void*)(smem + kV64LdsTotal + 16), 4, off, 0, 0, 0);
    };

    [[maybe_unused]] const uint32_t pf_ooff = (uint32_t)(((long long)(u + 512 * h) * out_sn + 4 * pp) * ESO);
    [[maybe_unused]] const uint32_t pf_voff = (uint32_t)(((long long)(u + 512 * h) * v_sn + 4 * pp) * ESI);
    [[maybe_unused]] auto pf_store = [&](auto ic) {
      constexpr int g = GP + decltype(ic)::value / 4, m = decltype(ic)::value % 4;
      if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(3);
      store16(rsrc_out(obp, out_sn, it > 0), pf_ooff + (uint32_t)((64 * g + 1024 * m) * out_sn * ESO), dfr[decltype(ic)::value]);
      if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(0);
    };
    [[maybe_unused]] auto pf_load = [&](auto ic) {
      constexpr int g = GP + decltype(ic)::value / 4, m = decltype(ic)::value % 4;
      if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(3);
      if constexpr (IN_BF16) {                       // stays packed (two dwords) until it trades places with the results in I2
        const rt_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs_next, pf_voff + (uint32_t)((64 * g + 1024 * m) * v_sn * ESI), 0, 0);
        dfr[decltype(ic)::value].x = __uint_as_float(t.x); dfr[decltype(ic)::value].y = __uint_as_float(t.y);
      } else {
        const pv_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_next, pf_voff + (uint32_t)((64 * g + 1024 * m) * v_sn * 4), 0, AUXL);
        dfr[decltype(ic)::value] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
      }
      if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(0);
    };
    [[maybe_unused]] auto lat_load = [&](auto ic) {
      constexpr int g = GP - LATE + decltype(ic)::value / 4, m = decltype(ic)::value % 4;
      static_assert(!IN_BF16 || LATE == 0, "fp32 rows");
      const pv_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_next, pf_voff + (uint32_t)((64 * g + 1024 * m) * v_sn * 4), 0, AUXL);
      lat[decltype(ic)::value] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
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
        // staged groups are in the slots (p64v_younger; checked against the ISA by tools/isa_lint.py)
        if (it == 0) asm volatile("s_waitcnt vmcnt(%0) ; lint: first" :: "n"(p64v_younger_first<SPLIT>() + 4 * PARK) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) ; lint: steady" :: "n"(p64v_younger<SPLIT, PF, LATE, PARK>()) : "memory");
        mark(1);                                   // stage 1 of the deferred groups, wait for the LDS-DMA
        static_for<0, SPLIT>([&](auto gc) { read_group(gc); });
      }
      if constexpr (PARK != 0 && i < PF) {
        // PARK: the results of the first reloaded group of the PREVIOUS tile were parked in the image behind the landing slots (own-wave area,
        // untracked asm LDS accesses: hipcc orders tracked ones behind the pending LDS-DMA with vmcnt(0)); they leave now, before the wait below
        static_for<0, 4>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          if constexpr (m % PF == i || (PF > 4)) {
            pv_f32x4 rv;
            const uint32_t pa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + 8 * SPLIT * GROUP_SLOT + (__builtin_amdgcn_readfirstlane(tid0 >> 6) * 4 + m) * 1024 + lane * 16);
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rv) : "v"(pa) : "memory");
            store16(rsrc_out(obp, out_sn, it > 0), pf_ooff + (uint32_t)((64 * SPLIT + 1024 * m) * out_sn * ESO), make_float4(rv.x, rv.y, rv.z, rv.w));
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (RLF != 0 && i < PF) {
        constexpr int R = GP - SPLIT;                                        // groups that are neither staged nor deferred
        const __amdgpu_buffer_rsrc_t rs_cur = rsrc_in(vb, v_sn);
        static_for<0, R>([&](auto rc) {
          if constexpr (decltype(rc)::value % PF == i) load_group(rs_cur, pf_voff, v_sn, std::integral_constant<int, SPLIT + decltype(rc)::value>{});
        });
        if constexpr (i == PF - 1) gate_fetch(gp);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (i == PF + SPLIT && TSTAMP == 2) { vpin8<8 * ((PF + SPLIT - 1) < PF ? GP + PF + SPLIT - 1 : SPLIT - 1), 1>(z); mark(2); }   // staged groups read + stage 1
      if (EARLY1 == 0 || i >= PF || it == 0) {      // (EARLY1: done at the end of the previous tile, except for the tile the prologue loaded)
      swap_group(std::integral_constant<int, g>{});
      bfly_plain<8, false, 8 * g, 1, 64>(z);       // over e -> ka at position 8g + ka (W_64^(g ka) is applied by the column butterflies below)
      vpin8<8 * g, 1>(z);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PFSP == 6 && i >= PF) { static_for<i * (4 * PF) / 16, (i + 1) * (4 * PF) / 16>([&](auto ic2) { pf_store(ic2); }); __builtin_amdgcn_sched_barrier(0); }
    });
    if constexpr (PFSP == 6) static_for<0, PF * (4 * PF) / 16>([&](auto ic2) { pf_store(ic2); });   // (the shares of the first PF slots: behind the wait for the LDS-DMA they would be counted as younger)
    if constexpr (TSTAMP == 2) { vpin8<8 * (GP - 1), 1>(z); mark(3); }   // wait for the reloaded groups + their stage 1
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
        vpin8<ka, 8>(z);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PFSP == 2 || PFSP == 4 || PFSP == 5 || PFSP == 7 || PFSP == 8) { static_for<ka * (4 * PF) / 8, (ka + 1) * (4 * PF) / 8>([&](auto ic) { pf_store(ic); }); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (PFSP == 6) { static_for<(8 + ka) * (4 * PF) / 16, (9 + ka) * (4 * PF) / 16>([&](auto ic) { pf_store(ic); }); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (ka == 0) p64v_barrier();      // every wave has emptied its landing slots (and finished E2's reads of the previous
                                                   // tile): the image may be written
        if constexpr (ka == 0 && MAPX == 3) {       // (the barrier waited for lgkmcnt(0): the scalar atomic has returned)
          if (__builtin_amdgcn_readfirstlane(tid0 >> 6) == 0) { asm volatile("" : "+s"(fut)); if (lane == 0) tick_lds[0] = fut; }
        }
        if constexpr (ka == 0 && (MAPX == 4 || MAPX == 5)) {       // leader: publish the ticket (fire and forget) and hand it to the other waves
          if (__builtin_amdgcn_readfirstlane(tid0 >> 6) == 0 && member4 == 0) {
            asm volatile("" : "+s"(fut));
            unsigned pub = ((unsigned)(it + 3) << 16) | (fut & 0xffffu);
            asm volatile("s_atomic_swap %0, %1, 0x0" :: "s"(pub), "s"(slot4) : "memory");
            if (lane == 0) tick_lds[0] = fut;
          }
        }
        p64v_write_col<ka, false>(z, img, p, u);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    // this tile's gate bins -> LDS.  They were requested behind the previous tile's last stores, so waiting for them means waiting
    // for every store of that tile to be acknowledged: as late as possible (the bins are first read after E1's barriers) — but BEFORE
    // the deferred requests below: hipcc waits for registers that were loaded before the loop's back edge with vmcnt(0), which behind
    // those requests would mean a full HBM round trip at the end of every F1.
    mark(4);                                       // F1 stage 2 + twiddles + first barrier + real-plane writes
    gate_commit();
    __builtin_amdgcn_sched_barrier(0);
    mark(5);                                       // gate commit (waits for the gate loads = for the previous tile's stores)
    // ---- the quiet part of the tile starts: the deferred results of the previous tile leave, the same row groups of the next tile
    //      are requested into the registers they vacate
    if constexpr (SYNCP == 2 || SYNCP == 5) gang_meet();
    if constexpr (SYNCP == 13) p64v_barrier();
    if constexpr (PFSP < 2) static_for<0, 4 * PF>([&](auto ic) { pf_store(ic); });
    if constexpr ((SYNCP == 12 || SYNCP == 13 || SYNCP >= 15) && PFSP < 2) p64v_barrier();
    // PFSP: the deferred loads one at a time between the butterflies of the middle phase (1) / of E1's read phase and the middle (2)
    if constexpr (PFSP == 0) static_for<0, 4 * PF>([&](auto ic) { pf_load(ic); });
    if constexpr (PFSP == 0) static_for<0, 4 * LATE>([&](auto ic) { lat_load(ic); });
    // (in the quiet part the touch is a plain dword load into a register that is looked at once, behind E2: an LDS-DMA here would put a
    //  vmcnt(0) in front of the exchange's LDS reads)
    [[maybe_unused]] uint32_t tch[8];
    [[maybe_unused]] auto l2_touch_reg = [&](auto gc) {
      constexpr int g = decltype(gc)::value;
      const int w = __builtin_amdgcn_readfirstlane(tid0 >> 6);
      const uint32_t off = (uint32_t)((long long)(64 * g + lane + 512 * (w & 1) + 1024 * (w >> 1)) * v_sn * ESI);
      tch[g] = __builtin_amdgcn_raw_buffer_load_b32(rs_next, off, 0, 0);
    };
    if constexpr (PFL2 == 2) static_for<SPLIT, GP>([&](auto gc) { l2_touch_reg(gc); });
    if constexpr (PFL2 == 3) static_for<0, GP>([&](auto gc) { l2_touch_reg(gc); });
    __builtin_amdgcn_sched_barrier(0);
    // ---- E1: position k1 -> image row k1, column (p, u); thread (p, s = u) reads row u, slot n2.  No barrier behind the last read:
    //      the image stays busy until the barrier in front of the middle phase's last stage.
    [[maybe_unused]] auto pf_load_slot = [&](auto sc) {       // PFSP 5 / 6: the deferred loads over 14 slots — E1's three gaps, the middle's eight groups, E2's three gaps
      constexpr int sl = decltype(sc)::value, L = 4 * PF;
      static_for<sl * L / 14, (sl + 1) * L / 14>([&](auto ic) { pf_load(ic); });
    };
    if constexpr (PFSP == 5 || PFSP == 6) p64v_exchange_rest<false>(z, img, p, u, [&](auto kc) { pf_load_slot(kc); });
    else if constexpr (PFSP == 7) p64v_exchange_rest<false>(z, img, p, u, [&](auto kc) {           // all deferred loads in E1's three gaps
      constexpr int k = decltype(kc)::value, L = 4 * PF;
      static_for<k * L / 3, (k + 1) * L / 3>([&](auto ic) { pf_load(ic); });
    });
    else if constexpr (PFSP == 8) p64v_exchange_rest<false>(z, img, p, u, [&](auto kc) {           // ... in the six gaps of E1 and E2
      constexpr int k = decltype(kc)::value, L = 4 * PF;
      static_for<k * L / 6, (k + 1) * L / 6>([&](auto ic) { pf_load(ic); });
    });
    else if constexpr (PFSP == 10) p64v_exchange_rest<false>(z, img, p, u, [&](auto kc) {          // all deferred STORES in E1's first gap; loads: a quarter in each of the other two, half over the middle
      constexpr int k = decltype(kc)::value, L = 4 * PF, H = L / 2;
      if constexpr (k == 0) static_for<0, L>([&](auto ic) { pf_store(ic); });
      else static_for<(k - 1) * H / 2, k * H / 2>([&](auto ic) { pf_load(ic); });
    });
    else if constexpr (PFSP == 11) p64v_exchange_rest<false>(z, img, p, u, [&](auto kc) {          // a third of the stores, then a sixth of the loads, in each gap; half of the loads over the middle
      constexpr int k = decltype(kc)::value, L = 4 * PF, H = L / 2;
      static_for<k * L / 3, (k + 1) * L / 3>([&](auto ic) { pf_store(ic); });
      if constexpr (k > 0) static_for<(k - 1) * H / 2, k * H / 2>([&](auto ic) { pf_load(ic); });
    });
    else if constexpr (PFSP == 9) p64v_exchange_rest<false>(z, img, p, u, [&](auto kc) {           // deferred STORES in E1's first gap, loads in the other five
      constexpr int k = decltype(kc)::value, L = 4 * PF;
      if constexpr (k == 0) static_for<0, L>([&](auto ic) { pf_store(ic); });
      else static_for<(k - 1) * L / 5, k * L / 5>([&](auto ic) { pf_load(ic); });
    });
    else
    if constexpr (PFSP == 4) p64v_exchange_rest<false>(z, img, p, u, [&](auto kc) {
      constexpr int k = decltype(kc)::value, H = (4 * PF) / 2;
      static_for<k * H / 3, (k + 1) * H / 3>([&](auto ic) { pf_load(ic); });
    });
    else p64v_exchange_rest<false>(z, img, p, u);

    [[maybe_unused]] unsigned got4 = 0;
    if constexpr (MAPX == 4 || MAPX == 5) {        // follower: ask for the pair's ticket of the tile after next (published by the leader at ITS F1 barrier)
      if (__builtin_amdgcn_readfirstlane(tid0 >> 6) == 0 && member4 != 0) asm volatile("s_load_dword %0, %1, 0x0 glc" : "=s"(got4) : "s"(slot4) : "memory");
    }
    // ---- middle: F2 -> gate -> I1 (kernel_regtile.h; bin of register (ka, kb): k = k1 + 64 k2, k2 = ka + 8 kb) -------------
    {
      const int k1 = u;
      wskew();
      p64v_stageA1<false>(z);
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
        if constexpr (PFSP == 3) { static_for<ka * (4 * PF) / 8, (ka + 1) * (4 * PF) / 8>([&](auto ic) { pf_store(ic); }); }
        if constexpr (PFSP >= 1 && PFSP < 4) { static_for<ka * (4 * PF) / 8, (ka + 1) * (4 * PF) / 8>([&](auto ic) { pf_load(ic); }); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (PFSP == 5 || PFSP == 6) { pf_load_slot(std::integral_constant<int, 3 + ka>{}); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (PFSP == 4 || PFSP == 10 || PFSP == 11) { constexpr int H = (4 * PF) / 2, R = 4 * PF - H; static_for<H + ka * R / 8, H + (ka + 1) * R / 8>([&](auto ic) { pf_load(ic); }); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (PFSP >= 1 && LATE > 0) { static_for<ka * (4 * LATE) / 8, (ka + 1) * (4 * LATE) / 8>([&](auto ic) { lat_load(ic); }); __builtin_amdgcn_sched_barrier(0); }
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
        vpin8<8 * ka, 1>(z);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ka + 1 < 8)
          static_for<0, 8>([&](auto kbc) {
            constexpr int k2n = ka + 1 + 8 * decltype(kbc)::value;
            gcur[decltype(kbc)::value] = fetch_gate(k2n, k2n >= 32);
            if constexpr (WITH_MEM) mcur[decltype(kbc)::value] = fetch_mem(k2n, k2n >= 32);
          });
        fftB_stage1_group<8, 8, true, ka>(z);
        vpin8<8 * ka, 1>(z);
        __builtin_amdgcn_sched_barrier(0);         // keep the gate prefetch one group deep (register budget)
      });
      // type B stage 2: radix-8 over ka (positions 8 ka + n_lo) -> natural order, position n2 = n_lo + 8 n_hi.  Every wave has long
      // finished E1's reads; behind this barrier the image is written again, column by column like in F1.
      if constexpr (MAPX == 4 || MAPX == 5) {      // follower: the slot must carry this sequence number; a leader that is behind is waited for (bounded)
        if (__builtin_amdgcn_readfirstlane(tid0 >> 6) == 0 && member4 != 0) {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(got4) :: "memory");
          for (int i = 0; i < (1 << 16) && (got4 >> 16) != (unsigned)(it + 3); ++i) {
            __builtin_amdgcn_s_sleep(8);
            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(got4) : "s"(slot4) : "memory");
          }
          const unsigned tk = (got4 >> 16) == (unsigned)(it + 3) ? (got4 & 0xffffu) : 0xffffu;   // (time-out: this workgroup stops)
          if (lane == 0) tick_lds[0] = tk;
        }
      }
      p64v_barrier();
      static_for<0, 8>([&](auto nc) {
        constexpr int nlo = decltype(nc)::value;
        fftB_stage2_group<8, 8, true, nlo>(z);
        vpin8<nlo, 8>(z);
        __builtin_amdgcn_sched_barrier(0);
        p64v_write_col<nlo, false>(z, img, p, u);
        __builtin_amdgcn_sched_barrier(0);
      });
    }

    // ---- E2: position n2 -> image row n2, column (p, k1 = u); thread (p, u) reads row u, slot k1.  The barrier behind the last read
    //      frees the image for the LDS-DMA below.
    if constexpr (PFSP == 5 || PFSP == 6) p64v_exchange_rest<true>(z, img, p, u, [&](auto kc) { pf_load_slot(std::integral_constant<int, 11 + decltype(kc)::value>{}); });
    else if constexpr (PFSP == 8) p64v_exchange_rest<true>(z, img, p, u, [&](auto kc) {
      constexpr int k = 3 + decltype(kc)::value, L = 4 * PF;
      static_for<k * L / 6, (k + 1) * L / 6>([&](auto ic) { pf_load(ic); });
    });
    else if constexpr (PFSP == 9) p64v_exchange_rest<true>(z, img, p, u, [&](auto kc) {
      constexpr int k = 3 + decltype(kc)::value, L = 4 * PF;
      static_for<(k - 1) * L / 5, k * L / 5>([&](auto ic) { pf_load(ic); });
    });
    else p64v_exchange_rest<true>(z, img, p, u);
    mark(6);                                       // deferred stores / loads issue, E1, middle, E2
    if constexpr (SYNCP == 1 || SYNCP == 6) gang_meet();
    if constexpr (MEET == 2) gang_arrive();

    // ---- the image is idle until the next F1: let the first row groups of the next tile land in it, and fetch its gate -----
    const uint32_t voff = (uint32_t)(((long long)(u + 512 * h) * v_sn + 4 * pp) * ESI);
    // (the twiddles of I2 are read BEFORE the LDS-DMA is issued: hipcc orders every LDS read behind a pending LDS-DMA with
    //  s_waitcnt vmcnt(0) — the whole HBM round trip of the requests below, at the start of every burst)
    float2 wa[8], wb[8];
    load_twiddles(wa, wb, u);
    [[maybe_unused]] int fut_t = 0;
    if constexpr (MAPX >= 3) fut_t = __builtin_amdgcn_readfirstlane((int)tick_lds[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // DSPREAD: 0 = all 4 * SPLIT LDS-DMA requests in one burst; 1 = spread over the conj-twiddle multiplications (one share per 8 positions);
    // 2 = spread over the twiddle multiplications and the eight butterflies of I2's first stage
    [[maybe_unused]] const uint32_t dvo = dma_voff(voff, v_sn);
    auto dma_one = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(3);
      if constexpr (IN_BF16) {                       // two requests per group (dma_group)
        constexpr int g = q / 2, mh = q % 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_next, (__attribute__((address_space(3))) void*)(slot + (2 * g + mh) * 1024), 16,
                                                 dvo + (uint32_t)((64 * g + 2048 * mh) * v_sn * ESI), 0, 0, AUXD);
      } else {
        constexpr int g = q / 4, m = q % 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_next, (__attribute__((address_space(3))) void*)(slot + (4 * g + m) * 1024), 16,
                                                 voff + (uint32_t)((64 * g + 1024 * m) * v_sn * 4), 0, 0, AUXD);
      }
      if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(0);
    };
    wskew();
    constexpr int NDMA = (IN_BF16 ? 2 : 4) * SPLIT, NSLOT = DSPREAD == 3 ? 24 : DSPREAD == 2 ? 16 : 8;   // (3: ... and over the eight groups of I2's last stage)
    if constexpr (PFL2 == 1) static_for<SPLIT, GP>([&](auto gc) { l2_touch(gc); });
    if constexpr (PFL2 == 2) static_for<SPLIT, GP>([&](auto gc) { const uint32_t t = tch[decltype(gc)::value]; asm volatile("" :: "v"(t)); });
    if constexpr (PFL2 == 3) static_for<0, GP>([&](auto gc) { const uint32_t t = tch[decltype(gc)::value]; asm volatile("" :: "v"(t)); });
    // STAG: the two waves of a SIMD (w, w + 4) take turns — a wave that issues its 16 LDS-DMA requests sits in the issue stage until the
    // texture path has taken them (~40 clocks each with the memory system busy: batch 15's phase times), and with all eight waves doing that
    // at once nobody computes.  Waves 0-3 issue theirs here, waves 4-7 behind I2's first stage; each set computes while the other one waits.
    // (The prefetched rows are looked at first: hipcc counts only UNCONDITIONAL requests as younger, so with the burst inside a branch its
    //  wait for those registers at the trade in the store burst would become a wait for the LDS-DMA.  They were requested a phase and a half ago.)
    [[maybe_unused]] const bool set_y = __builtin_amdgcn_readfirstlane(tid0 >> 6) >= 4;
    if constexpr (STAG != 0 && PF > 0) static_for<0, 4 * PF>([&](auto ic) { vpin4(dfr[decltype(ic)::value]); });
    if constexpr (STAG != 0) { if (!set_y) static_for<0, SPLIT>([&](auto gc) { dma_group(rs_next, dma_voff(voff, v_sn), v_sn, gc); }); }
    else
    if constexpr (DSPREAD == 0 && SYNCP != 10) static_for<0, SPLIT>([&](auto gc) { dma_group(rs_next, dma_voff(voff, v_sn), v_sn, gc); });
    asm volatile("" ::: "memory");                 // the vmcnt() at the top of the loop counts on these being older than every store below

    // ---- conj twiddle, I2, stores (spectre.py:553) interleaved with the loads that refill the released registers -----------
    static_for<1, 64>([&](auto jc) {
      constexpr int j = decltype(jc)::value, ja = j % 8, jb = j / 8;     // position j carries k1 = j
      if constexpr (ja > 0) z[j] = cmulc(z[j], wa[ja]);
      if constexpr (jb > 0) z[j] = cmulc(z[j], wb[jb]);
      if constexpr (ja == 7) {
        vpin8<8 * jb, 1>(z); __builtin_amdgcn_sched_barrier(0);   // one wb at a time
        if constexpr (DSPREAD != 0) {
          static_for<jb * NDMA / NSLOT, (jb + 1) * NDMA / NSLOT>([&](auto qc) { dma_one(qc); });
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    });
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DSPREAD == 2 || DSPREAD == 3) p64v_stageA1_cb<true>(z, [&](auto q0c) {
      constexpr int sl = 8 + decltype(q0c)::value;
      static_for<sl * NDMA / NSLOT, (sl + 1) * NDMA / NSLOT>([&](auto qc) { dma_one(qc); });
    });
    else if constexpr (STAG == 2) {                 // waves 4-7 issue half way through the first stage
      p64v_stageA1_cb<true>(z, [&](auto q0c) {
        if constexpr (decltype(q0c)::value == 3) { if (set_y) static_for<0, SPLIT>([&](auto gc) { dma_group(rs_next, dma_voff(voff, v_sn), v_sn, gc); }); }
      });
    }
    else p64v_stageA1<true>(z);
    asm volatile("" ::: "memory");
    if constexpr (STAG == 1) { if (set_y) static_for<0, SPLIT>([&](auto gc) { dma_group(rs_next, dma_voff(voff, v_sn), v_sn, gc); }); }
    asm volatile("" ::: "memory");
    {
      const uint32_t ooff = (uint32_t)(((long long)(u + 512 * h) * out_sn + 4 * pp) * ESO);
      if constexpr (SYNCP == 17) {
        // SPLIT BURST: the groups that are reloaded behind their stores go first — last stage, barrier, their stores, barrier, their reloads —
        // and those loads travel while the other groups' last stage and store burst run
        static_for<SPLIT, GP>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          fftA_stage2_group<8, 8, true, g>(z);
          vpin8<8 * g, 1>(z);
          swap_group(gc);
          __builtin_amdgcn_sched_barrier(0);
        });
        p64v_barrier();
        static_for<SPLIT, GP>([&](auto gc) {
          constexpr int g = decltype(gc)::value;
          static_for<0, 4>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            store16(rs_out, ooff + (uint32_t)((64 * g + 1024 * m) * out_sn * ESO), make_float4(z[8 * g + 2 * m].x, z[8 * g + 2 * m].y, z[8 * g + 2 * m + 1].x, z[8 * g + 2 * m + 1].y));
          });
          __builtin_amdgcn_sched_barrier(0);
        });
        p64v_barrier();
        static_for<SPLIT, GP>([&](auto gc) { load_group(rs_next, voff, v_sn, gc); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 8 - (GP - SPLIT)>([&](auto ic) {
          constexpr int g = (decltype(ic)::value + GP) % 8;
          fftA_stage2_group<8, 8, true, g>(z);
          vpin8<8 * g, 1>(z);
          swap_group(std::integral_constant<int, g>{});
          __builtin_amdgcn_sched_barrier(0);
        });
        p64v_barrier();
        static_for<0, 8 - (GP - SPLIT)>([&](auto ic) {
          constexpr int g = (decltype(ic)::value + GP) % 8;
          static_for<0, 4>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const float4 res = make_float4(z[8 * g + 2 * m].x, z[8 * g + 2 * m].y, z[8 * g + 2 * m + 1].x, z[8 * g + 2 * m + 1].y);
            if constexpr (g >= GP) {
              if (more) {
                const float4 nx = dfr[4 * (g - GP) + m];
                dfr[4 * (g - GP) + m] = res;
                z[8 * g + 2 * m] = make_float2(nx.x, nx.y);
                z[8 * g + 2 * m + 1] = make_float2(nx.z, nx.w);
              } else {
                store16(rs_out, ooff + (uint32_t)((64 * g + 1024 * m) * out_sn * ESO), res);
              }
            } else {
              store16(rs_out, ooff + (uint32_t)((64 * g + 1024 * m) * out_sn * ESO), res);
            }
          });
          __builtin_amdgcn_sched_barrier(0);
        });
      } else if constexpr (SYNCP >= 3 && SYNCP != 8) {
        // store BURST: every butterfly of I2's last stage first, then (SYNCP 3: the gang meets) all stores back to back, then the reloads —
        // the two workgroups of a pair put both halves of every line into the L2 within a couple of microseconds (tools/store_lab.hip: halves
        // that arrive within ~1 us of each other cost what a whole line costs)
        static_for<0, 8>([&](auto ic) {
          constexpr int g = (decltype(ic)::value + SPLIT) % 8;
          fftA_stage2_group<8, 8, true, g>(z);
          vpin8<8 * g, 1>(z);
          swap_group(std::integral_constant<int, g>{});
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (PARK != 0 && decltype(ic)::value == 0) {        // g = SPLIT: results -> LDS, its registers take the next tile's rows NOW (loads, in front of the burst)
            static_for<0, 4>([&](auto mc) {
              constexpr int m = decltype(mc)::value;
              pv_f32x4 res; res.x = z[8 * g + 2 * m].x; res.y = z[8 * g + 2 * m].y; res.z = z[8 * g + 2 * m + 1].x; res.w = z[8 * g + 2 * m + 1].y;
              const uint32_t pa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + 8 * SPLIT * GROUP_SLOT + (__builtin_amdgcn_readfirstlane(tid0 >> 6) * 4 + m) * 1024 + lane * 16);
              asm volatile("ds_write_b128 %0, %1" :: "v"(pa), "v"(res) : "memory");
            });
            load_group(rs_next, voff, v_sn, std::integral_constant<int, SPLIT>{});
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (DSPREAD == 3) {
            constexpr int sl = 16 + decltype(ic)::value;
            static_for<sl * NDMA / NSLOT, (sl + 1) * NDMA / NSLOT>([&](auto qc) { dma_one(qc); });
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        if constexpr (SYNCP == 3 || SYNCP == 5 || SYNCP == 7) gang_meet();
        mark(7);                                   // DMA issue, twiddles, I2 (wave 0's own)
        if constexpr (MEET == 1) gang_meet(); else if constexpr (MEET == 2) gang_wait(); else
        if constexpr (SYNCP >= 9) p64v_barrier();
        mark(8);                                   // barrier in front of the burst (the slowest wave's I2)
        if constexpr (PRIO == 4) __builtin_amdgcn_s_setprio(3);
        if constexpr (SYNCP == 10) static_for<0, NDMA>([&](auto qc) { dma_one(qc); });
        static_for<0, 8>([&](auto ic) {
          constexpr int g = (decltype(ic)::value + SPLIT) % 8;
          static_for<0, 4>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const float4 res = make_float4(z[8 * g + 2 * m].x, z[8 * g + 2 * m].y, z[8 * g + 2 * m + 1].x, z[8 * g + 2 * m + 1].y);
            if constexpr (g >= GP) {
              if (more) {
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
            } else if constexpr (PARK != 0 && g == SPLIT) {
              // parked (and its registers already hold the next tile's rows): nothing to store here
            } else {
              store16(rs_out, ooff + (uint32_t)((64 * g + 1024 * m) * out_sn * ESO), res);
              if constexpr (LATE > 0 && g >= GP - LATE) {           // the prefetched rows of the next tile move in
                const float4 nx = lat[4 * (g - (GP - LATE)) + m];
                z[8 * g + 2 * m] = make_float2(nx.x, nx.y);
                z[8 * g + 2 * m + 1] = make_float2(nx.z, nx.w);
              }
            }
          });
          __builtin_amdgcn_sched_barrier(0);
        });
        mark(9);                                   // store issue (+ the wait for the prefetched rows that trade places)
        if constexpr (SYNCP >= 11) p64v_barrier();
        mark(10);                                  // barrier behind the burst
        wskew();
        if constexpr (SYNCP == 15) __builtin_amdgcn_s_sleep(4);
        if constexpr (SYNCP == 16) __builtin_amdgcn_s_sleep(16);
        if constexpr (EARLY1 == 1) { if (more) { static_for<0, PF>([&](auto ic) {
            constexpr int g = GP + decltype(ic)::value;
            swap_group(std::integral_constant<int, g>{});
            bfly_plain<8, false, 8 * g, 1, 64>(z);
            vpin8<8 * g, 1>(z);
            __builtin_amdgcn_sched_barrier(0);
          }); } }
        if constexpr (RLF == 0) static_for<SPLIT + (PARK != 0 ? 1 : 0), GP - LATE>([&](auto gc) { load_group(rs_next, voff, v_sn, gc); });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (EARLY1 == 2) { if (more) { static_for<0, PF>([&](auto ic) {
            constexpr int g = GP + decltype(ic)::value;
            swap_group(std::integral_constant<int, g>{});
            bfly_plain<8, false, 8 * g, 1, 64>(z);
            vpin8<8 * g, 1>(z);
            __builtin_amdgcn_sched_barrier(0);
          }); } }
      } else {
      if constexpr (SYNCP == 8) p64v_barrier();
      static_for<0, 8>([&](auto ic) {
        constexpr int g = (decltype(ic)::value + SPLIT) % 8;             // register-loaded groups first: their reloads start earliest
        fftA_stage2_group<8, 8, true, g>(z);                              // rows g + 8e at positions 8g + e
        vpin8<8 * g, 1>(z);
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
    if constexpr (SYNCP == 14) p64v_barrier();
    if constexpr (RLF == 0) gate_fetch(gpn);     // committed to LDS at the end of the next tile's F1 (after the last tile: a harmless re-read of this tile's bins)
    if constexpr (MAPX >= 3) { cur_t = nxt_t; nxt_t = fut_t; }
    mark(11);                                      // reload issue + gate fetch issue
  }  // tile loop
  if constexpr (PARK != 0) {                       // the last tile's parked group
    coords();
    const uint32_t ooff_l = (uint32_t)(((long long)(u + 512 * h) * a.out_sn + 4 * pp) * ESO);
    static_for<0, 4>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      pv_f32x4 rv;
      const uint32_t pa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + 8 * SPLIT * GROUP_SLOT + (__builtin_amdgcn_readfirstlane(tid0 >> 6) * 4 + m) * 1024 + lane * 16);
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rv) : "v"(pa) : "memory");
      store16(rsrc_out(obp, a.out_sn, true), ooff_l + (uint32_t)((64 * SPLIT + 1024 * m) * a.out_sn * ESO), make_float4(rv.x, rv.y, rv.z, rv.w));
    });
  }
  if constexpr (TSTAMP) {                          // when did this workgroup start / finish (100 MHz)?  a.mem + 256 words: [2 wg], [2 wg + 1]
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid0 == 0) {
      unsigned long long* tp = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.mem)) + 128 + 2 * blockIdx.x;
      tp[0] = t_start; tp[1] = __builtin_amdgcn_s_memrealtime();
      unsigned long long* cp2 = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.mem)) + 4096 + 2 * blockIdx.x;   // shader clocks
      cp2[0] = c_start; cp2[1] = __builtin_readcyclecounter();
      if constexpr (TSTAMP == 2) {
        unsigned long long* pp2 = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.mem)) + 1024 + 12 * blockIdx.x;
        for (int k = 0; k < 12; ++k) pp2[k] = ph_acc[k];
      }
    }
  }
}

hipError_t launch_p64v_unused(const RegtileArgs& a, bool in_bf16, bool out_bf16, hipStream_t stream);

}  // namespace sfft
