// kernel_regtile_mixedp.h — persistent, software-pipelined variant of kernel_regtile_mixed.h (n_fft = RF * RS, e.g. 3000 = 60 x 50) for
// fp32 rows in and out, every 16-channel tile inside one gate group, any sequence length (rows >= N_in are the buffer instructions'
// out-of-range case, as in kernel_regtile64p.h).  Same mathematics, same tile, same exchanges; replaces /root/reference/spectre.py:506,
// :542-553.
//
// A mixed-radix tile needs 2 * max(RF, RS) data registers per thread (120 for 60 x 50) of the 256 a 400-thread workgroup may use, so —
// unlike at n_fft = 4096, where the tile fills the register file — P row blocks can be DEFERRED the way kernel_regtile64p.h defers three
// row groups: their results stay in 2 P registers through F1 of the next tile and are stored there (a quiet part of the tile), the same
// registers then request those row blocks of the tile after, which trade places with the next results at the end of I2.  The other
// RF - P row blocks are stored at the end of the tile and reloaded behind their stores.  One workgroup per CU walks through its tiles,
// pairs of workgroups on adjacent tiles (one 128-byte line per row).  Everything learned there about hipcc's waits applies: LDS-only
// barriers (a __syncthreads() would drain the deferred requests), every request unconditional (an empty buffer range when there is no
// next / previous tile), the gate bins requested raw and fixed up at commit time, the twiddle bases requested BEFORE the deferred block.
#pragma once
#include "kernel_regtile_mixed.h"
#include "kernel_tickets.h"

namespace sfft {

// Bin k = u + RF*k2 (u < RF a lane quantity, k2 a compile-time one) against N/2: 0 = at or below for every u, 1 = above for every u
// (the gate is conj(g[N - k]) there), 2 = depends on u (the one or two k2 that straddle N/2).
template <int RF, int N> constexpr int mixed_gate_side(int k2) { return 2 * (RF - 1 + RF * k2) <= N ? 0 : (2 * RF * k2 > N ? 1 : 2); }

__device__ __forceinline__ void rt_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One float of the exchange image, read by a single ds_read_b32 with an immediate offset.  hipcc pairs neighbouring reads into
// ds_read2_b32, whose two results must sit in consecutive VGPRs — but x and y of one value already do (64-bit global loads and stores), so
// every pair costs a v_mov (320 per tile at 60 x 50).  The wave's LDS queue has room for twice the instructions; the VALU does not have
// room for the moves.  The result is not tracked by the compiler's s_waitcnt insertion: the caller reads behind rt_lds_barrier() (which
// waits for lgkmcnt(0)) and then passes the values through mp_pin() so that no use is scheduled above that barrier.
template <int OFF_BYTES>
__device__ __forceinline__ float mp_lds_read(uint32_t addr) {
  static_assert(OFF_BYTES >= 0 && OFF_BYTES < 65536, "16-bit immediate");
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF_BYTES));
  return v;
}
__device__ __forceinline__ void mp_pin1(float& x) { asm volatile("" : "+v"(x)); }
template <int LO, int HI, bool IM, int NZ>
__device__ __forceinline__ void mp_pin(float2 (&z)[NZ]) {
  static_for<LO, HI>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (IM) mp_pin1(z[i].y); else mp_pin1(z[i].x);
  });
}

// (Round 3 tried the 4096 kernel's early exchange writes here — E1's real plane written position by position behind the twiddle
//  multiplication, the image-freeing barrier moved in front of it or to the top of the tile: 1.87 / 1.89 ms against 1.59, and 1.24 against
//  1.15 ms without memory traffic; profiles/r03_mixedp_ab.log.  The scattered ds_write_b32 between the multiplications cost more than the
//  write phase they replace, and the earlier barrier exposes the slowest wave's loads.  Not kept.)
// STAGED row blocks (round 4): S of the RF row blocks of the NEXT tile — the first S of the order F1 uses them, S even — are requested
// BEFORE this tile's stores, as LDS-DMA (buffer_load_dwordx4 ... lds, no register needed) into the exchange image, which is idle between
// E2 and the next E1, and into the LDS behind the gate that a mixed-radix tile leaves free (3000: 96 KiB image + 12 KiB gate of 160).
// A wave's loads and stores retire through one in-order counter, so a row block reloaded behind its store is only there when that store
// has been acknowledged (same-box ablation at 3000 with every block deferred or reloaded: loads alone + 0.19 ms, stores alone + 0.03,
// both + 0.59); staged blocks are older than every store.  One request = 1 KiB = the 8 row classes of the wave x 2 row blocks x 64 bytes:
// lane l fetches quarter l & 3 of the row of class (l >> 2) & 7 of block pair member l >> 5, and thread (p, u) reads its 8 bytes back
// out of its OWN wave's slot (no barrier), conflict-free (the wave's 64 reads cover 512 contiguous bytes).  Waves without a row class
// (u >= RS for all their threads) issue the same requests into a 1-KiB dump, so that every wave counts the same requests.
template <int RF, int RS, int S> constexpr int mixedp_slot_bytes() { return (S / 2) * 1024; }
template <int RF, int RS, int S> constexpr int mixedp_row_waves() { return (RS + 7) / 8; }                       // waves that own row classes
template <int RF, int RS, int S> constexpr int mixedp_image_waves() {                                             // ... whose slots lie in the image
  return S == 0 ? 0 : (mixed_image_bytes<RF, RS>() / mixedp_slot_bytes<RF, RS, S>() < mixedp_row_waves<RF, RS, S>() ? mixed_image_bytes<RF, RS>() / mixedp_slot_bytes<RF, RS, S>() : mixedp_row_waves<RF, RS, S>());
}
template <int RF, int RS, int S> constexpr int mixedp_extra_off() { return (mixed_lds_total<RF, RS>() + 15) & ~15; }   // behind the gate
template <int RF, int RS, int S> constexpr int mixedp_dump_off() { return mixedp_extra_off<RF, RS, S>() + (mixedp_row_waves<RF, RS, S>() - mixedp_image_waves<RF, RS, S>()) * mixedp_slot_bytes<RF, RS, S>(); }
// ... and with staged blocks the twiddle bases of every row class live in LDS as well (written once per workgroup): as global loads
// they are 2 x (RA + RB - 2) of the tile's VMEM requests per thread (34 of ~150 at 60 x 50), they sit in the same in-order counter as the
// LDS-DMA and the stores, the ones in front of I2 are consumed as soon as they are requested (a memory latency per tile), and their
// 2 (RA + RB) registers stay live through the transform.  One row of 16-byte-aligned float2 per row class: W^(u j), j = 1 .. RA - 1, then
// W^(u RA j), j = 1 .. RB - 1.
template <int RF> constexpr int mixedp_tw_row() { return ((Split<RF>::RA + Split<RF>::RB - 2) + 1) & ~1; }       // float2 per row class
template <int RF, int RS, int S> constexpr int mixedp_tw_off() { return mixedp_dump_off<RF, RS, S>() + 1024; }
template <int RF, int RS, int S> constexpr int mixedp_lds_total() { return S == 0 ? mixed_lds_total<RF, RS>() : mixedp_tw_off<RF, RS, S>() + RS * mixedp_tw_row<RF>() * 8; }

// XP = experiment switches (tools/mixedp_stage_bench.hip): bit 0 = a workgroup barrier between I2 and the store burst, bit 1 = one behind
// the burst (in front of the reloads / the gate fetch), bits 2 / 3 = the deferred loads in the gaps of E1 / of E1 and E2
// Launched with WHOLE waves (mixedp_launch_threads): an LDS-DMA request uses all 64 lanes of a wave — lanes 32-63 fetch the second row
// block of a pair — and at 60 x 60 the 480 threads of the team would leave wave 7, which owns row classes 56-59, with half its lanes.
// The threads beyond the team (u >= RF and u >= RS) own neither rows nor bins and only take part in the barriers and the requests.
template <int RF, int RS> constexpr int mixedp_launch_threads() { return (mixed_threads<RF, RS>() + 63) & ~63; }
// TICKETS (round 5): the pairs take adjacent tile pairs from one chip-wide counter in address order instead of walking through their own
// regions (kernel_tickets.h; the long story is in kernel_regtile64p.h): two LDS words behind everything else carry the tile after next.
template <int RF, int RS, int S> constexpr int mixedp_tk_off() { return (mixedp_lds_total<RF, RS, S>() + 15) & ~15; }
template <int RF, int RS, int S, bool TICKETS> constexpr int mixedp_lds_bytes() { return TICKETS ? mixedp_tk_off<RF, RS, S>() + 16 : mixedp_lds_total<RF, RS, S>(); }
template <int RF, int RS, int P, bool FIRST = false, int S = 0, int XP = 0, bool TICKETS = false>
__global__ void __launch_bounds__((mixedp_launch_threads<RF, RS>()), 1) spectre_mix_regtile_mixedp(const RegtileArgs a) {
  static_assert(!TICKETS || ((XP & 8) != 0 && S > 0), "TICKETS: built on the form with the deferred loads in the gaps of E1 / E2");
  static_assert(mixedp_lds_bytes<RF, RS, S, TICKETS>() <= 160 * 1024, "LDS budget");
  constexpr int D0 = FIRST ? 0 : RF - P;           // the deferred row blocks are [D0, D0 + P) of the order F1 uses them (the last ones: measured
                                                   // 1 % better than the first ones, 1.586 vs 1.605 ms)
  static_assert(S % 2 == 0 && S >= 0 && (S == 0 || (!FIRST && S <= D0)), "staged row blocks: the first S of F1's order, in pairs; the deferred ones are the last P");
  static_assert(mixedp_lds_total<RF, RS, S>() <= 160 * 1024, "LDS budget");
  constexpr int N = RF * RS, NZ = mixed_team<RF, RS>(), NT = mixed_threads<RF, RS>();
  static_assert(N % 2 == 0 && P >= 0 && P <= RF, "even n_fft; P deferred row blocks");
  constexpr int RAF = Split<RF>::RA, RBF = Split<RF>::RB;
  constexpr int ROW1 = mixed_row(RS), ROW2 = mixed_row(RF);
  constexpr int GS = (N / 2 + 1 + NT - 1) / NT;     // gate bins per thread
  constexpr float inv_n = 1.0f / (float)N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);
  float2* glds = reinterpret_cast<float2*>(smem + mixed_image_bytes<RF, RS>());
  typedef __attribute__((address_space(3))) float lds_float;
  auto lane_base = [&](int off) { lds_float* b = (lds_float*)(img + off); asm volatile("" : "+v"(b)); return b; };

  // Only threadIdx.x stays live across the tile loop; the lane coordinates are re-derived from an opaque copy per tile (otherwise LICM
  // hoists every per-lane address out of the loop and the allocator spills them; kernel_regtile64p.h).
  const int tid0 = threadIdx.x;
  int tid, p, u;
  bool rows, bins;
  auto coords = [&]() {
    int t = tid0;
    asm volatile("" : "+v"(t));
    tid = t; p = t & (kPC - 1); u = t / kPC;
    rows = u < RS; bins = u < RF;
  };
  coords();

  const int wg_lin = xcd_contiguous(blockIdx.x, a.n_wg);
  const int pair_base = (wg_lin >> 1) * a.tpw * 2 + (wg_lin & 1);   // workgroups 2m, 2m+1 walk through adjacent tiles
  if constexpr (!TICKETS) { if (pair_base >= a.n_tiles) return; }
  [[maybe_unused]] GangTickets<2> tk;
  [[maybe_unused]] int cur_tile = -2, nxt_tile = -2;
  if constexpr (TICKETS) tk.init(a.tickets, wg_lin >> 1, wg_lin & 1, a.n_tiles, reinterpret_cast<volatile int*>(smem + mixedp_tk_off<RF, RS, S>()), tid0);

  float2 z[NZ];
  float2 dfr[P > 0 ? P : 1];                        // deferred results of the previous tile / prefetched row blocks of the next one
  float2 gst[GS];
  static_for<0, (P > 0 ? P : 1)>([&](auto ic) { dfr[decltype(ic)::value] = make_float2(0.f, 0.f); });
  char* obp = nullptr;                              // output tile of the deferred results

  auto tile_ptrs = [&](int tile, const char*& vb, char*& ob, const float2*& gp) {
    const int b = tile / a.tiles_per_row, ct = tile - b * a.tiles_per_row;
    vb = reinterpret_cast<const char*>(a.v) + ((size_t)b * a.v_sb + (size_t)ct * 16) * 4;
    ob = reinterpret_cast<char*>(a.out) + ((size_t)b * a.out_sb + (size_t)ct * 16) * 4;
    gp = a.gate + ((size_t)b * a.G + (ct * 16) / a.d_g) * a.F;
  };
  // Buffer resources: base = the tile's first row, num_records = the bytes of its rows below N_in (0 = nothing: no such tile).  The
  // range check covers the VGPR offset, so the row-block offset goes there too; threads that own no row class (RF > RS) get an offset
  // beyond every range, so every wave issues the same requests.
  auto rsrc = [&](const void* base, long long sn, int nrow) {   // nrow = a.rows_in / a.rows_out, or 0: no such tile
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)((long long)nrow * sn * 4), kRsrcFlags);
  };
  // row blocks in the order F1 uses them
  auto row_q = [](auto ic) { constexpr int i = decltype(ic)::value; return std::integral_constant<int, in_order<RF>(i)>{}; };
  auto load_row = [&](__amdgpu_buffer_rsrc_t rs, uint32_t voff, long long sn, auto qc) -> float2 {
    const rt_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + (uint32_t)((long long)decltype(qc)::value * RS * sn * 4), 0, 0);
    return make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
  };
  auto store_row = [&](__amdgpu_buffer_rsrc_t rs, uint32_t ooff, long long sn, auto qc, float2 v) {
    rt_u32x2 t;
    t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(t, rs, ooff + (uint32_t)((long long)decltype(qc)::value * RS * sn * 4), 0, 0);
  };
  // ---- staged row blocks: this wave's slot (SGPRs), the lane's request offset and read-back address
  [[maybe_unused]] const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  [[maybe_unused]] const bool wave_rows = wv < mixedp_row_waves<RF, RS, S>();
  [[maybe_unused]] const int slot_off = !wave_rows ? mixedp_dump_off<RF, RS, S>()
                                       : wv < mixedp_image_waves<RF, RS, S>() ? wv * mixedp_slot_bytes<RF, RS, S>()
                                       : mixedp_extra_off<RF, RS, S>() + (wv - mixedp_image_waves<RF, RS, S>()) * mixedp_slot_bytes<RF, RS, S>();
  [[maybe_unused]] const int slot_step = wave_rows ? 1024 : 0;
  // lane part of a request's offset, once per burst: (row class, quarter) of this lane, and which member of the block pair it fetches
  [[maybe_unused]] uint32_t dma_base = 0;
  [[maybe_unused]] bool dma_second = false;
  [[maybe_unused]] auto stage_coords = [&](long long sn) {
    int t = tid0;
    asm volatile("" : "+v"(t));
    const int l = t & 63, ud = 8 * (t >> 6) + ((l >> 2) & 7);
    const uint32_t inside = (uint32_t)(((long long)ud * sn + 4 * (l & 3)) * 4);
    dma_base = ud < RS ? inside : 0x80000000u;        // (a select, not a branch: beyond every range, like the row offsets of threads without rows)
    dma_second = (l >> 5) != 0;
  };
  [[maybe_unused]] auto stage_pair = [&](__amdgpu_buffer_rsrc_t rs, long long sn, auto jc) {    // row blocks row_q(2j), row_q(2j + 1) of the tile behind rs
    constexpr int j = decltype(jc)::value;
    constexpr int qa = in_order<RF>(2 * j), qb = in_order<RF>(2 * j + 1);
    const uint32_t ca = (uint32_t)((long long)qa * RS * sn * 4), cb = (uint32_t)((long long)qb * RS * sn * 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + slot_off + j * slot_step), 16,
                                             dma_base + (dma_second ? cb : ca), 0, 0, 0);
  };
  // read-back: one lane address per tile, the row block is an immediate offset (waves without rows read the image's first bytes: unused)
  [[maybe_unused]] auto staged_base = [&]() -> const char* {
    int t = tid0;
    asm volatile("" : "+v"(t));
    const int l = t & 63;
    return smem + (wave_rows ? slot_off : 0) + ((l >> 3) * 4 + ((l & 7) >> 1)) * 16 + (l & 1) * 8;
  };
  [[maybe_unused]] auto staged_read = [&](const char* rb, auto ic) -> float2 {                  // this thread's 8 bytes of row block row_q(i)
    constexpr int i = decltype(ic)::value;
    return *reinterpret_cast<const float2*>(rb + (i / 2) * 1024 + (i & 1) * 512);
  };

  auto gate_fetch = [&](const float2* gp) {          // raw; all arithmetic happens in gate_commit (nothing computed before the back edge)
    static_for<0, GS>([&](auto ic) {
      const int k = tid + NT * decltype(ic)::value;
      gst[decltype(ic)::value] = gp[k <= N / 2 ? k : N / 2];
    });
  };
  auto gate_commit = [&]() {
    static_for<0, GS>([&](auto ic) {
      const int k = tid + NT * decltype(ic)::value;
      float2 g = gst[decltype(ic)::value];
      asm volatile("" : "+v"(g.x), "+v"(g.y));       // consumed here by every wave (kernel_regtile64p.h)
      if (k == 0 || k == N / 2) g.y = 0.f;           // irfft ignores Im(DC), Im(Nyquist) (spectre.py:551)
      if (a.conj_gate) g.y = -g.y;
      if (k <= N / 2) glds[k] = make_float2(g.x * inv_n, g.y * inv_n);
    });
  };
  if constexpr (S > 0) {                             // the twiddle bases of every row class -> LDS, once
    constexpr int TR = mixedp_tw_row<RF>();
    float2* twl = reinterpret_cast<float2*>(smem + mixedp_tw_off<RF, RS, S>());
    for (int i = threadIdx.x; i < RS * TR; i += NT) {
      const int uu = i / TR, e = i - TR * uu;
      twl[i] = e < RAF - 1 ? a.tw[uu * (e + 1)] : e < RAF + RBF - 2 ? a.tw[uu * RAF * (e - (RAF - 1) + 1)] : make_float2(0.f, 0.f);
    }
    __syncthreads();
  }
  auto load_twiddle_bases = [&](float2 (&wa)[RAF], float2 (&wb)[RBF]) {   // W_N^(u ka), W_N^(u RAF kb)
    const int uu = rows ? u : 0;
    if constexpr (S > 0) {
      constexpr int TR = mixedp_tw_row<RF>();
      const float4* t = reinterpret_cast<const float4*>(smem + mixedp_tw_off<RF, RS, S>()) + uu * (TR / 2);
      float2 flat[TR];
      static_for<0, TR / 2>([&](auto ic) { constexpr int i = decltype(ic)::value; const float4 q = t[i]; flat[2 * i] = make_float2(q.x, q.y); flat[2 * i + 1] = make_float2(q.z, q.w); });
      static_for<1, RAF>([&](auto jc) { constexpr int j = decltype(jc)::value; wa[j] = flat[j - 1]; });
      static_for<1, RBF>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = flat[RAF - 1 + j - 1]; });
      return;
    }
    static_for<1, RAF>([&](auto jc) { constexpr int j = decltype(jc)::value; wa[j] = a.tw[uu * j]; });
    static_for<1, RBF>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = a.tw[uu * RAF * j]; });
  };

  // TICKETS: round 0 follows the pair's tickets; every later round processes ONE tile nobody has claimed (the sweep).  Static map: one round.
  for (int round = 0;; ++round) {
  if constexpr (TICKETS) {
    if (round == 0) tk.first_two(tid0, cur_tile, nxt_tile);
    else {
      asm volatile("s_waitcnt vmcnt(0) ; lint: drain" ::: "memory");
      if (!tk.sweep(tid0, mixedp_launch_threads<RF, RS>(), wg_lin, a.n_wg, cur_tile)) break;
      nxt_tile = -2;
    }
    if (cur_tile == -2) continue;
  }
  [[maybe_unused]] bool prev_live = false;          // TICKETS: dfr holds the deferred results of a real previous tile
  // ---- prologue: the whole first tile is loaded the way the non-deferred row blocks of every later tile are --------------------
  {
    const char* vb; char* ob; const float2* gp;
    tile_ptrs(TICKETS ? (cur_tile >= 0 ? cur_tile : 0) : pair_base, vb, ob, gp);
    const __amdgpu_buffer_rsrc_t rs = rsrc(vb, a.v_sn, (!TICKETS || cur_tile >= 0) ? a.rows_in : 0);
    const uint32_t voff = rows ? (uint32_t)(((long long)u * a.v_sn + 2 * p) * 4) : 0x80000000u;
    if constexpr (S > 0) stage_coords(a.v_sn);
    static_for<0, S / 2>([&](auto jc) { stage_pair(rs, a.v_sn, jc); });
    asm volatile("" ::: "memory");
    static_for<S, RF>([&](auto ic) { constexpr int q = decltype(row_q(ic))::value; z[q] = load_row(rs, voff, a.v_sn, std::integral_constant<int, q>{}); });
    gate_fetch(gp);
  }

  for (int it = 0; TICKETS || it < a.tpw; ++it) {
    const int tile = TICKETS ? (cur_tile >= 0 ? cur_tile : 0) : pair_base + 2 * it;
    if (TICKETS ? cur_tile == -2 : tile >= a.n_tiles) break;                    // workgroup-uniform
    const bool more = TICKETS ? nxt_tile >= 0 : (it + 1 < a.tpw) && (tile + 2 < a.n_tiles);
    [[maybe_unused]] const bool cur_live = TICKETS ? cur_tile >= 0 : true;       // false: a phantom tile (empty buffer ranges)
    coords();
    [[maybe_unused]] const bool tk_lead = TICKETS && round == 0 && tk.leading(), tk_foll = TICKETS && round == 0 && tk.following();
    if constexpr (TICKETS) { tk.begin_tile(); if (tk_lead) tk.draw(); }
    long long v_sn = a.v_sn, out_sn = a.out_sn;
    asm volatile("" : "+s"(v_sn), "+s"(out_sn));
    const char* vb; char* ob; const float2* gp;
    tile_ptrs(tile, vb, ob, gp);
    const char* vbn = vb; char* obn = ob; const float2* gpn = gp;
    if (more) tile_ptrs(TICKETS ? nxt_tile : tile + 2, vbn, obn, gpn);
    // (obp is carried across the back edge in VGPRs, so hipcc wraps each of the P deferred stores in a waterfall loop — v_readfirstlane,
    //  v_cmp, s_and_saveexec.  Recomputing the pointer from tile - 2 removes the loops and is SLOWER, 1.6 % at 60 x 50 and 7 % at 64 x 40
    //  (profiles/r03_mixedp_opt_matrix.log, OPT=8 against OPT=0): the loops pace the stores of the quiet part.  Left as hipcc emits it.)
    const __amdgpu_buffer_rsrc_t rs_next = rsrc(vbn, v_sn, more ? a.rows_in : 0), rs_out = rsrc(ob, out_sn, cur_live ? a.rows_out : 0),
                                 rs_prev = rsrc(obp, out_sn, (TICKETS ? prev_live : it > 0) ? a.rows_out : 0);
    const uint32_t voff = rows ? (uint32_t)(((long long)u * v_sn + 2 * p) * 4) : 0x80000000u;
    const uint32_t ooff = rows ? (uint32_t)(((long long)u * out_sn + 2 * p) * 4) : 0x80000000u;

    // ---- rows u + RS*q, q < RF: F1 over q, W_N^(u k1) ---------------------------------------------------------------------
    float2 wa[RAF], wb[RBF];
    if constexpr (S == 0) load_twiddle_bases(wa, wb);
    // the staged row blocks: requested before the previous tile's stores.  hipcc puts an s_waitcnt vmcnt(N) in front of these reads itself
    // (N = the requests that are certainly younger than the last LDS-DMA: that tile's stores, its reloads, the gate fetch); the same wait is
    // ALSO written out by hand, tagged for fft_amd/isa_lint.py, so that a compiler whose LDS-DMA alias tracking changes cannot turn the
    // read-back into a silent race: completion is in order, so with N requests issued behind the last LDS-DMA on every path
    // (the fewest: a tile with a successor — RF - P stores, RF - S - P reloads, GS gate loads; first tile: RF - S loads, GS gate loads),
    // vmcnt(N) means the staged row blocks have landed.
    if constexpr (S > 0) {
      if (it == 0) asm volatile("s_waitcnt vmcnt(%0) ; lint: first" :: "n"(RF - S + GS) : "memory");
      else if constexpr ((RF - P) + (RF - S - P) + GS <= 63) asm volatile("s_waitcnt vmcnt(%0) ; lint: steady" :: "n"((RF - P) + (RF - S - P) + GS) : "memory");
      // (more than 63 requests behind the LDS-DMA — 64 x 40, 60 x 40 —: the counter has 6 bits and a wave cannot have more than 63 requests
      //  outstanding, so by the time the 64th has been issued the LDS-DMA in front of them has completed; nothing to wait for)
      const char* rb = staged_base();
      static_for<0, S>([&](auto ic) { z[decltype(row_q(ic))::value] = staged_read(rb, ic); });
    }
    if (rows) {
      fft_ct<RF, false, IdentityMap, NZ>(z);
      if constexpr (S > 0) { __builtin_amdgcn_sched_barrier(0); load_twiddle_bases(wa, wb); }
      static_for<1, RF>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value, ka = k1 % RAF, kb = k1 / RAF, pos = out_pos<RF>(k1);
        if constexpr (ka > 0) z[pos] = cmul(z[pos], wa[ka]);
        if constexpr (kb > 0) z[pos] = cmul(z[pos], wb[kb]);
      });
    }
    gate_commit();                                   // this tile's bins (requested behind the previous tile's stores) -> LDS
    __builtin_amdgcn_sched_barrier(0);
    // ---- the quiet part of the tile starts: the deferred results of the previous tile leave, the same row blocks of the next tile
    //      are requested into the registers they vacate
    if constexpr ((XP & 16) == 0) static_for<D0, D0 + P>([&](auto ic) { store_row(rs_prev, ooff, out_sn, row_q(ic), dfr[decltype(ic)::value - D0]); });
    // XP bits 2 / 3 (tools/mixedp_stage_bench.hip): the deferred loads behind the barriers of E1 (5 gaps) / of E1 and E2 (9 gaps) instead of here
    // (kernel_regtile64p.h SPREAD: a wave that has just passed a barrier of an exchange waits for the LDS anyway)
    constexpr int NGAP = (XP & 8) != 0 ? 9 : (XP & 4) != 0 ? 5 : 0;
    constexpr bool GAP_ST = (XP & 16) != 0;          // bit 4: the deferred STORES over E1's five gaps, the loads over E2's four
    [[maybe_unused]] auto gap_loads = [&](auto kc) {
      if constexpr (GAP_ST) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < 5) static_for<D0 + k * P / 5, D0 + (k + 1) * P / 5>([&](auto ic) { store_row(rs_prev, ooff, out_sn, row_q(ic), dfr[decltype(ic)::value - D0]); });
        else static_for<D0 + (k - 5) * P / 4, D0 + (k - 4) * P / 4>([&](auto ic) { dfr[decltype(ic)::value - D0] = load_row(rs_next, voff, v_sn, row_q(ic)); });
      } else
      if constexpr (NGAP > 0 && decltype(kc)::value < NGAP) {
        constexpr int k = decltype(kc)::value;
        static_for<D0 + k * P / NGAP, D0 + (k + 1) * P / NGAP>([&](auto ic) { dfr[decltype(ic)::value - D0] = load_row(rs_next, voff, v_sn, row_q(ic)); });
      }
    };
    if constexpr (NGAP == 0 && !GAP_ST) static_for<D0, D0 + P>([&](auto ic) { dfr[decltype(ic)::value - D0] = load_row(rs_next, voff, v_sn, row_q(ic)); });
    __builtin_amdgcn_sched_barrier(0);

    // ---- E1 (kernel_regtile_mixed.h); the first barrier also separates it from the previous tile's E2 reads ---------------
    // Writes: ds_write_addtid_b32 from one of two M0 bases.  Reads: single ds_read_b32 (mp_lds_read), pinned behind the barrier that
    // waits for them.
    const uint32_t r1 = (uint32_t)(uintptr_t)lane_base(u * ROW1 + p), r2 = (uint32_t)(uintptr_t)lane_base(u * ROW2 + p);
    // a write goes to img[off + tid]: consecutive lanes, consecutive dwords — ds_write_addtid_b32 (mp_write_addtid) from the wave's base
    const MixedM0 m0 = mixed_m0(img, tid);
    auto wr = [&](auto offc, float v) { mixed_write_addtid<decltype(offc)::value * 4>(v, m0); };
    rt_lds_barrier();
    if constexpr (TICKETS) { if (tk_lead) tk.publish(); }      // (the barrier waited for lgkmcnt(0): the ticket is here)
    gap_loads(std::integral_constant<int, 0>{});
    if (rows) static_for<0, RF>([&](auto kc) { constexpr int k1 = decltype(kc)::value; wr(std::integral_constant<int, k1 * ROW1>{}, z[out_pos<RF>(k1)].x); });
    rt_lds_barrier();
    if constexpr (TICKETS) { if (tk_lead) tk.result(tid0 & 63); }
    gap_loads(std::integral_constant<int, 1>{});
    if (bins) static_for<0, RS>([&](auto nc) { constexpr int n2 = decltype(nc)::value; z[n2].x = mp_lds_read<n2 * kPC * 4>(r1); });
    rt_lds_barrier();
    mp_pin<0, RS, false>(z);
    gap_loads(std::integral_constant<int, 2>{});
    if (rows) static_for<0, RF>([&](auto kc) { constexpr int k1 = decltype(kc)::value; wr(std::integral_constant<int, k1 * ROW1>{}, z[out_pos<RF>(k1)].y); });
    rt_lds_barrier();
    gap_loads(std::integral_constant<int, 3>{});
    if (bins) static_for<0, RS>([&](auto nc) { constexpr int n2 = decltype(nc)::value; z[n2].y = mp_lds_read<n2 * kPC * 4>(r1); });
    rt_lds_barrier();
    mp_pin<0, RS, true>(z);
    gap_loads(std::integral_constant<int, 4>{});
    if constexpr (TICKETS) { if (tk_foll) tk.ask(); }

    // ---- bins k = u + RF*k2: F2 over n2, gate, I1 over k2 ------------------------------------------------------------------
    using BinMap = OutPosMap<RS>;
    if (bins) {
      fft_ct<RS, false, IdentityMap, NZ>(z);
      const float2* g_lo = glds + u;                 // g_lo[RF k2] = g[k], g_hi[N - RF k2] = g[N - k]
      const float2* g_hi = glds - u;
      static_for<0, RS>([&](auto kc) {
        constexpr int k2 = decltype(kc)::value, pos = BinMap::at(k2);
        // Hermitian extension above N/2: conj(g[N - k]).  Which side bin k = u + RF*k2 is on is known at compile time for all k2 but
        // the one or two that straddle N/2, so the reads are immediate offsets from two lane constants and can all be in flight at once
        // (a per-bin select serialised them behind s_waitcnt lgkmcnt(0): 24 exposed LDS latencies per tile)
        constexpr int side = mixed_gate_side<RF, N>(k2);
        if constexpr (side == 0) z[pos] = cmul(z[pos], g_lo[RF * k2]);                  // spectre.py:545
        else if constexpr (side == 1) z[pos] = cmulc(z[pos], g_hi[N - RF * k2]);
        else {
          const bool upper = 2 * (u + RF * k2) > N;
          float2 g = upper ? g_hi[N - RF * k2] : g_lo[RF * k2];
          if (upper) g.y = -g.y;
          z[pos] = cmul(z[pos], g);
        }
      });
      fft_ct<RS, true, BinMap, NZ>(z);
    }

    // ---- E2 ---------------------------------------------------------------------------------------------------------------
    if (bins) static_for<0, RS>([&](auto nc) { constexpr int n2 = decltype(nc)::value; wr(std::integral_constant<int, n2 * ROW2>{}, z[BinMap::at(out_pos<RS>(n2))].x); });
    if constexpr (TICKETS) { if (tk_foll) tk.check(); }
    rt_lds_barrier();
    gap_loads(std::integral_constant<int, 5>{});
    if (rows) static_for<0, RF>([&](auto kc) { constexpr int k1 = decltype(kc)::value; z[k1].x = mp_lds_read<k1 * kPC * 4>(r2); });
    rt_lds_barrier();
    if constexpr (TICKETS) { if (tk_foll) tk.result(tid0 & 63); }
    mp_pin<0, RF, false>(z);
    gap_loads(std::integral_constant<int, 6>{});
    if (bins) static_for<0, RS>([&](auto nc) { constexpr int n2 = decltype(nc)::value; wr(std::integral_constant<int, n2 * ROW2>{}, z[BinMap::at(out_pos<RS>(n2))].y); });
    rt_lds_barrier();
    gap_loads(std::integral_constant<int, 7>{});
    if (rows) static_for<0, RF>([&](auto kc) { constexpr int k1 = decltype(kc)::value; z[k1].y = mp_lds_read<k1 * kPC * 4>(r2); });
    rt_lds_barrier();
    mp_pin<0, RF, true>(z);
    gap_loads(std::integral_constant<int, 8>{});

    // ---- conj twiddle, I2 over k1, store rows u + RS*n1 (spectre.py:553), reload / trade places -------------------------------
    [[maybe_unused]] int fut_tile = -2;
    if constexpr (TICKETS) { if (round == 0) fut_tile = tk.handed_over(); }
    load_twiddle_bases(wa, wb);
    if constexpr (S > 0) {
      // every wave is behind E2's last read: the image is idle, the next tile's first S row blocks may land in it.  The twiddle bases come
      // out of LDS and are read BEFORE the requests (hipcc orders every LDS read behind a pending LDS-DMA with s_waitcnt vmcnt(0)); the
      // prefetched row blocks are looked at here (requested a phase and a half ago) so that no wait for them is placed behind the requests.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      static_for<0, (P > 0 ? P : 1)>([&](auto ic) { mp_pin1(dfr[decltype(ic)::value].x); mp_pin1(dfr[decltype(ic)::value].y); });
      stage_coords(v_sn);
      static_for<0, S / 2>([&](auto jc) { stage_pair(rs_next, v_sn, jc); });
      asm volatile("" ::: "memory");
    }
    if (rows) {
      static_for<1, RF>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value, ka = k1 % RAF, kb = k1 / RAF;
        if constexpr (ka > 0) z[k1] = cmulc(z[k1], wa[ka]);
        if constexpr (kb > 0) z[k1] = cmulc(z[k1], wb[kb]);
      });
      fft_ct<RF, true, IdentityMap, NZ>(z);
    }
    // results: row block n1 lives at z[out_pos<RF>(n1)].  Everything is read out first (SSA values: no register moves), then the row
    // blocks of the next tile move in: reloaded behind their stores, or — the deferred ones — out of the registers that now keep results.
    // (Round 3 tried two other orders, profiles/r03_mixedp_ab_3_interleave.log and ..._4_grouped_release.log: every block reloaded right
    //  behind its own store, 1.63 against 1.50 ms; and I2's sub-transforms one at a time, each one's 15 results stored and reloaded before
    //  the next starts, with the registers in F1's order of use so that the first butterflies' blocks are requested first: 1.58 against
    //  1.56 ms.  Mixing the two directions at a finer grain costs more than the earlier requests gain.)
    float2 res[RF];
    static_for<0, RF>([&](auto nc) { constexpr int n1 = decltype(nc)::value; res[n1] = z[out_pos<RF>(n1)]; });
    if constexpr ((XP & 1) != 0) { static_for<0, RF>([&](auto nc) { mp_pin1(res[decltype(nc)::value].x); mp_pin1(res[decltype(nc)::value].y); }); rt_lds_barrier(); }
    static_for<0, RF>([&](auto ic) {
      constexpr int i = decltype(ic)::value, q = decltype(row_q(ic))::value;
      if constexpr (i < D0 || i >= D0 + P) store_row(rs_out, ooff, out_sn, std::integral_constant<int, q>{}, res[q]);
    });
    if (more) {
      static_for<D0, D0 + P>([&](auto ic) {
        constexpr int q = decltype(row_q(ic))::value, i = decltype(ic)::value - D0;
        const float2 nx = dfr[i];
        dfr[i] = res[q];
        z[q] = nx;
      });
    } else {
      static_for<D0, D0 + P>([&](auto ic) { constexpr int q = decltype(row_q(ic))::value; store_row(rs_out, ooff, out_sn, std::integral_constant<int, q>{}, res[q]); });
    }
    if constexpr ((XP & 2) != 0) rt_lds_barrier();
    static_for<S, RF>([&](auto ic) {
      constexpr int i = decltype(ic)::value, q = decltype(row_q(ic))::value;
      if constexpr (i < D0 || i >= D0 + P) z[q] = load_row(rs_next, voff, v_sn, std::integral_constant<int, q>{});
    });
    obp = ob;
    gate_fetch(gpn);                                 // (after the last tile: a harmless re-read of this tile's bins)
    if constexpr (TICKETS) {
      prev_live = cur_live && more;                   // (only then did this tile's deferred results go into dfr)
      cur_tile = nxt_tile; nxt_tile = fut_tile;
      if (round == 0) tk.advance(fut_tile);
    }
  }
  if constexpr (!TICKETS) break;
  }  // rounds
}

template <int RF, int RS>
hipError_t launch_regtile_mixedp(const RegtileArgs& a, hipStream_t stream);

}  // namespace sfft
