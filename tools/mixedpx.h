// mixedpx.h — a copy of fft_amd/csrc/kernel_regtile_mixedp.h (round 3) with its tidy-ups behind a template parameter, for
// tools/mixedp_opt_bench.hip only (profiles/r03_mixedp_opt_matrix.log).  OPT bits: 1 = compile-time gate side, 2 = single ds_read_b32 with
// immediate offsets, 4 = second write base beyond 64 KiB, 8 = the previous tile's output base recomputed instead of carried, 16 = exchange writes as
// ds_write_addtid_b32 (M0 + offset + 4 * lane).  The library
// kernel is OPT = 23 (1 + 2 + 4 + 16).  Not part of the product.
#pragma once
#include "../fft_amd/csrc/kernel_regtile_mixed.h"

namespace sfft {

// Bin k = u + RF*k2 (u < RF a lane quantity, k2 a compile-time one) against N/2: 0 = at or below for every u, 1 = above for every u
// (the gate is conj(g[N - k]) there), 2 = depends on u (the one or two k2 that straddle N/2).
template <int RF, int N> constexpr int xmixed_gate_side(int k2) { return 2 * (RF - 1 + RF * k2) <= N ? 0 : (2 * RF * k2 > N ? 1 : 2); }

__device__ __forceinline__ void xrt_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One float of the exchange image, read by a single ds_read_b32 with an immediate offset.  hipcc pairs neighbouring reads into
// ds_read2_b32, whose two results must sit in consecutive VGPRs — but x and y of one value already do (64-bit global loads and stores), so
// every pair costs a v_mov (320 per tile at 60 x 50).  The wave's LDS queue has room for twice the instructions; the VALU does not have
// room for the moves.  The result is not tracked by the compiler's s_waitcnt insertion: the caller reads behind xrt_lds_barrier() (which
// waits for lgkmcnt(0)) and then passes the values through xmp_pin() so that no use is scheduled above that barrier.
template <int OFF_BYTES>
__device__ __forceinline__ float xmp_lds_read(uint32_t addr) {
  static_assert(OFF_BYTES >= 0 && OFF_BYTES < 65536, "16-bit immediate");
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF_BYTES));
  return v;
}
template <int OFF_BYTES>
__device__ __forceinline__ void xmp_write_addtid(float v, uint32_t m0v) {
  static_assert(OFF_BYTES >= 0 && OFF_BYTES < 65536, "16-bit immediate");
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%2" :: "v"(v), "s"(m0v), "n"(OFF_BYTES) : "memory", "m0");
}
__device__ __forceinline__ void xmp_pin1(float& x) { asm volatile("" : "+v"(x)); }
template <int LO, int HI, bool IM, int NZ>
__device__ __forceinline__ void xmp_pin(float2 (&z)[NZ]) {
  static_for<LO, HI>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (IM) xmp_pin1(z[i].y); else xmp_pin1(z[i].x);
  });
}

// (Round 3 tried the 4096 kernel's early exchange writes here — E1's real plane written position by position behind the twiddle
//  multiplication, the image-freeing barrier moved in front of it or to the top of the tile: 1.87 / 1.89 ms against 1.59, and 1.24 against
//  1.15 ms without memory traffic; profiles/r03_mixedp_ab.log.  The scattered ds_write_b32 between the multiplications cost more than the
//  write phase they replace, and the earlier barrier exposes the slowest wave's loads.  Not kept.)
template <int RF, int RS, int P, int OPT = 7, bool FIRST = false>
__global__ void __launch_bounds__(kPC * (RF > RS ? RF : RS), 1) spectre_mix_regtile_mixedpx(const RegtileArgs a) {
  constexpr int D0 = FIRST ? 0 : RF - P;           // the deferred row blocks are [D0, D0 + P) of the order F1 uses them (the last ones: measured
                                                   // 1 % better than the first ones, 1.586 vs 1.605 ms)
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
  constexpr int HI = 16384;                          // floats: the second write base sits 64 KiB into the image
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
  if (pair_base >= a.n_tiles) return;

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
  auto row_q = [](auto ic) { constexpr int i = decltype(ic)::value; return std::integral_constant<int, (i / RAF) + RBF * (i % RAF)>{}; };
  auto load_row = [&](__amdgpu_buffer_rsrc_t rs, uint32_t voff, long long sn, auto qc) -> float2 {
    const rt_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + (uint32_t)((long long)decltype(qc)::value * RS * sn * 4), 0, 0);
    return make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
  };
  auto store_row = [&](__amdgpu_buffer_rsrc_t rs, uint32_t ooff, long long sn, auto qc, float2 v) {
    rt_u32x2 t;
    t.x = __float_as_uint(v.x); t.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(t, rs, ooff + (uint32_t)((long long)decltype(qc)::value * RS * sn * 4), 0, 0);
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
  auto load_twiddle_bases = [&](float2 (&wa)[RAF], float2 (&wb)[RBF]) {   // W_N^(u ka), W_N^(u RAF kb)
    const int uu = rows ? u : 0;
    static_for<1, RAF>([&](auto jc) { constexpr int j = decltype(jc)::value; wa[j] = a.tw[uu * j]; });
    static_for<1, RBF>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = a.tw[uu * RAF * j]; });
  };

  // ---- prologue: the whole first tile is loaded the way the non-deferred row blocks of every later tile are --------------------
  {
    const char* vb; char* ob; const float2* gp;
    tile_ptrs(pair_base, vb, ob, gp);
    const __amdgpu_buffer_rsrc_t rs = rsrc(vb, a.v_sn, a.rows_in);
    const uint32_t voff = rows ? (uint32_t)(((long long)u * a.v_sn + 2 * p) * 4) : 0x80000000u;
    static_for<0, RF>([&](auto ic) { constexpr int q = decltype(row_q(ic))::value; z[q] = load_row(rs, voff, a.v_sn, std::integral_constant<int, q>{}); });
    gate_fetch(gp);
  }

  for (int it = 0; it < a.tpw; ++it) {
    const int tile = pair_base + 2 * it;
    if (tile >= a.n_tiles) break;                    // workgroup-uniform
    const bool more = (it + 1 < a.tpw) && (tile + 2 < a.n_tiles);
    coords();
    long long v_sn = a.v_sn, out_sn = a.out_sn;
    asm volatile("" : "+s"(v_sn), "+s"(out_sn));
    const char* vb; char* ob; const float2* gp;
    tile_ptrs(tile, vb, ob, gp);
    const char* vbn = vb; char* obn = ob; const float2* gpn = gp;
    if (more) tile_ptrs(tile + 2, vbn, obn, gpn);
    if constexpr (OPT & 8) {                         // previous tile's output base recomputed instead of carried (no waterfall loops)
      const char* vbp = vb; const float2* gpp = gp;
      obp = ob;
      if (it > 0) tile_ptrs(tile - 2, vbp, obp, gpp);
    }
    const __amdgpu_buffer_rsrc_t rs_next = rsrc(vbn, v_sn, more ? a.rows_in : 0), rs_out = rsrc(ob, out_sn, a.rows_out),
                                 rs_prev = rsrc(obp, out_sn, it > 0 ? a.rows_out : 0);
    const uint32_t voff = rows ? (uint32_t)(((long long)u * v_sn + 2 * p) * 4) : 0x80000000u;
    const uint32_t ooff = rows ? (uint32_t)(((long long)u * out_sn + 2 * p) * 4) : 0x80000000u;

    // ---- rows u + RS*q, q < RF: F1 over q, W_N^(u k1) ---------------------------------------------------------------------
    float2 wa[RAF], wb[RBF];
    load_twiddle_bases(wa, wb);
    if (rows) {
      fft_ct<RF, false, IdentityMap, NZ>(z);
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
    static_for<D0, D0 + P>([&](auto ic) { store_row(rs_prev, ooff, out_sn, row_q(ic), dfr[decltype(ic)::value - D0]); });
    static_for<D0, D0 + P>([&](auto ic) { dfr[decltype(ic)::value - D0] = load_row(rs_next, voff, v_sn, row_q(ic)); });
    __builtin_amdgcn_sched_barrier(0);

    // ---- E1 (kernel_regtile_mixed.h); the first barrier also separates it from the previous tile's E2 reads ---------------
    // Writes: image rows beyond 64 KiB are addressed from a second lane constant (one v_add_u32 per write otherwise).  Reads: single
    // ds_read_b32 (xmp_lds_read), pinned behind the barrier that waits for them.
    lds_float* const w_lo = lane_base(u * kPC + p);
    lds_float* const w_hi = lane_base(u * kPC + p + HI);
    const uint32_t r1 = (uint32_t)(uintptr_t)lane_base(u * ROW1 + p), r2 = (uint32_t)(uintptr_t)lane_base(u * ROW2 + p);
    const uint32_t m0_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_float*)img + (uint32_t)(tid >> 6) * 256u), m0_hi = m0_lo + 49152u;
    auto wr = [&](auto offc, float v) {
      constexpr int off = decltype(offc)::value;
      if constexpr (OPT & 16) {                       // ds_write_addtid_b32: address = M0 + offset + 4 * lane, no address VGPR, 2 cycles instead of 4
        constexpr int byte = off * 4, kHiBase = 49152;
        if constexpr (byte < 65536 - 2048) xmp_write_addtid<byte>(v, m0_lo); else xmp_write_addtid<byte - kHiBase>(v, m0_hi);
      }
      else if constexpr (!(OPT & 4)) img[off + u * kPC + p] = v;
      else if constexpr (off * 4 + 4096 < 65536) w_lo[off] = v; else w_hi[off - HI] = v;
    };
    xrt_lds_barrier();
    if (rows) static_for<0, RF>([&](auto kc) { constexpr int k1 = decltype(kc)::value; wr(std::integral_constant<int, k1 * ROW1>{}, z[out_pos<RF>(k1)].x); });
    xrt_lds_barrier();
    if (bins) static_for<0, RS>([&](auto nc) { constexpr int n2 = decltype(nc)::value; if constexpr (OPT & 2) z[n2].x = xmp_lds_read<n2 * kPC * 4>(r1); else z[n2].x = img[u * ROW1 + n2 * kPC + p]; });
    xrt_lds_barrier();
    if constexpr (OPT & 2) xmp_pin<0, RS, false>(z);
    if (rows) static_for<0, RF>([&](auto kc) { constexpr int k1 = decltype(kc)::value; wr(std::integral_constant<int, k1 * ROW1>{}, z[out_pos<RF>(k1)].y); });
    xrt_lds_barrier();
    if (bins) static_for<0, RS>([&](auto nc) { constexpr int n2 = decltype(nc)::value; if constexpr (OPT & 2) z[n2].y = xmp_lds_read<n2 * kPC * 4>(r1); else z[n2].y = img[u * ROW1 + n2 * kPC + p]; });
    xrt_lds_barrier();
    if constexpr (OPT & 2) xmp_pin<0, RS, true>(z);

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
        constexpr int side = (OPT & 1) ? xmixed_gate_side<RF, N>(k2) : 2;
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
    xrt_lds_barrier();
    if (rows) static_for<0, RF>([&](auto kc) { constexpr int k1 = decltype(kc)::value; if constexpr (OPT & 2) z[k1].x = xmp_lds_read<k1 * kPC * 4>(r2); else z[k1].x = img[u * ROW2 + k1 * kPC + p]; });
    xrt_lds_barrier();
    if constexpr (OPT & 2) xmp_pin<0, RF, false>(z);
    if (bins) static_for<0, RS>([&](auto nc) { constexpr int n2 = decltype(nc)::value; wr(std::integral_constant<int, n2 * ROW2>{}, z[BinMap::at(out_pos<RS>(n2))].y); });
    xrt_lds_barrier();
    if (rows) static_for<0, RF>([&](auto kc) { constexpr int k1 = decltype(kc)::value; if constexpr (OPT & 2) z[k1].y = xmp_lds_read<k1 * kPC * 4>(r2); else z[k1].y = img[u * ROW2 + k1 * kPC + p]; });
    xrt_lds_barrier();
    if constexpr (OPT & 2) xmp_pin<0, RF, true>(z);

    // ---- conj twiddle, I2 over k1, store rows u + RS*n1 (spectre.py:553), reload / trade places -------------------------------
    load_twiddle_bases(wa, wb);
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
    static_for<0, RF>([&](auto ic) {
      constexpr int i = decltype(ic)::value, q = decltype(row_q(ic))::value;
      if constexpr (i < D0 || i >= D0 + P) z[q] = load_row(rs_next, voff, v_sn, std::integral_constant<int, q>{});
    });
    if constexpr (!(OPT & 8)) obp = ob;
    gate_fetch(gpn);                                 // (after the last tile: a harmless re-read of this tile's bins)
  }
}

template <int RF, int RS>
hipError_t launch_regtile_mixedpx(const RegtileArgs& a, hipStream_t stream);

}  // namespace sfft
