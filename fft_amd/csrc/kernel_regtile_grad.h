// kernel_regtile_grad.h — register-resident gate gradient for n_fft = RF*RS (256 ... 4096) on gfx950.
//
// Backward of /root/reference/spectre.py:545 with respect to the filter (what autograd derives there):
//   dgate[b,g,k] = (w_k / N) * sum_{c in g} conj(X_c[k]) * R_c[k],   X = rfft(V), R = rfft(dOut),  w_k = 2 (1 at DC/Nyquist).
//
// One complex transform yields both spectra of a channel: with z = x_c + i*dy_c and A = DFT_N(z),
//   X[k] = (A[k] + conj(A[N-k])) / 2,   R[k] = (A[k] - conj(A[N-k])) / (2i)
//   conj(X[k]) R[k] = Im(A[k] A[N-k]) / 2  -  i (|A[k]|^2 - |A[N-k]|^2) / 4,
// so the inverse transform disappears and the only cross-thread traffic after the forward transform is the partner
// bin A[N-k].  V and dOut are read exactly once; nothing but (B, G, F) partial sums is written.
//
// Geometry.  A tile is 8 channels (one per lane p) x all n_fft rows, the same thread <-> row mapping, F1, twiddle and
// E1 exchange as kernel_regtile.h; F2 leaves thread (p,u), set t with bins k = k1 + RF*k2, k1 = u + RS*t.  Bins with
// k2 < RS/2 are the half spectrum (plus k2 = RS/2 for k1 = 0: Nyquist); their partners (RF - k1, RS - 1 - k2), or
// (0, RS - k2) for k1 = 0, all have k2 >= RS/2: every thread writes its upper RS/2 values of each set to LDS and reads
// its partners' — half an exchange.  The 8 channels are summed with DPP adds; lane 0 of each 8-lane team adds the sums
// into a half-spectrum accumulator in LDS that lives across the workgroup's tiles.
//
// Work split.  A 32-byte row segment is a quarter of a 128-byte line; the S workgroups that share one (batch, group)
// take channel tiles s, s+S, ... and are adjacent in the XCD-contiguous order, so the four quarters of a line are
// requested from the same L2 within microseconds of each other (S is a multiple of 4 whenever the group has >= 4
// tiles).  Each workgroup stores its unscaled partial sums; spectre_gate_grad_regtile_finish adds the S partials in a
// fixed order (deterministic, no atomics) and applies w_k / N.
#pragma once
#include "kernel_regtile.h"

namespace sfft {

struct GateGradArgs {
  const void* v;        // (B, N_in, D) f32 | bf16
  const void* dout;     // (B, min(N_in, n_fft), D) same dtype
  float2* part;         // (B*G, S, F) partial sums
  const float2* tw;     // exp(-2 pi i m / N)
  int B, N_in, D, G, d_g, F;
  int T, S;             // channel tiles per group, workgroups per (batch, group)
  int n_wg;             // B * G * S
  int grid;             // prefetch form: workgroups to launch (one per CU, a multiple of S; the kernel walks through the work items)
  int prefetch;         // != 0: the prefetch form where it exists (64 x 64, fast mode); 0: one work item per workgroup everywhere
  long long v_sb, v_sn, dout_sb, dout_sn;   // element strides
};

template <int RF, int RS> constexpr int gate_grad_partner_bytes() { return 2 * RF * kPC * (RS / 2 + 1) * 4; }
template <int RF, int RS, int XV = SFFT_EXCHANGE_B128(RF, RS)> constexpr int gate_grad_image_bytes() {
  return regtile_image_bytes<RF, RS, XV>() > gate_grad_partner_bytes<RF, RS>() ? regtile_image_bytes<RF, RS, XV>()
                                                                                 : gate_grad_partner_bytes<RF, RS>();
}
template <int RF, int RS> constexpr int gate_grad_tw_off() { return (gate_grad_image_bytes<RF, RS>() + (RF * RS / 2 + 1) * 8 + 15) & ~15; }
template <int RF> constexpr int gate_grad_tw_row() { return FftCfg<RF>::RA + FftCfg<RF>::RB - 2; }     // float2 per row class u
// PN > 0 (prefetch form): F1's twiddle bases W^(u j), j < RA, and W^(u RA j), j < RB, live in LDS (written once per workgroup)
template <int RF, int RS, int PN = 0> constexpr int gate_grad_lds_total() {
  return PN > 0 ? gate_grad_tw_off<RF, RS>() + RS * gate_grad_tw_row<RF>() * 8 : gate_grad_image_bytes<RF, RS>() + (RF * RS / 2 + 1) * 8;
}

constexpr bool kGateGradSpread = false;  // round 4: the 64-load burst behind the partner exchange moved into the product phase, one more pair per
                                         // product: 2.66 -> 2.75 ms (f32), 2.19 -> 2.26 (bf16) on one box — the product phase is request-bound already; not shipped
// Workgroup barrier that orders LDS traffic only: __syncthreads() is a fence + s_barrier and hipcc implements the fence with
// s_waitcnt vmcnt(0), which would wait for the next tile's rows requested just before it (kernel_regtile64p.h has the long story).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// sum over the 8 lanes of a team (lanes 8r .. 8r+7): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
__device__ __forceinline__ float team_sum8(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
  return v;
}

// GENERAL: row predicates (N_in < n_fft) and channel predicates (d_g % 8 != 0).
// PN (round 5): PREFETCH registers.  s_memtime had shown the two request phases of a tile (the 64-load burst behind the partner writes,
// the 64 loads behind the products) running AT the 32-byte pattern's load-only rate (3.5 TB/s) for 44 k of the tile's 72 k cycles, and
// the memory system idle through F1 / E1 / F2 (PMC: 385 read-credit stall cycles per launch): all 2 RF data registers are busy there.
// But a 512-thread workgroup may use 256 registers, and the transforms need ~ 60 besides the data: PN of the next tile's row blocks —
// the LAST PN of the lower half in product order, i.e. the requests F1 used to wait for — are requested into 2 PN registers of their
// own a few at a time between the butterfly groups of F1 and F2 and behind the inner barriers of E1 (which therefore orders LDS traffic
// only), and copied into their data registers as those die in the product phase.
// EARLY: the partner writes of the upper half and the requests into those registers follow each butterfly group of F2 instead of
// forming one burst behind it.
template <int RF, int RS, bool IO_BF16, bool GENERAL, int XV = SFFT_EXCHANGE_B128(RF, RS), int PN = 0, bool EARLY = false>
__global__ void __launch_bounds__(kPC * RS) spectre_gate_grad_regtile(const GateGradArgs a) {
  static_assert(!EARLY || PN > 0, "EARLY belongs to the prefetch form (E1 ends with an LDS-only barrier there)");
  static_assert(RF == RS || RF == 2 * RS, "n_fft = RS*RS or 2*RS*RS");
  static_assert(PN == 0 || (XV && !GENERAL && PN <= RF / 2), "prefetch form: fast mode, b128 exchange, at most the lower half");
  constexpr int N = RF * RS, NS = RF / RS;
  constexpr int RAF = FftCfg<RF>::RA, RBF = FftCfg<RF>::RB;
  constexpr int RAS = FftCfg<RS>::RA, RBS = FftCfg<RS>::RB;
  constexpr int ROW1 = RS * kPC + kPC;
  constexpr int ES = IO_BF16 ? 2 : 4;
  constexpr int PS2 = RS / 2 + 1, RW2 = kPC * PS2, PLANE2 = RF * RW2;   // partner image: [k1][p][k2 - RS/2], odd column stride
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);
  float2* acc = reinterpret_cast<float2*>(smem + gate_grad_image_bytes<RF, RS, XV>());

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int p0 = lane & (kPC - 1);
  const int u0 = (lane / kPC) + (64 / kPC) * wave;

  // Work items = (batch, group, s) triples, a.n_wg of them.  PN == 0: one per workgroup (the grid has a.n_wg workgroups).  Prefetch
  // form: PERSISTENT — the grid has one workgroup per CU, workgroup L takes items L, L + grid, L + 2 grid, ... and the first tile of the
  // next item is requested during the last tile of the current one like any other next tile (the 16 items of a CU at the headline shape
  // used to begin with 16 exposed tile loads, ~12 % of the launch).  The S workgroups of one (batch, group) stay neighbours on one XCD.
  const int n_grid = PN > 0 ? (int)gridDim.x : a.n_wg;
  int item = xcd_contiguous(blockIdx.x, n_grid);
  int fb, fg;                                        // (batch, group) of the tile being REQUESTED (the first tile, then always the next one)
  { const int bg0 = item / a.S; fb = bg0 / a.G; fg = bg0 - fb * a.G; }

  for (int k = tid; k <= N / 2; k += kPC * RS) acc[k] = make_float2(0.f, 0.f);   // ordered by E1's barriers
  // prefetch form: the twiddle bases of every row class in LDS — as global loads they were 14 of a thread's 142 requests per tile, sat in
  // the same in-order counter as the prefetched rows, and all 28 registers stayed live through F1's second stage
  constexpr int TWR = gate_grad_tw_row<RF>();
  float2* twl = reinterpret_cast<float2*>(smem + gate_grad_tw_off<RF, RS>());
  if constexpr (PN > 0) {
    static_assert(gate_grad_lds_total<RF, RS, PN>() <= 160 * 1024, "LDS budget");
    for (int i = tid; i < RS * TWR; i += kPC * RS) {
      const int uu = i / TWR, jj = i - uu * TWR;
      twl[i] = a.tw[jj < RAF - 1 ? uu * (jj + 1) : uu * RAF * (jj - (RAF - 1) + 1)];
    }
  }

  // Loads: buffer instructions — workgroup-uniform base of the tile (SGPRs) + one 32-bit lane offset + the row-block offset.  Fast
  // mode: the row-block offset is the instruction's scalar offset (no VALU at all).  GENERAL: it is added to the lane offset, because
  // the range check covers only that operand: rows >= N_in and the lanes of a ragged last tile (lane offset 0x80000000) are the
  // out-of-range case, which returns 0 = rfft's zero padding — no predicates, no pointer selects.
  float2 z[RF];
  char *req_v = nullptr, *req_d = nullptr;           // prefetch form: base addresses of the tile being requested
  auto set_request = [&](int jt_) {
    const int c0 = fg * a.d_g + kPC * jt_;
    req_v = const_cast<char*>(reinterpret_cast<const char*>(a.v)) + ((size_t)fb * a.v_sb + c0) * ES;
    req_d = const_cast<char*>(reinterpret_cast<const char*>(a.dout)) + ((size_t)fb * a.dout_sb + c0) * ES;
    asm volatile("" : "+s"(req_v), "+s"(req_d));
  };
  // live = false (after the workgroup's last tile): an empty range.  The requests for the next tile are issued unconditionally — hipcc
  // counts only requests that are guaranteed to be younger when it computes a wait, see kernel_regtile64p.h.
  auto fetch_row = [&](int jt_, auto qc, long long v_sn, long long d_sn, int p, int u, bool live = true) -> float2 {
    constexpr int q = decltype(qc)::value;
    const int rows = a.N_in < N ? a.N_in : N;
    char *pv, *pd;
    if constexpr (PN > 0) {                          // the tile's two base addresses: formed once per tile (top of the loop), four SGPRs
      pv = req_v; pd = req_d;
    } else {
      const int c0 = live ? fg * a.d_g + kPC * jt_ : 0; // first channel of the tile
      pv = const_cast<char*>(reinterpret_cast<const char*>(a.v)) + ((size_t)fb * a.v_sb + c0) * ES;
      pd = const_cast<char*>(reinterpret_cast<const char*>(a.dout)) + ((size_t)fb * a.dout_sb + c0) * ES;
    }
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(pv, 0, !live ? 0 : GENERAL ? (int)(rows * v_sn * ES) : 0x7fffffff, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(pd, 0, !live ? 0 : GENERAL ? (int)(rows * d_sn * ES) : 0x7fffffff, kRsrcFlags);
    uint32_t vo = (uint32_t)(((long long)u * v_sn + p) * ES), dof = (uint32_t)(((long long)u * d_sn + p) * ES);
    uint32_t vs = (uint32_t)((long long)q * RS * v_sn * ES), ds = (uint32_t)((long long)q * RS * d_sn * ES);
    if constexpr (GENERAL) {
      const bool cok = kPC * jt_ + p < a.d_g;
      vo = cok ? vo + vs : 0x80000000u; dof = cok ? dof + ds : 0x80000000u;
      vs = 0; ds = 0;
    }
    float x, dy;
    if constexpr (IO_BF16) {
      x = __uint_as_float((uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rv, vo, vs, 0) << 16);
      dy = __uint_as_float((uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rd, dof, ds, 0) << 16);
    } else {
      x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, vo, vs, 0));
      dy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, dof, ds, 0));
    }
    return make_float2(x, dy);
  };
  auto load_row = [&](int jt_, auto qc, long long v_sn, long long d_sn, int p, int u, bool live = true) {
    z[decltype(qc)::value] = fetch_row(jt_, qc, v_sn, d_sn, p, u, live);
  };
  // prefetch registers: slot i <-> the (NLOW - PN + i)-th lower-half position in product order (t, ka, kb < RBS/2)
  constexpr int NLOW = NS * RAS * (RBS / 2), PNZ = PN > 0 ? PN : 1;
  float2 zn[PNZ];
  auto low_pos = [](auto ic) {                       // product-order index -> register position j
    constexpr int i = decltype(ic)::value, t = i / (RAS * (RBS / 2)), ka = (i / (RBS / 2)) % RAS, kb = i % (RBS / 2);
    return std::integral_constant<int, t * RS + RBS * ka + kb>{};
  };
  // request the slots of share `sl` of `nsl` (compile-time) for tile jt_.  The first TOPN slots are share "top" (sl = -1): issued at the
  // very top of the tile, IN FRONT of the wait for this tile's rows — the wave has to sit out that wait anyway, and without them the
  // request path would run dry from the last arrival until F1's first stage is through.
  constexpr int TOPN = PN / 4;                       // (0, 2, 6, 10, 12 of 20 / 24 measured: within 1 % of each other, 6 best)
  auto prefetch = [&](auto slc, auto nslc, int jt_, long long v_sn, long long d_sn, int p, int u, bool live) {
    if constexpr (PN > 0) {
      constexpr int sl = decltype(slc)::value, nsl = decltype(nslc)::value;
      constexpr int LO = sl < 0 ? 0 : TOPN + sl * (PN - TOPN) / nsl, HI = sl < 0 ? TOPN : TOPN + (sl + 1) * (PN - TOPN) / nsl;
      static_for<LO, HI>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        zn[i] = fetch_row(jt_, low_pos(std::integral_constant<int, NLOW - PN + i>{}), v_sn, d_sn, p, u, live);
      });
      __builtin_amdgcn_sched_barrier(0);             // keep the share where it is
    }
  };
  // EARLY: upper-half row blocks per butterfly group requested inside F2; the other RBS/2 - kE2 follow that group's products.  Measured at
  // (256, 4096, 768), PN = 24, same process: kE2 = 4 (all inside F2) -11.6 %, 3 -13.0 %, 2 -17.3 %, 1 -20.0 %, 0 -15.4 % against the form
  // without prefetch registers — F2 is dense arithmetic and does not take requests well, the product phase (LDS latencies, DPP chains) does.
  constexpr int kE2 = 1;
  constexpr int kS1 = RAF, kSE = 3;
  constexpr int kSlots = kS1 + kSE;                  // F1's stage-2 groups, E1's inner barriers
  {  // the first tile of this workgroup; every later one is requested inside the previous iteration
    long long v_sn = a.v_sn, d_sn = a.dout_sn;
    if constexpr (PN > 0) set_request(item % a.S);
    static_for<0, RF>([&](auto ic) {
      constexpr int q = (decltype(ic)::value / RAF) + RBF * (decltype(ic)::value % RAF);   // order of use in F1
      load_row(item % a.S, std::integral_constant<int, q>{}, v_sn, d_sn, p0, u0);
    });
  }

  if constexpr (PN > 0) lds_barrier();               // the twiddle rows are written (LDS only: the first tile's rows stay in flight)

  for (int jt = item % a.S;;) {
    // the tile after this one: the next of this item, or (prefetch form) the first of this workgroup's next item
    const bool item_ends = jt + a.S >= a.T;          // workgroup-uniform, like everything here
    int nitem = item, njt = jt + a.S;
    if (PN > 0 && item_ends) { nitem = item + n_grid; njt = nitem % a.S; }
    const bool cont = item_ends ? (PN > 0 && nitem < a.n_wg) : true;
    if (item_ends && cont) { const int nbg = nitem / a.S; fb = nbg / a.G; fg = nbg - fb * a.G; }
    if constexpr (PN > 0) set_request(cont ? njt : 0);
    int p = p0, u = u0;
    if constexpr (PN > 0) {                          // one register across the loop instead of two
      int t = tid;
      asm volatile("" : "+v"(t));
      p = t & (kPC - 1); u = t / kPC;               // = (lane & 7, lane / 8 + 8 wave)
    }
    asm volatile("" : "+v"(p), "+v"(u));            // see kernel_regtile.h: keeps per-lane addresses out of LICM
    long long v_sn = a.v_sn, d_sn = a.dout_sn;
    asm volatile("" : "+s"(v_sn), "+s"(d_sn));
#ifdef SFFT_DG_NO_ROWS                                // timing experiment (tools/build_variant.sh): no rows but the workgroup's first tile = the compute floor
    const bool more = false;
#else
    const bool more = cont;
#endif

    // ---- F1 + W_N^(u*k1) ------------------------------------------------------------------------------------
    {
      if constexpr (TOPN > 0) prefetch(std::integral_constant<int, -1>{}, std::integral_constant<int, 1>{}, njt, v_sn, d_sn, p, u, more);
      fftA_stage1<RAF, RBF, false>(z);
      float2 wa[RAF], wb[RBF];
      __builtin_amdgcn_sched_barrier(0);
      const float2* twu = twl + u * TWR;
      if constexpr (PN == 0) {
        static_for<1, RAF>([&](auto jc) { constexpr int j = decltype(jc)::value; wa[j] = a.tw[u * j]; });
        static_for<1, RBF>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = a.tw[u * RAF * j]; });
      }
      static_for<0, RAF>([&](auto kac) {
        fftA_stage2_group<RAF, RBF, false, decltype(kac)::value>(z);
        prefetch(kac, std::integral_constant<int, kSlots>{}, njt, v_sn, d_sn, p, u, more);
      });
      if constexpr (PN > 0) {                        // behind the butterflies: 2 (RB - 1) registers that stage 2 does not have to carry
        __builtin_amdgcn_sched_barrier(0);
        static_for<1, RBF>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = twu[RAF - 1 + j - 1]; });
      }
      static_for<1, RF>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int ka = j / RBF, kb = j % RBF;
        if constexpr (PN > 0 && ka > 0 && kb == 0) wa[ka] = twu[ka - 1];      // one ds_read_b64 per group, just in time
        if constexpr (ka > 0) z[j] = cmul(z[j], wa[ka]);
        if constexpr (kb > 0) z[j] = cmul(z[j], wb[kb]);
      });
    }

    // ---- E1 (identical to the forward kernel) ------------------------------------------------------------------
    if constexpr (XV) {
      constexpr int PS = RS + 4, RW = kPC * PS;
      exchange_planes_b128_w2<RF, RAS, RBS, true, RF, 1, 0, RW, (PN > 0)>(z, img, p * PS + u,
          [](auto rc, auto) { constexpr int k1 = decltype(rc)::value; return std::integral_constant<int, RBF * (k1 % RAF) + k1 / RAF>{}; },
          [&](auto mc) { constexpr int m = decltype(mc)::value; constexpr int t = m / RS, n2 = m % RS;
                         return (u + RS * t) * RW + p * PS + n2; },
          [&](auto hc) { prefetch(std::integral_constant<int, kS1 + decltype(hc)::value>{}, std::integral_constant<int, kSlots>{}, njt, v_sn, d_sn, p, u, more); });
    } else {
      exchange_planes<RF, RAS, RBS>(z, img,
          [&](auto jc) { constexpr int j = decltype(jc)::value; constexpr int k1 = (j / RBF) + RAF * (j % RBF);
                         return k1 * ROW1 + u * kPC + p; },
          [&](auto mc) { constexpr int m = decltype(mc)::value; constexpr int t = m / RS, n2 = m % RS;
                         return (u + RS * t) * ROW1 + n2 * kPC + p; });
    }

    // ---- F2: position t*RS + RBS*ka + kb holds A[k1 + RF*k2], k1 = u + RS*t, k2 = ka + RAS*kb ------------------
    float nyq = 0.f;                                  // Nyquist (set 0, ka = 0, kb = RBS/2): needed after its register is reused
    static_for<0, NS>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if constexpr (EARLY) {
        float* wre = img + (u + RS * t) * RW2 + p * PS2;
        fftA_stage1<RAS, RBS, false, t * RS, RF>(z);
        static_for<0, RAS>([&](auto kac) {
          constexpr int ka = decltype(kac)::value;
          fftA_stage2_group<RAS, RBS, false, ka, t * RS, RF>(z);
          if constexpr (t == 0 && ka == 0) nyq = z[RBS / 2].x * z[RBS / 2].y;
          static_for<RBS / 2, RBS>([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value, j = t * RS + RBS * ka + kb, k2 = ka + RAS * kb;
            wre[k2 - RS / 2] = z[j].x;
            wre[PLANE2 + k2 - RS / 2] = z[j].y;
          });
          static_for<RBS / 2, RBS / 2 + kE2>([&](auto kbc) {     // (the other RBS/2 - kE2 of the group: behind this group's products)
            load_row(njt, std::integral_constant<int, t * RS + RBS * ka + decltype(kbc)::value>{}, v_sn, d_sn, p, u, more);
          });
          __builtin_amdgcn_sched_barrier(0);
        });
      } else {
        fftA<RAS, RBS, false, t * RS, RF>(z);
      }
    });

    // ---- partner exchange: upper half (k2 >= RS/2, i.e. kb >= RBS/2) of every set goes to LDS.  A register that has been written
    //      is dead: the row of the NEXT tile that lives at its position is requested into it right away, so that half of the next
    //      tile is in flight while the products are formed (and the other half follows register by register as they are consumed).
    if constexpr (!EARLY) {
      nyq = z[RBS / 2].x * z[RBS / 2].y;
      static_for<0, NS>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        float* wre = img + (u + RS * t) * RW2 + p * PS2;
        static_for<0, RAS>([&](auto kac) {
          static_for<RBS / 2, RBS>([&](auto kbc) {
            constexpr int ka = decltype(kac)::value, kb = decltype(kbc)::value;
            constexpr int j = t * RS + RBS * ka + kb, k2 = ka + RAS * kb;
            wre[k2 - RS / 2] = z[j].x;
            wre[PLANE2 + k2 - RS / 2] = z[j].y;
          });
        });
      });
      // (kGateGradSpread: these 64 loads per thread one pair at a time behind the products below instead of back to back here — measured slower)
      if constexpr (!kGateGradSpread) static_for<0, NS>([&](auto tc) {
        static_for<0, RAS>([&](auto kac) {
          static_for<RBS / 2, RBS>([&](auto kbc) {
            constexpr int j = decltype(tc)::value * RS + RBS * decltype(kac)::value + decltype(kbc)::value;
            load_row(njt, std::integral_constant<int, j>{}, v_sn, d_sn, p, u, more);
          });
        });
      });
    }
    lds_barrier();                                    // (not __syncthreads(): its fence would wait for the requests just issued)
    static_for<0, NS>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      const int k1 = u + RS * t;
      const bool k1z = (k1 == 0);
      // partner row RF - k1 (0 for k1 = 0); partner k2' = RS - 1 - k2 (RS - k2 for k1 = 0)  ->  slot RS/2 - k2 - (k1 != 0)
      const float* rre = img + ((RF - k1) & (RF - 1)) * RW2 + p * PS2 + (RS / 2) - (k1z ? 0 : 1);
      static_for<0, RAS>([&](auto kac) {
        static_for<0, RBS / 2>([&](auto kbc) {
          constexpr int ka = decltype(kac)::value, kb = decltype(kbc)::value;
          constexpr int j = t * RS + RBS * ka + kb, k2 = ka + RAS * kb;
          float pr = rre[-k2], pi = rre[PLANE2 - k2];
          if constexpr (k2 == 0) {                    // DC is its own partner (the slot read above is padding)
            pr = k1z ? z[j].x : pr;
            pi = k1z ? z[j].y : pi;
          }
          const float q = z[j].x * pi + z[j].y * pr;                                   // Im(A A')
          const float e = (z[j].x * z[j].x + z[j].y * z[j].y) - (pr * pr + pi * pi);   // |A|^2 - |A'|^2
          const float sr = team_sum8(0.5f * q), si = team_sum8(-0.25f * e);
          // bin k1 + RF*k2 belongs to this team alone, and after team_sum8 its eight lanes hold the same sums: ALL of them do the
          // read-modify-write (same address, same value).  Guarding it with `if (p == 0)` made every bin its own basic block — 32 serial
          // chains of LDS read, DPP adds, LDS read-modify-write per tile, nothing overlapped (s_memtime: the product phase took as long
          // as both transforms together).
          // (short transforms keep the guard: 0.52 -> 0.56 ms at (512,512,768) without it)
          if (N > 512 || p == 0) {
            float2 cur = acc[k1 + RF * k2];
            cur.x += sr; cur.y += si;
            acc[k1 + RF * k2] = cur;
          }
          constexpr int li = (t * RAS + ka) * (RBS / 2) + kb;                                   // product-order index of this position
          if constexpr (li >= NLOW - PN) z[j] = zn[li - (NLOW - PN)];                          // prefetched during F1 / E1 / F2
          else load_row(njt, std::integral_constant<int, j>{}, v_sn, d_sn, p, u, more);   // z[j] is dead
          if constexpr (kGateGradSpread) load_row(njt, std::integral_constant<int, j + RBS / 2>{}, v_sn, d_sn, p, u, more);   // ... and so is its upper-half twin
        });
        if constexpr (EARLY && kE2 < RBS / 2) static_for<RBS / 2 + kE2, RBS>([&](auto kbc) {
          load_row(njt, std::integral_constant<int, t * RS + RBS * decltype(kac)::value + decltype(kbc)::value>{}, v_sn, d_sn, p, u, more);
        });
      });
      if constexpr (t == 0) {                         // Nyquist: k1 = 0, k2 = RS/2 (ka = 0, kb = RBS/2): Re(A) Im(A)
        const float sr = team_sum8(k1z ? nyq : 0.f);
        if (p == 0 && k1z) acc[N / 2].x += sr;
      }
    });
    lds_barrier();                                    // partner image is read; next tile's E1 may overwrite it (and every accumulator update is done)
    if (item_ends) {                                  // this item's partial sums; a thread clears what it has read (the next update is barriers away)
      float2* dst = a.part + (size_t)item * a.F;      // item = (batch * G + group) * S + s
      int k0 = tid;
      asm volatile("" : "+v"(k0));                   // (nothing of this block is worth a register across the tile loop)
      for (int k = k0; k <= N / 2; k += kPC * RS) {
        dst[k] = acc[k];
        if (PN > 0) acc[k] = make_float2(0.f, 0.f);
      }
      if (!cont) break;
    }
    item = nitem; jt = njt;
  }
}

// dgate[bg, k] = (w_k / N) * sum_s part[bg, s, k]   (template only so that the header can be included by several TUs)
template <int UNUSED = 0>
__global__ void spectre_gate_grad_regtile_finish(const float2* __restrict__ part, float2* __restrict__ dgate, int S, int F, int n,
                                                 long long total) {
  const float inv_n = 1.0f / (float)n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long bg = i / F;
    const int k = (int)(i - bg * F);
    const float2* src = part + (size_t)bg * S * F + k;
    float2 sum = make_float2(0.f, 0.f);
    for (int s = 0; s < S; ++s) { sum.x += src[(size_t)s * F].x; sum.y += src[(size_t)s * F].y; }
    const float w = (k == 0 || 2 * k == n) ? inv_n : 2.0f * inv_n;
    dgate[i] = make_float2(sum.x * w, sum.y * w);
  }
}

template <int RF, int RS>
hipError_t launch_gate_grad_regtile(const GateGradArgs& a, bool io_bf16, bool general, hipStream_t stream);

constexpr int kGateGradPrefetch = 24;   // row blocks in prefetch registers of the 64 x 64 fast-mode kernel (20: within 1 %; 28: spills)
#define SFFT_DEFINE_GATE_GRAD_LAUNCHER(RF_, RS_)                                                             \
  template <>                                                                                                \
  hipError_t launch_gate_grad_regtile<RF_, RS_>(const GateGradArgs& a, bool io_bf16, bool general,           \
                                                hipStream_t stream) {                                        \
    dim3 grid(a.n_wg), block(regtile_threads<RF_, RS_>());                                                   \
    const int pn = (RF_ == 64 && RS_ == 64 && !general && a.prefetch != 0) ? 1 : 0;   /* prefetch form (persistent) */ \
    if (pn && a.grid > 0 && a.grid < a.n_wg) grid = dim3(a.grid);                                            \
    const size_t lds = pn ? gate_grad_lds_total<RF_, RS_, kGateGradPrefetch>() : gate_grad_lds_total<RF_, RS_>(); \
    const int key = (io_bf16 ? 2 : 0) | (general ? 1 : 0) | (pn << 2);                                       \
    static std::atomic<bool> lds_opt_in[16][8];                                                              \
    auto go = [&](auto kern) -> hipError_t {                                                                 \
      int dev = 0;                                                                                           \
      (void)hipGetDevice(&dev);                                                                              \
      if (dev < 0 || dev >= 16 || !lds_opt_in[dev][key]) {                                                   \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e != hipSuccess) return e;                                                                       \
        if (dev >= 0 && dev < 16) lds_opt_in[dev][key] = true;                                               \
      }                                                                                                      \
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                                 \
      return hipGetLastError();                                                                              \
    };                                                                                                       \
    if constexpr (RF_ == 64 && RS_ == 64) {                                                                  \
      constexpr int XV_ = SFFT_EXCHANGE_B128(RF_, RS_);                                                      \
      if (pn) return io_bf16 ? go(spectre_gate_grad_regtile<RF_, RS_, true, false, XV_, kGateGradPrefetch, true>)   \
                             : go(spectre_gate_grad_regtile<RF_, RS_, false, false, XV_, kGateGradPrefetch, true>);  \
    }                                                                                                        \
    switch (key & 3) {                                                                                       \
      case 0: return go(spectre_gate_grad_regtile<RF_, RS_, false, false>);                                  \
      case 1: return go(spectre_gate_grad_regtile<RF_, RS_, false, true>);                                   \
      case 2: return go(spectre_gate_grad_regtile<RF_, RS_, true, false>);                                   \
      default: return go(spectre_gate_grad_regtile<RF_, RS_, true, true>);                                   \
    }                                                                                                        \
  }

}  // namespace sfft
