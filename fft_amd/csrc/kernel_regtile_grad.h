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
  long long v_sb, v_sn, dout_sb, dout_sn;   // element strides
};

template <int RF, int RS> constexpr int gate_grad_partner_bytes() { return 2 * RF * kPC * (RS / 2 + 1) * 4; }
template <int RF, int RS, int XV = SFFT_EXCHANGE_B128(RF, RS)> constexpr int gate_grad_image_bytes() {
  return regtile_image_bytes<RF, RS, XV>() > gate_grad_partner_bytes<RF, RS>() ? regtile_image_bytes<RF, RS, XV>()
                                                                                 : gate_grad_partner_bytes<RF, RS>();
}
template <int RF, int RS> constexpr int gate_grad_lds_total() { return gate_grad_image_bytes<RF, RS>() + (RF * RS / 2 + 1) * 8; }

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
template <int RF, int RS, bool IO_BF16, bool GENERAL, int XV = SFFT_EXCHANGE_B128(RF, RS)>
__global__ void __launch_bounds__(kPC * RS) spectre_gate_grad_regtile(const GateGradArgs a) {
  static_assert(RF == RS || RF == 2 * RS, "n_fft = RS*RS or 2*RS*RS");
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

  const int wg_lin = xcd_contiguous(blockIdx.x, a.n_wg);
  const int s = wg_lin % a.S, bg = wg_lin / a.S;
  const int b = bg / a.G, g = bg - b * a.G;

  for (int k = tid; k <= N / 2; k += kPC * RS) acc[k] = make_float2(0.f, 0.f);   // ordered by E1's barriers

  // Loads: buffer instructions — workgroup-uniform base of the tile (SGPRs) + one 32-bit lane offset + the row-block offset.  Fast
  // mode: the row-block offset is the instruction's scalar offset (no VALU at all).  GENERAL: it is added to the lane offset, because
  // the range check covers only that operand: rows >= N_in and the lanes of a ragged last tile (lane offset 0x80000000) are the
  // out-of-range case, which returns 0 = rfft's zero padding — no predicates, no pointer selects.
  float2 z[RF];
  // live = false (after the workgroup's last tile): an empty range.  The requests for the next tile are issued unconditionally — hipcc
  // counts only requests that are guaranteed to be younger when it computes a wait, see kernel_regtile64p.h.
  auto load_row = [&](int jt_, auto qc, long long v_sn, long long d_sn, int p, int u, bool live = true) {
    constexpr int q = decltype(qc)::value;
    const int c0 = live ? g * a.d_g + kPC * jt_ : 0; // first channel of the tile
    const int rows = a.N_in < N ? a.N_in : N;
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.v)) + ((size_t)b * a.v_sb + c0) * ES, 0, !live ? 0 : GENERAL ? (int)(rows * v_sn * ES) : 0x7fffffff, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.dout)) + ((size_t)b * a.dout_sb + c0) * ES, 0, !live ? 0 : GENERAL ? (int)(rows * d_sn * ES) : 0x7fffffff, kRsrcFlags);
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
    z[q] = make_float2(x, dy);
  };
  {  // the first tile of this workgroup; every later one is requested inside the previous iteration
    long long v_sn = a.v_sn, d_sn = a.dout_sn;
    static_for<0, RF>([&](auto ic) {
      constexpr int q = (decltype(ic)::value / RAF) + RBF * (decltype(ic)::value % RAF);   // order of use in F1
      load_row(s, std::integral_constant<int, q>{}, v_sn, d_sn, p0, u0);
    });
  }

  for (int jt = s; jt < a.T; jt += a.S) {
    int p = p0, u = u0;
    asm volatile("" : "+v"(p), "+v"(u));            // see kernel_regtile.h: keeps per-lane addresses out of LICM
    long long v_sn = a.v_sn, d_sn = a.dout_sn;
    asm volatile("" : "+s"(v_sn), "+s"(d_sn));
    const bool more = jt + a.S < a.T;                // workgroup-uniform

    // ---- F1 + W_N^(u*k1) ------------------------------------------------------------------------------------
    {
      fftA_stage1<RAF, RBF, false>(z);
      float2 wa[RAF], wb[RBF];
      __builtin_amdgcn_sched_barrier(0);
      static_for<1, RAF>([&](auto jc) { constexpr int j = decltype(jc)::value; wa[j] = a.tw[u * j]; });
      static_for<1, RBF>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = a.tw[u * RAF * j]; });
      static_for<0, RAF>([&](auto kac) { fftA_stage2_group<RAF, RBF, false, decltype(kac)::value>(z); });
      static_for<1, RF>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int ka = j / RBF, kb = j % RBF;
        if constexpr (ka > 0) z[j] = cmul(z[j], wa[ka]);
        if constexpr (kb > 0) z[j] = cmul(z[j], wb[kb]);
      });
    }

    // ---- E1 (identical to the forward kernel) ------------------------------------------------------------------
    if constexpr (XV) {
      constexpr int PS = RS + 4, RW = kPC * PS;
      exchange_planes_b128_w2<RF, RAS, RBS, true, RF, 1, 0, RW>(z, img, p * PS + u,
          [](auto rc, auto) { constexpr int k1 = decltype(rc)::value; return std::integral_constant<int, RBF * (k1 % RAF) + k1 / RAF>{}; },
          [&](auto mc) { constexpr int m = decltype(mc)::value; constexpr int t = m / RS, n2 = m % RS;
                         return (u + RS * t) * RW + p * PS + n2; });
    } else {
      exchange_planes<RF, RAS, RBS>(z, img,
          [&](auto jc) { constexpr int j = decltype(jc)::value; constexpr int k1 = (j / RBF) + RAF * (j % RBF);
                         return k1 * ROW1 + u * kPC + p; },
          [&](auto mc) { constexpr int m = decltype(mc)::value; constexpr int t = m / RS, n2 = m % RS;
                         return (u + RS * t) * ROW1 + n2 * kPC + p; });
    }

    // ---- F2: position t*RS + RBS*ka + kb holds A[k1 + RF*k2], k1 = u + RS*t, k2 = ka + RAS*kb ------------------
    static_for<0, NS>([&](auto tc) { fftA<RAS, RBS, false, decltype(tc)::value * RS, RF>(z); });

    // ---- partner exchange: upper half (k2 >= RS/2, i.e. kb >= RBS/2) of every set goes to LDS.  A register that has been written
    //      is dead: the row of the NEXT tile that lives at its position is requested into it right away, so that half of the next
    //      tile is in flight while the products are formed (and the other half follows register by register as they are consumed).
    const float nyq = z[RBS / 2].x * z[RBS / 2].y;   // Nyquist (set 0, ka = 0, kb = RBS/2): needed after its register is reused
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
          load_row(jt + a.S, std::integral_constant<int, j>{}, v_sn, d_sn, p, u, more);
        });
      });
    });
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
          load_row(jt + a.S, std::integral_constant<int, j>{}, v_sn, d_sn, p, u, more);       // z[j] is dead
          if constexpr (kGateGradSpread) load_row(jt + a.S, std::integral_constant<int, j + RBS / 2>{}, v_sn, d_sn, p, u, more);   // ... and so is its upper-half twin
        });
      });
      if constexpr (t == 0) {                         // Nyquist: k1 = 0, k2 = RS/2 (ka = 0, kb = RBS/2): Re(A) Im(A)
        const float sr = team_sum8(k1z ? nyq : 0.f);
        if (p == 0 && k1z) acc[N / 2].x += sr;
      }
    });
    lds_barrier();                                    // partner image is read; next tile's E1 may overwrite it
  }

  __syncthreads();
  float2* dst = a.part + ((size_t)bg * a.S + s) * a.F;
  for (int k = tid; k <= N / 2; k += kPC * RS) dst[k] = acc[k];
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

#define SFFT_DEFINE_GATE_GRAD_LAUNCHER(RF_, RS_)                                                             \
  template <>                                                                                                \
  hipError_t launch_gate_grad_regtile<RF_, RS_>(const GateGradArgs& a, bool io_bf16, bool general,           \
                                                hipStream_t stream) {                                        \
    const dim3 grid(a.n_wg), block(regtile_threads<RF_, RS_>());                                             \
    const size_t lds = gate_grad_lds_total<RF_, RS_>();                                                      \
    const int key = (io_bf16 ? 2 : 0) | (general ? 1 : 0);                                                   \
    static std::atomic<bool> lds_opt_in[16][4];                                                                      \
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
    switch (key) {                                                                                           \
      case 0: return go(spectre_gate_grad_regtile<RF_, RS_, false, false>);                                  \
      case 1: return go(spectre_gate_grad_regtile<RF_, RS_, false, true>);                                   \
      case 2: return go(spectre_gate_grad_regtile<RF_, RS_, true, false>);                                   \
      default: return go(spectre_gate_grad_regtile<RF_, RS_, true, true>);                                   \
    }                                                                                                        \
  }

}  // namespace sfft
