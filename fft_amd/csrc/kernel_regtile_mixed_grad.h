// kernel_regtile_mixed_grad.h — register-resident gate gradient for the mixed-radix tile lengths (n_fft = RF * RS with
// RS even: 3000 = 60 x 50, ...).  Same mathematics and work split as kernel_regtile_grad.h (one packed transform per
// channel, z = x_c + i dOut_c; conj(X[k]) R[k] = Im(A[k] A[N-k]) / 2 - i (|A[k]|^2 - |A[N-k]|^2) / 4; partner bins
// through a half exchange; 8-channel tiles; S workgroups per (batch, group); deterministic finish kernel), on the
// thread layout and compile-time mixed-radix engine of kernel_regtile_mixed.h.
#pragma once
#include "kernel_regtile_grad.h"
#include "kernel_regtile_mixed.h"

namespace sfft {

constexpr int mixed_partner_stride(int RS) { return ((RS / 2) & 1) ? RS / 2 : RS / 2 + 1; }   // odd: conflict-free b32
template <int RF, int RS> constexpr int mixed_grad_image_bytes() {
  constexpr int e = mixed_image_bytes<RF, RS>(), pz = 2 * RF * kPC * mixed_partner_stride(RS) * 4;
  return e > pz ? e : pz;
}
template <int RF, int RS> constexpr int mixed_grad_lds_total() { return mixed_grad_image_bytes<RF, RS>() + (RF * RS / 2 + 1) * 8; }

template <int RF, int RS, bool IO_BF16, bool GENERAL>
__global__ void __launch_bounds__(kPC * (RF > RS ? RF : RS), (!GENERAL && (RF > RS ? RF : RS) <= 50 ? 4 : 1))   // see kernel_regtile_mixed.h
spectre_gate_grad_regtile_mixed(const GateGradArgs a) {
  static_assert(RS % 2 == 0, "the half exchange needs an even RS");
  constexpr int N = RF * RS, NZ = mixed_team<RF, RS>(), NT = mixed_threads<RF, RS>();
  constexpr int RAF = Split<RF>::RA, RBF = Split<RF>::RB;
  constexpr int ROW1 = mixed_row(RS);
  constexpr int ES = IO_BF16 ? 2 : 4;
  constexpr int PS2 = mixed_partner_stride(RS), RW2 = kPC * PS2, PLANE2 = RF * RW2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);
  float2* acc = reinterpret_cast<float2*>(smem + mixed_grad_image_bytes<RF, RS>());

  const int tid = threadIdx.x;
  const int p0 = tid & (kPC - 1);
  const int u0 = tid / kPC;

  const int wg_lin = xcd_contiguous(blockIdx.x, a.n_wg);
  const int s = wg_lin % a.S, bg = wg_lin / a.S;
  const int b = bg / a.G, g = bg - b * a.G;

  for (int k = tid; k <= N / 2; k += NT) acc[k] = make_float2(0.f, 0.f);   // ordered by E1's barriers

  for (int jt = s; jt < a.T; jt += a.S) {
    int p = p0, u = u0;
    asm volatile("" : "+v"(p), "+v"(u));            // keeps per-lane addresses out of LICM (see kernel_regtile.h)
    const bool rows = u < RS, bins = u < RF;
    float2 z[NZ];
    if (rows) {
      // buffer loads (kernel_regtile_grad.h): workgroup-uniform tile base + a 32-bit lane offset; GENERAL: rows >= N_in and the lanes of
      // a ragged last tile are the out-of-range case (0 = rfft's zero padding) — no predicates, no 64-bit address arithmetic
      const int c0 = g * a.d_g + kPC * jt;
      const int nrow = a.N_in < N ? a.N_in : N;
      const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(a.v)) + ((size_t)b * a.v_sb + c0) * ES, 0, GENERAL ? (int)((long long)nrow * a.v_sn * ES) : 0x7fffffff, kRsrcFlags);
      const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(a.dout)) + ((size_t)b * a.dout_sb + c0) * ES, 0, GENERAL ? (int)((long long)nrow * a.dout_sn * ES) : 0x7fffffff, kRsrcFlags);
      uint32_t voff = (uint32_t)(((long long)u * a.v_sn + p) * ES), doff = (uint32_t)(((long long)u * a.dout_sn + p) * ES);
      if constexpr (GENERAL) { if (kPC * jt + p >= a.d_g) { voff = 0x80000000u; doff = 0x80000000u; } }
      static_for<0, RF>([&](auto ic) {
        constexpr int q = in_order<RF>(decltype(ic)::value);
        uint32_t vo = voff, dof = doff, vs = (uint32_t)((long long)q * RS * a.v_sn * ES), ds = (uint32_t)((long long)q * RS * a.dout_sn * ES);
        if constexpr (GENERAL) { vo += vs; dof += ds; vs = 0; ds = 0; }   // the range check covers the VGPR offset only
        float x, dy;
        if constexpr (IO_BF16) {                    // the raw halves now, the shift behind every request (below)
          x = __uint_as_float((uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rv, vo, vs, 0));
          dy = __uint_as_float((uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rd, dof, ds, 0));
        } else {
          x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, vo, vs, 0));
          dy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, dof, ds, 0));
        }
        z[q] = make_float2(x, dy);
      });
      if constexpr (IO_BF16) {
        // Unpacked in a second pass: with the shift inside the loop hipcc serialises the requests at some (RF, RS) — load, s_waitcnt
        // vmcnt(0), shift, next load — one request in flight per wave (dgate at bf16 4.8 ms against 2.0 at fp32, (500,1536,768);
        // tools/dtype_sweep.py).  The same happened to the forward kernel (kernel_regtile_mixed.h).
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, RF>([&](auto ic) {
          constexpr int q = decltype(ic)::value;
          z[q] = make_float2(__uint_as_float(__float_as_uint(z[q].x) << 16), __uint_as_float(__float_as_uint(z[q].y) << 16));
        });
      }
      fft_ct<RF, false, IdentityMap, NZ>(z);
      float2 wa[RAF], wb[RBF];
      static_for<1, RAF>([&](auto jc) { constexpr int j = decltype(jc)::value; wa[j] = a.tw[u * j]; });
      static_for<1, RBF>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = a.tw[u * RAF * j]; });
      static_for<1, RF>([&](auto kc) {
        constexpr int k1 = decltype(kc)::value, ka = k1 % RAF, kb = k1 / RAF, pos = out_pos<RF>(k1);
        if constexpr (ka > 0) z[pos] = cmul(z[pos], wa[ka]);
        if constexpr (kb > 0) z[pos] = cmul(z[pos], wb[kb]);
      });
    }

    const MixedM0 m0 = mixed_m0(img, tid);
    // ---- E1 (as in kernel_regtile_mixed.h) ------------------------------------------------------------------------
    if (rows) static_for<0, RF>([&](auto kc) { constexpr int k1 = decltype(kc)::value; mixed_write_addtid<k1 * ROW1 * 4>(z[out_pos<RF>(k1)].x, m0); });
    __syncthreads();
    if (bins) static_for<0, RS>([&](auto nc) { constexpr int n2 = decltype(nc)::value; z[n2].x = img[u * ROW1 + n2 * kPC + p]; });
    __syncthreads();
    if (rows) static_for<0, RF>([&](auto kc) { constexpr int k1 = decltype(kc)::value; mixed_write_addtid<k1 * ROW1 * 4>(z[out_pos<RF>(k1)].y, m0); });
    __syncthreads();
    if (bins) static_for<0, RS>([&](auto nc) { constexpr int n2 = decltype(nc)::value; z[n2].y = img[u * ROW1 + n2 * kPC + p]; });
    __syncthreads();

    // ---- F2: bin k = u + RF*k2 at z[out_pos<RS>(k2)]; upper half (k2 >= RS/2) parked in LDS for the partners ---------
    if (bins) {
      fft_ct<RS, false, IdentityMap, NZ>(z);
      float* wre = img + u * RW2 + p * PS2;
      static_for<RS / 2, RS>([&](auto kc) {
        constexpr int k2 = decltype(kc)::value, pos = out_pos<RS>(k2);
        wre[k2 - RS / 2] = z[pos].x;
        wre[PLANE2 + k2 - RS / 2] = z[pos].y;
      });
    }
    __syncthreads();
    if (bins) {
      const int k1 = u;
      const bool k1z = (k1 == 0);
      // partner bin N - k = (RF - k1, RS - 1 - k2), or (0, RS - k2) for k1 = 0  ->  slot RS/2 - k2 - (k1 != 0)
      const float* rre = img + (k1z ? 0 : RF - k1) * RW2 + p * PS2 + (RS / 2) - (k1z ? 0 : 1);
      static_for<0, RS / 2>([&](auto kc) {
        constexpr int k2 = decltype(kc)::value, pos = out_pos<RS>(k2);
        float pr, pi;
        if constexpr (k2 == 0) {                    // k1 = 0: DC is its own partner (no slot RS/2 exists: read slot 0)
          const float* r0 = img + (k1z ? 0 : RF - k1) * RW2 + p * PS2 + (k1z ? 0 : RS / 2 - 1);
          pr = k1z ? z[pos].x : r0[0];
          pi = k1z ? z[pos].y : r0[PLANE2];
        } else {
          pr = rre[-k2];
          pi = rre[PLANE2 - k2];
        }
        const float q = z[pos].x * pi + z[pos].y * pr;                                       // Im(A A')
        const float e = (z[pos].x * z[pos].x + z[pos].y * z[pos].y) - (pr * pr + pi * pi);   // |A|^2 - |A'|^2
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
      });
      {                                             // Nyquist: k1 = 0, k2 = RS/2: Re(A) Im(A)
        constexpr int pos = out_pos<RS>(RS / 2);
        const float sr = team_sum8(k1z ? z[pos].x * z[pos].y : 0.f);
        if (p == 0 && k1z) acc[N / 2].x += sr;
      }
    }
    __syncthreads();
  }

  __syncthreads();
  float2* dst = a.part + ((size_t)bg * a.S + s) * a.F;
  for (int k = tid; k <= N / 2; k += NT) dst[k] = acc[k];
}

template <int RF, int RS>
hipError_t launch_gate_grad_mixed(const GateGradArgs& a, bool io_bf16, bool general, hipStream_t stream);

#define SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(RF_, RS_)                                                       \
  template <>                                                                                                \
  hipError_t launch_gate_grad_mixed<RF_, RS_>(const GateGradArgs& a, bool io_bf16, bool general,             \
                                              hipStream_t stream) {                                          \
    const dim3 grid(a.n_wg), block(mixed_threads<RF_, RS_>());                                               \
    const size_t lds = mixed_grad_lds_total<RF_, RS_>();                                                     \
    const int key = (io_bf16 ? 2 : 0) | (general ? 1 : 0);                                                   \
    static std::atomic<bool> lds_opt_in[16][4];                                                                      \
    auto go = [&](auto kern) -> hipError_t {                                                                 \
      int dev = 0;                                                                                           \
      (void)hipGetDevice(&dev);                                                                              \
      if (dev < 0 || dev >= 16 || !lds_opt_in[dev][key]) {                                                   \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e != hipSuccess) return e;                                                                       \
        if ((e = mixed_check_lds_layout(reinterpret_cast<const void*>(kern))) != hipSuccess) return e;       \
        if (dev >= 0 && dev < 16) lds_opt_in[dev][key] = true;                                               \
      }                                                                                                      \
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                                 \
      return hipGetLastError();                                                                              \
    };                                                                                                       \
    switch (key) {                                                                                           \
      case 0: return go(spectre_gate_grad_regtile_mixed<RF_, RS_, false, false>);                            \
      case 1: return go(spectre_gate_grad_regtile_mixed<RF_, RS_, false, true>);                             \
      case 2: return go(spectre_gate_grad_regtile_mixed<RF_, RS_, true, false>);                             \
      default: return go(spectre_gate_grad_regtile_mixed<RF_, RS_, true, true>);                             \
    }                                                                                                        \
  }

}  // namespace sfft
