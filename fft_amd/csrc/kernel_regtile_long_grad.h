// kernel_regtile_long_grad.h — register-resident gate gradient for n_fft = RF x 128 (6144, 8192).
//
// Mathematics and work split of kernel_regtile_grad.h (one packed transform per channel, z = x_c + i dOut_c;
// conj(X[k]) R[k] = Im(A[k] A[N-k]) / 2 - i (|A[k]|^2 - |A[N-k]|^2) / 4; S workgroups per (batch, group); deterministic
// finish kernel) on the thread layout of kernel_regtile_long.h: 4-channel tiles (one channel per lane p), the 128-point
// transform owned by a lane pair (h = lane & 1).  After the pair butterfly lane h = 0 holds the half spectrum
// k = k1 + RF k2' (k2' < 64) and lane h = 1 the upper half, whose values it parks in LDS ([k1][p][k2' - 64]) for the
// partners:  N - k = (RF - k1, 63 - k2') in that layout, or (0, 64 - k2') for k1 = 0.  The four channels of a tile are
// summed across lanes (DPP for lane ^ 2, a bpermute for lane ^ 4), and lane p keeps the running sums of the bins with
// k2' % 4 == p in 16 complex registers for all the tiles of the workgroup (the LDS has no room left for accumulators).
#pragma once
#include "kernel_regtile_grad.h"
#include "kernel_regtile_long.h"

namespace sfft {

constexpr int kLongPartnerStride = 65;                                    // slots per (k1, p) column, odd
constexpr int long_grad_image_bytes(int RF) {
  return long_image_bytes(RF) > 2 * RF * kLongPC * kLongPartnerStride * 4 ? long_image_bytes(RF) : 2 * RF * kLongPC * kLongPartnerStride * 4;
}

template <int RF, bool IO_BF16, bool GENERAL>
__global__ void __launch_bounds__(512) spectre_gate_grad_regtile_long(const GateGradArgs a) {
  constexpr int N = RF * kLongRS, RS = kLongRS, PC = kLongPC;
  constexpr int RAF = Split<RF>::RA, RBF = Split<RF>::RB;
  constexpr int ES = IO_BF16 ? 2 : 4;
  constexpr int PS = kLongPartnerStride, RW = PC * PS, PLANE = RF * RW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x;
  const int wg_lin = xcd_contiguous(blockIdx.x, a.n_wg);
  const int s = wg_lin % a.S, bg = wg_lin / a.S;
  const int b = bg / a.G, g = bg - b * a.G;

  const int pb0 = (tid >> 1) & 3;
  float msk[4];                                       // lane p keeps the bins with k2' % 4 == p
  static_for<0, 4>([&](auto qc) { msk[decltype(qc)::value] = (pb0 == decltype(qc)::value) ? 1.f : 0.f; });

  float2 acc[16];
  static_for<0, 16>([&](auto ic) { acc[decltype(ic)::value] = make_float2(0.f, 0.f); });
  float acc_nyq = 0.f;

  for (int jt = s; jt < a.T; jt += a.S) {
    // opaque per-iteration copy of the thread index: otherwise every per-lane address is hoisted out of the tile loop
    // and kept in registers next to the 64 complex values and the 16 accumulators (see kernel_regtile.h)
    int tix = tid;
    asm volatile("" : "+v"(tix));
    // row role: tid = p + 4 n2;  bin role: tid = h + 2 p + 8 k1
    const int pa = tix & 3, n2 = tix >> 2;
    const int h = tix & 1, pb = (tix >> 1) & 3, k1 = tix >> 3;
    const bool bins = k1 < RF;
    const int k1c = bins ? k1 : 0;
    const bool k1z = (k1c == 0);
    const int peer4 = ((tix & 63) ^ 4) << 2;          // ds_bpermute address of lane ^ 4
    const int cl = PC * jt + pa;                     // channel inside the group
    bool cok = true;
    if constexpr (GENERAL) cok = cl < a.d_g;

    float2 z[64];
    {
      // buffer loads (kernel_regtile_grad.h): workgroup-uniform tile base + a 32-bit lane offset; GENERAL: rows >= N_in and the lanes of
      // a ragged last tile are the out-of-range case (0 = rfft's zero padding) — no predicates, no 64-bit address arithmetic
      const int c0 = g * a.d_g + PC * jt;
      const int nrow = a.N_in < N ? a.N_in : N;
      const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(a.v)) + ((size_t)b * a.v_sb + c0) * ES, 0, GENERAL ? (int)((long long)nrow * a.v_sn * ES) : 0x7fffffff, kRsrcFlags);
      const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(a.dout)) + ((size_t)b * a.dout_sb + c0) * ES, 0, GENERAL ? (int)((long long)nrow * a.dout_sn * ES) : 0x7fffffff, kRsrcFlags);
      uint32_t voff = (uint32_t)(((long long)n2 * a.v_sn + pa) * ES), doff = (uint32_t)(((long long)n2 * a.dout_sn + pa) * ES);
      if constexpr (GENERAL) { if (!cok) { voff = 0x80000000u; doff = 0x80000000u; } }
      static_for<0, RF>([&](auto ic) {
        constexpr int q = in_order<RF>(decltype(ic)::value);
        uint32_t vo = voff, dof = doff, vs = (uint32_t)((long long)q * RS * a.v_sn * ES), ds = (uint32_t)((long long)q * RS * a.dout_sn * ES);
        if constexpr (GENERAL) { vo += vs; dof += ds; vs = 0; ds = 0; }   // the range check covers the VGPR offset only
        float x, dy;
        if constexpr (IO_BF16) {
          x = __uint_as_float((uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rv, vo, vs, 0) << 16);
          dy = __uint_as_float((uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rd, dof, ds, 0) << 16);
        } else {
          x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, vo, vs, 0));
          dy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, dof, ds, 0));
        }
        z[q] = make_float2(x, dy);
      });
      fft_ct<RF, false, IdentityMap, 64>(z);
      float2 wa[RAF], wb[RBF];
      static_for<1, RAF>([&](auto jc) { constexpr int j = decltype(jc)::value; wa[j] = a.tw[n2 * j]; });
      static_for<1, RBF>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = a.tw[n2 * RAF * j]; });
      static_for<1, RF>([&](auto kc) {
        constexpr int kk = decltype(kc)::value, ka = kk % RAF, kb = kk / RAF, pos = out_pos<RF>(kk);
        if constexpr (ka > 0) z[pos] = cmul(z[pos], wa[ka]);
        if constexpr (kb > 0) z[pos] = cmul(z[pos], wb[kb]);
      });
    }

    // ---- E1 (as in kernel_regtile_long.h) -------------------------------------------------------------------------
    {
      float* wbase = img + n2 * PC + pa;
      const float* rbase = img + k1c * kLongRow1 + h * PC + pb;
      static_for<0, RF>([&](auto kc) { constexpr int kk = decltype(kc)::value; wbase[kk * kLongRow1] = z[out_pos<RF>(kk)].x; });
      __syncthreads();
      static_for<0, 64>([&](auto mc) { constexpr int m = decltype(mc)::value; z[m].x = rbase[m * 2 * PC]; });
      __syncthreads();
      static_for<0, RF>([&](auto kc) { constexpr int kk = decltype(kc)::value; wbase[kk * kLongRow1] = z[out_pos<RF>(kk)].y; });
      __syncthreads();
      static_for<0, 64>([&](auto mc) { constexpr int m = decltype(mc)::value; z[m].y = rbase[m * 2 * PC]; });
      __syncthreads();
    }

    // ---- F2 + pair butterfly: lane h holds A[k1 + RF (k2' + 64 h)] at position 8 ka + kb, k2' = ka + 8 kb ---------------
    fftA<8, 8, false>(z);
    {
      const float sgn = h ? -1.f : 1.f;
      static_for<0, 64>([&](auto jc) {
        constexpr int j = decltype(jc)::value, k2p = (j / 8) + 8 * (j % 8);
        constexpr float wc = (float)TwTab<128>::c[k2p], ws = (float)TwTab<128>::s[k2p];
        float2 own = z[j];
        if constexpr (k2p > 0) {
          const float cc = h ? wc : 1.f, ss = h ? ws : 0.f;
          own = make_float2(own.x * cc + own.y * ss, own.y * cc - own.x * ss);
        }
        z[j] = make_float2(fmaf(sgn, own.x, dpp_swap1(own.x)), fmaf(sgn, own.y, dpp_swap1(own.y)));
      });
    }

    // ---- upper half (h = 1) -> LDS [k1][p][k2']; lower half reads its partners ------------------------------------------
    if (bins && h) {
      float* wre = img + k1 * RW + pb * PS;
      static_for<0, 64>([&](auto jc) {
        constexpr int j = decltype(jc)::value, k2p = (j / 8) + 8 * (j % 8);
        wre[k2p] = z[j].x;
        wre[PLANE + k2p] = z[j].y;
      });
    }
    __syncthreads();
    {
      // partner of (k1, k2') with k2' < 64: row RF - k1 (0 for k1 = 0), slot 63 - k2' (64 - k2' for k1 = 0; k2' = 0 is DC, its own partner)
      const float* rre = img + (k1z ? 0 : RF - k1c) * RW + pb * PS + (k1z ? 64 : 63);
      int roff = 0;                                   // opaque per group of 8: keeps the 64 partner addresses from being formed up front
      static_for<0, 64>([&](auto jc) {
        constexpr int j = decltype(jc)::value, k2p = (j / 8) + 8 * (j % 8);
        if constexpr (j % 8 == 0 && j > 0) {
          asm volatile("" : "+v"(roff));
          __builtin_amdgcn_sched_barrier(0);           // partner reads 8 deep (register budget)
        }
        float pr, pi;
        if constexpr (k2p == 0) {
          const float* r0 = img + (k1z ? 0 : RF - k1c) * RW + pb * PS + (k1z ? 0 : 63);
          pr = k1z ? z[j].x : r0[0];
          pi = k1z ? z[j].y : r0[PLANE];
        } else {
          pr = rre[roff - k2p];
          pi = rre[roff + PLANE - k2p];
        }
        float sr = 0.5f * (z[j].x * pi + z[j].y * pr);                                         // Im(A A') / 2
        float si = -0.25f * ((z[j].x * z[j].x + z[j].y * z[j].y) - (pr * pr + pi * pi));       // -(|A|^2 - |A'|^2) / 4
        // sum over the four channel lanes (lane ^ 2, lane ^ 4)
        sr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), 0x4E, 0xF, 0xF, false));
        si += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), 0x4E, 0xF, 0xF, false));
        sr += __int_as_float(__builtin_amdgcn_ds_bpermute(peer4, __float_as_int(sr)));
        si += __int_as_float(__builtin_amdgcn_ds_bpermute(peer4, __float_as_int(si)));
        acc[k2p >> 2].x = fmaf(msk[k2p & 3], sr, acc[k2p >> 2].x);
        acc[k2p >> 2].y = fmaf(msk[k2p & 3], si, acc[k2p >> 2].y);
      });
      {   // Nyquist: k1 = 0, h = 1, k2' = 0: Re(A) Im(A)
        float sn = (h && k1z) ? z[0].x * z[0].y : 0.f;
        sn += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sn), 0x4E, 0xF, 0xF, false));
        sn += __int_as_float(__builtin_amdgcn_ds_bpermute(peer4, __float_as_int(sn)));
        acc_nyq += sn;
      }
    }
    __syncthreads();                                  // partner image read: the next tile's E1 may overwrite it
  }

  float2* dst = a.part + ((size_t)bg * a.S + s) * a.F;
  const int h = tid & 1, pb = pb0, k1 = tid >> 3;
  if (k1 < RF && !h) {
    static_for<0, 16>([&](auto qc) { constexpr int q = decltype(qc)::value; dst[k1 + RF * (4 * q + pb)] = acc[q]; });
  }
  if (h && k1 == 0 && pb == 0) dst[N / 2] = make_float2(acc_nyq, 0.f);
}

template <int RF>
inline hipError_t launch_gate_grad_long(const GateGradArgs& a, bool io_bf16, bool general, hipStream_t stream) {
  const dim3 grid(a.n_wg), block(512);
  const size_t lds = long_grad_image_bytes(RF);
  const int key = (io_bf16 ? 2 : 0) | (general ? 1 : 0);
  static std::atomic<bool> lds_opt_in[16][4];
  auto go = [&](auto kern) -> hipError_t {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 16 || !lds_opt_in[dev][key]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 16) lds_opt_in[dev][key] = true;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    return hipGetLastError();
  };
  switch (key) {
    case 0: return go(spectre_gate_grad_regtile_long<RF, false, false>);
    case 1: return go(spectre_gate_grad_regtile_long<RF, false, true>);
    case 2: return go(spectre_gate_grad_regtile_long<RF, true, false>);
    default: return go(spectre_gate_grad_regtile_long<RF, true, true>);
  }
}

}  // namespace sfft
