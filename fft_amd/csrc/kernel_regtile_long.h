// kernel_regtile_long.h — register-resident spectral mix for n_fft = RF x 128, RF in {40, 48, 56, 64}
// (5120, 6144, 7168, 8192; the library builds 6144 and 8192) on gfx950.
//
// Same plan as kernel_regtile.h (which see: /root/reference/spectre.py:506, :542-553 in one kernel, two real channels
// per complex sequence, LDS only as the transposition buffer), with two changes forced by the length:
//   * a tile is 8 channels (4 packed sequences, 32-byte fp32 row segments) x n_fft rows = 256 KiB at 8192 — the same half
//     register file as the 4096-row tile; four neighbouring tiles share every 128-byte line and are adjacent in the
//     XCD-contiguous order, so the line is fetched once (the gate-gradient kernels use the same trick);
//   * the second transform has 128 points but a thread holds 64 values: a lane PAIR (h = lane & 1) owns one (column,
//     k1), each lane transforming the 64 samples n2 = 2m + h, and the radix-2 step across the pair goes through a DPP
//     lane swap:   X[k2']      = E_0[k2'] + W_128^k2' E_1[k2']      (kept by h = 0)
//                  X[k2' + 64] = E_0[k2'] - W_128^k2' E_1[k2']      (kept by h = 1)
//     and the inverse runs the same butterfly as decimation in frequency before the two 64-point inverse transforms.
//
//   thread roles   rows : tid = p + 4 n2          (n2 < 128: loads, F1 over n1 < RF, twiddles, I2, stores)
//                  bins : tid = h + 2 p + 8 k1     (k1 < RF, h < 2: F2, pair butterfly, gate, inverse)
//   n = n2 + 128 n1,  k = k1 + RF k2,  k2 = k2' + 64 h.  The RF-point transforms use the mixed-radix engine (RF = 64: 8 x 8).
#pragma once
#include "kernel_regtile.h"
#include "fft_regs_mixed.h"   // TwTab<128>

namespace sfft {

constexpr int kLongPC = 4;                     // packed sequences per tile (8 channels)
constexpr int kLongRS = 128;
constexpr int kLongRow1 = kLongRS * kLongPC + 8;     // E1 image [k1][n2][p]: row stride 520 floats (8-float pad: banks)
constexpr int long_row2(int RF) { return RF * kLongPC + 4; }   // E2 image [n2][sigma(k1)][p]: row stride = 4 (mod 32) floats
constexpr int long_image_bytes(int RF) { return (RF * kLongRow1 > kLongRS * long_row2(RF) ? RF * kLongRow1 : kLongRS * long_row2(RF)) * 4; }

// slot of bin class k1 inside an E2 row: the two low bits move up by one so that the 32 lanes of a write group
// (h', p, k1 & 3) fall into 32 different banks; the reader's k1 is a compile-time constant, so this costs nothing
__host__ __device__ constexpr int long_sigma(int k1) { return (k1 & ~7) | ((k1 & 3) << 1) | ((k1 >> 2) & 1); }

__device__ __forceinline__ float dpp_swap1(float v) {      // value of the neighbouring lane (lane ^ 1)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}

// MODE 0: N_in >= n_fft and D % 8 == 0 (no predicates); 1: row / channel predicates; 2: predicates + memory_fft
template <int RF, bool IN_BF16, bool OUT_BF16, int MODE>
__global__ void __launch_bounds__(512) spectre_mix_regtile_long(const RegtileArgs a) {
  static_assert(RF % 8 == 0 && RF <= 64, "RF: multiple of 8 up to 64");
  constexpr bool GENERAL = MODE != 0, WITH_MEM = MODE == 2;
  constexpr int N = RF * kLongRS, RS = kLongRS, PC = kLongPC, kLongRow2 = long_row2(RF);
  constexpr int RAF = Split<RF>::RA, RBF = Split<RF>::RB;
  constexpr int ES_IN = IN_BF16 ? 2 : 4, ES_OUT = OUT_BF16 ? 2 : 4;
  constexpr float inv_n = 1.0f / (float)N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x;
  const int tile = xcd_contiguous(blockIdx.x, a.n_wg);
  if (tile >= a.n_tiles) return;                    // workgroup-uniform
  const int b = tile / a.tiles_per_row;
  const int ct = tile - b * a.tiles_per_row;

  // row role
  const int pa = tid & (PC - 1), n2 = tid >> 2;
  const int ca = ct * (2 * PC) + 2 * pa;
  bool ca_ok = true;
  if constexpr (GENERAL) ca_ok = ca < a.D;
  // bin role
  const int h = tid & 1, pb = (tid >> 1) & (PC - 1), k1 = tid >> 3;
  const bool bins = k1 < RF;                         // (RF < 64: the last threads have no bin class)
  const int cb_raw = ct * (2 * PC) + 2 * pb;
  const int cb = cb_raw < a.D ? cb_raw : 0;

  auto load_twiddle_bases = [&](float2 (&wa)[RAF], float2 (&wb)[RBF]) {     // W_N^(n2 ka), W_N^(n2 RAF kb)
    static_for<1, RAF>([&](auto jc) { constexpr int j = decltype(jc)::value; wa[j] = a.tw[n2 * j]; });
    static_for<1, RBF>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = a.tw[n2 * RAF * j]; });
  };

  float2 z[64];

  // ---- load rows n2 + 128 q, F1 over q, W_N^(n2 k1) ----------------------------------------------------------------
  {
    const char* vb = reinterpret_cast<const char*>(a.v) + ((size_t)b * a.v_sb + (size_t)ct * (2 * PC)) * ES_IN;
    const uint32_t voff = (uint32_t)(((long long)n2 * a.v_sn + 2 * pa) * ES_IN);
    // general modes: rows >= N_in and the lanes of a ragged last tile are the out-of-range case of the buffer instructions
    // (kernel_regtile.h): loads return rfft's zero padding, stores are dropped — no predicates, no pointer selects
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs_in;
    [[maybe_unused]] uint32_t voff_c = voff;
    if constexpr (GENERAL) {
      const int nrow = a.N_in < N ? a.N_in : N;
      rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vb), 0, (int)((long long)nrow * a.v_sn * ES_IN), kRsrcFlags);
      voff_c = ca_ok ? voff : 0x80000000u;
    }
    if constexpr (IN_BF16) {
      // bf16 rows: every request first, the unpacking afterwards (kernel_regtile_mixed.h: written as one loop hipcc serialises the
      // requests — load, s_waitcnt vmcnt(0), unpack, next load; tools/serial_load_scan.py found 48 / 64 such pairs in the general modes)
      static_for<0, RF>([&](auto ic) {
        constexpr int q = in_order<RF>(decltype(ic)::value);
        uint32_t wv;
        if constexpr (GENERAL) wv = __builtin_amdgcn_raw_buffer_load_b32(rs_in, voff_c + (uint32_t)((long long)q * RS * a.v_sn * ES_IN), 0, 0);
        else wv = *reinterpret_cast<const uint32_t*>(vb + (size_t)q * RS * a.v_sn * ES_IN + voff);
        z[q].x = __uint_as_float(wv);
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, RF>([&](auto ic) {
        constexpr int q = in_order<RF>(decltype(ic)::value);
        const uint32_t wv = __float_as_uint(z[q].x);
        z[q] = make_float2(__uint_as_float(wv << 16), __uint_as_float(wv & 0xffff0000u));
      });
    } else
    static_for<0, RF>([&](auto ic) {
      constexpr int q = in_order<RF>(decltype(ic)::value);
      if constexpr (GENERAL) {
        const uint32_t off = voff_c + (uint32_t)((long long)q * RS * a.v_sn * ES_IN);
        if constexpr (IN_BF16) {
          const uint32_t wv = __builtin_amdgcn_raw_buffer_load_b32(rs_in, off, 0, 0);
          z[q] = make_float2(__uint_as_float(wv << 16), __uint_as_float(wv & 0xffff0000u));
        } else {
          const rt_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs_in, off, 0, 0);
          z[q] = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
        }
      } else {
        const char* ptr = vb + (size_t)q * RS * a.v_sn * ES_IN + voff;
        if constexpr (IN_BF16) {
          const uint32_t wv = *reinterpret_cast<const uint32_t*>(ptr);
          z[q] = make_float2(__uint_as_float(wv << 16), __uint_as_float(wv & 0xffff0000u));
        } else {
          z[q] = *reinterpret_cast<const float2*>(ptr);
        }
      }
    });
    fft_ct<RF, false, IdentityMap, 64>(z);             // k1 at position out_pos<RF>(k1)
    float2 wa[RAF], wb[RBF];
    load_twiddle_bases(wa, wb);
    static_for<1, RF>([&](auto kc) {
      constexpr int kk = decltype(kc)::value, ka = kk % RAF, kb = kk / RAF, pos = out_pos<RF>(kk);
      if constexpr (ka > 0) z[pos] = cmul(z[pos], wa[ka]);
      if constexpr (kb > 0) z[pos] = cmul(z[pos], wb[kb]);
    });
  }

  // ---- E1: (n2, k1) -> bin thread (k1, h = n2 & 1), slot m = n2 >> 1; one float plane at a time --------------------
  {
    float* wbase = img + n2 * PC + pa;
    const float* rbase = img + (bins ? k1 : 0) * kLongRow1 + h * PC + pb;
    static_for<0, RF>([&](auto kc) { constexpr int kk = decltype(kc)::value; wbase[kk * kLongRow1] = z[out_pos<RF>(kk)].x; });
    __syncthreads();
    static_for<0, 64>([&](auto mc) { constexpr int m = decltype(mc)::value; z[m].x = rbase[m * 2 * PC]; });   // real parts replaced first,
    __syncthreads();                                                                                            // imaginary parts still in place
    static_for<0, RF>([&](auto kc) { constexpr int kk = decltype(kc)::value; wbase[kk * kLongRow1] = z[out_pos<RF>(kk)].y; });
    __syncthreads();
    static_for<0, 64>([&](auto mc) { constexpr int m = decltype(mc)::value; z[m].y = rbase[m * 2 * PC]; });
    __syncthreads();
  }

  // ---- F2 (64 points over m), radix-2 across the lane pair, gate (+ memory), inverse radix-2, I1 ---------------------
  {
    fftA<8, 8, false>(z);                              // E_h[k2'], k2' = ka + 8 kb at position 8 ka + kb
    const float sgn = h ? -1.f : 1.f;
    const int grp = cb / a.d_g;
    const float2* gp = a.gate + (size_t)b * a.G * a.F;   // workgroup-uniform base + 32-bit lane offset
    const int kc1 = bins ? k1 : 0;                    // threads without a bin class compute on garbage and never store it
    // bin k = k1 + RF (k2' + 64 h): h = 1 is the upper half -> conj(g[N - k]), N - k = RF (64 - k2') - k1: base + step * k2'.
    // The opaque copies below make every group of 8 bins compute its indices when it needs them: left alone, instruction
    // selection evaluates all 64 up front and keeps them in 64 registers.
    int mbase = h ? RF * 64 - kc1 : kc1;
    int gstep = h ? -RF : RF;
    int gbase = grp * a.F + mbase;
    static_for<0, 64>([&](auto jc) {
      constexpr int j = decltype(jc)::value, k2p = (j / 8) + 8 * (j % 8);
      if constexpr (j % 8 == 0 && j > 0) {
        asm volatile("" : "+v"(gbase), "+v"(gstep), "+v"(mbase));
        __builtin_amdgcn_sched_barrier(0);             // keep the gate loads 8 deep (register budget)
      }
      // h = 1 first applies W_128^k2' to its half; then X = partner + sgn * own  (E_0 + W E_1 | E_0 - W E_1)
      constexpr float wc = (float)TwTab<128>::c[k2p], ws = (float)TwTab<128>::s[k2p];
      float2 own = z[j];
      if constexpr (k2p > 0) {
        const float c = h ? wc : 1.f, s = h ? ws : 0.f;
        own = make_float2(own.x * c + own.y * s, own.y * c - own.x * s);
      }
      float2 x = make_float2(fmaf(sgn, own.x, dpp_swap1(own.x)), fmaf(sgn, own.y, dpp_swap1(own.y)));
      float2 g = gp[gbase + gstep * k2p];
      if (a.conj_gate) g.y = -g.y;
      const bool edge = (k2p == 0) && (kc1 == 0);      // DC (h = 0) and Nyquist (h = 1): irfft ignores Im
      if (edge) g.y = 0.f;
      if (h) g.y = -g.y;
      g.x *= inv_n; g.y *= inv_n;
      float2 y = cmul(x, g);
      if constexpr (WITH_MEM) {                         // spectre.py:548-549
        const float4 m = *reinterpret_cast<const float4*>(a.mem + ((size_t)(mbase + gstep * k2p) * a.D + cb) * 2);
        float2 add;
        if (edge)   add = make_float2(m.x, m.z);
        else if (h) add = make_float2(m.x + m.w, m.z - m.y);
        else        add = make_float2(m.x - m.w, m.y + m.z);
        y.x += add.x * inv_n; y.y += add.y * inv_n;
      }
      // inverse, decimation in frequency: U_0 = Y_lo + Y_hi (h' = 0), U_1 = (Y_lo - Y_hi) conj(W_128^k2') (h' = 1)
      float2 u = make_float2(fmaf(sgn, y.x, dpp_swap1(y.x)), fmaf(sgn, y.y, dpp_swap1(y.y)));
      if constexpr (k2p > 0) {
        const float c = h ? wc : 1.f, s = h ? ws : 0.f;
        u = make_float2(u.x * c - u.y * s, u.y * c + u.x * s);
      }
      z[j] = u;
    });
    static_for<0, 8>([&](auto kac) { fftB_stage1_group<8, 8, true, decltype(kac)::value>(z); });
    fftB_stage2<8, 8, true>(z);                        // C[n2 = 2 m + h] at position m
  }

  // ---- E2: (k1, n2 = 2m + h) -> row thread n2, slot sigma(k1) -----------------------------------------------------------
  {
    float* wbase = img + h * kLongRow2 + long_sigma(k1) * PC + pb;
    const float* rbase = img + n2 * kLongRow2 + pa;
    if (bins) static_for<0, 64>([&](auto mc) { constexpr int m = decltype(mc)::value; wbase[m * 2 * kLongRow2] = z[m].x; });
    __syncthreads();
    static_for<0, RF>([&](auto kc) { constexpr int kk = decltype(kc)::value; z[kk].x = rbase[long_sigma(kk) * PC]; });
    __syncthreads();
    if (bins) static_for<0, 64>([&](auto mc) { constexpr int m = decltype(mc)::value; wbase[m * 2 * kLongRow2] = z[m].y; });
    __syncthreads();
    static_for<0, RF>([&](auto kc) { constexpr int kk = decltype(kc)::value; z[kk].y = rbase[long_sigma(kk) * PC]; });
  }

  // ---- conj twiddle, I2 over k1, store rows n2 + 128 n1 ----------------------------------------------------------------
  {
    float2 wa[RAF], wb[RBF];
    load_twiddle_bases(wa, wb);
    static_for<1, RF>([&](auto jc) {
      constexpr int j = decltype(jc)::value, ja = j % RAF, jb = j / RAF;   // position j carries k1 = j = ja + RAF jb
      if constexpr (ja > 0) z[j] = cmulc(z[j], wa[ja]);
      if constexpr (jb > 0) z[j] = cmulc(z[j], wb[jb]);
    });
    fft_ct<RF, true, IdentityMap, 64>(z);              // n1 at position out_pos<RF>(n1)
    char* ob = reinterpret_cast<char*>(a.out) + ((size_t)b * a.out_sb + (size_t)ct * (2 * PC)) * ES_OUT;
    const uint32_t ooff = (uint32_t)(((long long)n2 * a.out_sn + 2 * pa) * ES_OUT);
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs_out;
    [[maybe_unused]] uint32_t ooff_c = ooff;
    if constexpr (GENERAL) {
      const int nrow = a.N_in < N ? a.N_in : N;              // spectre.py:553 keeps rows < min(N, n_fft)
      rs_out = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)((long long)nrow * a.out_sn * ES_OUT), kRsrcFlags);
      ooff_c = ca_ok ? ooff : 0x80000000u;
    }
    static_for<0, RF>([&](auto nc) {
      constexpr int n1 = decltype(nc)::value, j = out_pos<RF>(n1);
      if constexpr (GENERAL) {
        const uint32_t off = ooff_c + (uint32_t)((long long)n1 * RS * a.out_sn * ES_OUT);
        if constexpr (OUT_BF16) {
          __builtin_amdgcn_raw_buffer_store_b32(f32x2_to_bf16x2_rne(z[j].x, z[j].y), rs_out, off, 0, 0);
        } else {
          rt_u32x2 t;
          t.x = __float_as_uint(z[j].x); t.y = __float_as_uint(z[j].y);
          __builtin_amdgcn_raw_buffer_store_b64(t, rs_out, off, 0, 0);
        }
      } else {
        char* ptr = ob + (size_t)n1 * RS * a.out_sn * ES_OUT + ooff;
        if constexpr (OUT_BF16) *reinterpret_cast<uint32_t*>(ptr) = f32x2_to_bf16x2_rne(z[j].x, z[j].y);
        else *reinterpret_cast<float2*>(ptr) = z[j];
      }
    });
  }
}

template <int RF>
inline hipError_t launch_regtile_long(const RegtileArgs& a, bool in_bf16, bool out_bf16, int mode, hipStream_t stream) {
  const dim3 grid(a.n_wg), block(512);
  const size_t lds = long_image_bytes(RF);
  const int key = (in_bf16 ? 8 : 0) | (out_bf16 ? 4 : 0) | mode;
  static std::atomic<bool> lds_opt_in[16][16];
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
    case 0: return go(spectre_mix_regtile_long<RF, false, false, 0>);
    case 1: return go(spectre_mix_regtile_long<RF, false, false, 1>);
    case 2: return go(spectre_mix_regtile_long<RF, false, false, 2>);
    case 4: return go(spectre_mix_regtile_long<RF, false, true, 0>);
    case 8: return go(spectre_mix_regtile_long<RF, true, false, 0>);
    case 12: return go(spectre_mix_regtile_long<RF, true, true, 0>);
    case 13: return go(spectre_mix_regtile_long<RF, true, true, 1>);
    case 14: return go(spectre_mix_regtile_long<RF, true, true, 2>);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sfft
