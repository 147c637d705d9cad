// kernel_regtile_quad.h — register-resident spectral mix for n_fft = RF x 256, RF in {40, 48, 56, 64}
// (10240, 12288, 14336, 16384; the library builds 12288 and 16384) on gfx950.
//
// kernel_regtile_long.h taken one step further: a tile is 4 channels (2 packed sequences, 16-byte fp32 row segments; the
// eight tiles of a 128-byte line are neighbours in the XCD-contiguous order) x n_fft rows — again 256 KiB of registers at
// 16384 — and the 256-point second transform belongs to a lane QUAD (h = lane & 3): lane h transforms the 64 samples
// n2 = 4m + h, and the radix-4 step across the quad is two DPP butterflies (lane ^ 2, then lane ^ 1):
//     X[k2' + 64 r] = sum_h  W_256^(h k2')  (-i)^(h r)  E_h[k2'],      lane h keeps r = bitrev2(h)
// with the inverse as the same two butterflies in decimation-in-frequency order before the four 64-point inverses.
//
//   thread roles   rows : tid = p + 2 n2          (n2 < 256)
//                  bins : tid = h + 4 p + 8 k1     (k1 < RF)
//   n = n2 + 256 n1,  k = k1 + RF k2,  k2 = k2' + 64 r.
#pragma once
#include "kernel_regtile_long.h"

namespace sfft {

constexpr int kQuadPC = 2, kQuadRS = 256;
constexpr int kQuadRow1 = kQuadRS * kQuadPC + 8;          // E1 image [k1][n2][p], row stride 520 floats
constexpr int kQuadCol2 = 65;                             // E2 image [n2][p][sigma(k1)]: 64 slots + 1 (odd column stride)
constexpr int kQuadRow2 = kQuadPC * kQuadCol2;            // 130 = 2 (mod 32)
constexpr int quad_image_bytes(int RF) { return (RF * kQuadRow1 > kQuadRS * kQuadRow2 ? RF * kQuadRow1 : kQuadRS * kQuadRow2) * 4; }
// slot of bin class k1 = (e:2, c:2, b:2) inside an E2 column: b goes to bits 3-4 so that the 32 lanes of a write group
// (h, p, b) fall into 32 different banks (2h + p + 8b + const); the reader's k1 is a compile-time constant
__host__ __device__ constexpr int quad_sigma(int k1) { return ((k1 & 3) << 3) | ((k1 >> 2) & 3) | (((k1 >> 4) & 1) << 2) | (((k1 >> 5) & 1) << 5); }

// Materialise 8 complex registers at this point of the program (empty asm with 16 in/out operands): keeps the
// instruction selector from evaluating a butterfly network lazily, output by output, with its partial sums spread over
// the whole register file
template <int BASE, int NTOT>
__device__ __forceinline__ void pin8(float2 (&z)[NTOT]) {
  asm volatile("" : "+v"(z[BASE].x), "+v"(z[BASE].y), "+v"(z[BASE + 1].x), "+v"(z[BASE + 1].y), "+v"(z[BASE + 2].x), "+v"(z[BASE + 2].y),
               "+v"(z[BASE + 3].x), "+v"(z[BASE + 3].y), "+v"(z[BASE + 4].x), "+v"(z[BASE + 4].y), "+v"(z[BASE + 5].x), "+v"(z[BASE + 5].y),
               "+v"(z[BASE + 6].x), "+v"(z[BASE + 6].y), "+v"(z[BASE + 7].x), "+v"(z[BASE + 7].y));
}

__device__ __forceinline__ float dpp_swap2(float v) {      // value of lane ^ 2
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
}

template <int RF, bool IN_BF16, bool OUT_BF16, int MODE>
__global__ void __launch_bounds__(512) spectre_mix_regtile_quad(const RegtileArgs a) {
  static_assert(RF % 8 == 0 && RF <= 64, "RF: multiple of 8 up to 64");
  constexpr bool GENERAL = MODE != 0, WITH_MEM = MODE == 2;
  constexpr int N = RF * kQuadRS, RS = kQuadRS, PC = kQuadPC;
  constexpr int RAF = Split<RF>::RA, RBF = Split<RF>::RB;
  constexpr int ES_IN = IN_BF16 ? 2 : 4, ES_OUT = OUT_BF16 ? 2 : 4;
  constexpr float inv_n = 1.0f / (float)N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x;
  const int tile = xcd_contiguous(blockIdx.x, a.n_wg);
  if (tile >= a.n_tiles) return;
  const int b = tile / a.tiles_per_row;
  const int ct = tile - b * a.tiles_per_row;

  const int pa = tid & 1, n2 = tid >> 1;                          // row role
  const int ca = ct * (2 * PC) + 2 * pa;
  bool ca_ok = true;
  if constexpr (GENERAL) ca_ok = ca < a.D;
  const int h = tid & 3, pb = (tid >> 2) & 1, k1 = tid >> 3;      // bin role
  const bool bins = k1 < RF;
  const int k1c = bins ? k1 : 0;
  const int cb_raw = ct * (2 * PC) + 2 * pb;
  const int cb = cb_raw < a.D ? cb_raw : 0;
  const int r = ((h & 1) << 1) | (h >> 1);                        // this lane keeps k2 = k2' + 64 r

  auto load_twiddle_bases = [&](float2 (&wa)[RAF], float2 (&wb)[RBF]) {
    static_for<1, RAF>([&](auto jc) { constexpr int j = decltype(jc)::value; wa[j] = a.tw[n2 * j]; });
    static_for<1, RBF>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = a.tw[n2 * RAF * j]; });
  };

  float2 z[64];
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
    fft_ct<RF, false, IdentityMap, 64>(z);
    float2 wa[RAF], wb[RBF];
    load_twiddle_bases(wa, wb);
    static_for<1, RF>([&](auto kc) {
      constexpr int kk = decltype(kc)::value, ka = kk % RAF, kb = kk / RAF, pos = out_pos<RF>(kk);
      if constexpr (ka > 0) z[pos] = cmul(z[pos], wa[ka]);
      if constexpr (kb > 0) z[pos] = cmul(z[pos], wb[kb]);
    });
  }

  // ---- E1: (n2, k1) -> bin thread (k1, h = n2 & 3), slot m = n2 >> 2 -------------------------------------------------
  {
    float* wbase = img + n2 * PC + pa;
    const float* rbase = img + k1c * kQuadRow1 + h * PC + pb;
    static_for<0, RF>([&](auto kc) { constexpr int kk = decltype(kc)::value; wbase[kk * kQuadRow1] = z[out_pos<RF>(kk)].x; });
    __syncthreads();
    static_for<0, 64>([&](auto mc) { constexpr int m = decltype(mc)::value; z[m].x = rbase[m * 4 * PC]; });
    __syncthreads();
    static_for<0, RF>([&](auto kc) { constexpr int kk = decltype(kc)::value; wbase[kk * kQuadRow1] = z[out_pos<RF>(kk)].y; });
    __syncthreads();
    static_for<0, 64>([&](auto mc) { constexpr int m = decltype(mc)::value; z[m].y = rbase[m * 4 * PC]; });
    __syncthreads();
  }

  // ---- F2 (64 points over m), radix-4 across the quad, gate (+ memory), inverse radix-4, I1 --------------------------
  {
    fftA<8, 8, false>(z);                              // E_h[k2'] at position 8 ka + kb, k2' = ka + 8 kb
    static_for<0, 8>([&](auto kac) { pin8<8 * decltype(kac)::value>(z); });   // whole butterflies, not lazily per output
    const float s2 = (h & 2) ? -1.f : 1.f;             // lane ^ 2 stage: lanes 0,1 own + partner; lanes 2,3 partner - own
    const float s1 = (h & 1) ? -1.f : 1.f;             // lane ^ 1 stage
    const bool h3 = (h == 3);                          // the lane whose intermediate carries the -i (forward) / +i (inverse) factor
    const bool upper = r >= 2;                         // k2 = k2' + 64 r >= 128: conj(g[N - k]) (Nyquist at k1 = 0, k2' = 0, r = 2)
    const int grp = cb / a.d_g;
    const float2* gp = a.gate + (size_t)b * a.G * a.F;   // workgroup-uniform base + 32-bit lane offset: one address register per load
    // bin k = k1 + RF (k2' + 64 r); upper half reads g[N - k]: both are  base + step * k2'  with lane constants base, step
    int gbase = grp * a.F + (upper ? RF * (256 - 64 * r) - k1c : k1c + RF * 64 * r);
    int gstep = upper ? -RF : RF;
    int mbase = upper ? RF * (256 - 64 * r) - k1c : k1c + RF * 64 * r;
    const float2* tw256 = a.tw;
    int hstep = (N / 256) * h;
    static_for<0, 64>([&](auto jc) {
      constexpr int j = decltype(jc)::value, k2p = (j / 8) + 8 * (j % 8);
      if constexpr (j % 8 == 0 && j > 0) {            // gate / twiddle loads at most 8 deep (register budget): the opaque copies
        asm volatile("" : "+v"(gbase), "+v"(gstep), "+v"(mbase), "+v"(hstep));   // pin this group's addresses behind the previous group (instruction
        __builtin_amdgcn_sched_barrier(0);             // selection is free to hoist plain loads over a sched_barrier alone)
      }
      // W_256^(h k2') = W_N^((N/256) h k2') from the plan's table (L1-resident; selecting among literals costs more registers)
      float wc = 1.f, ws = 0.f;
      if constexpr (k2p > 0) {
        const float2 w = tw256[hstep * k2p];
        wc = w.x; ws = -w.y;                           // table holds exp(-i ...) = (cos, -sin)
      }
      float2 v = z[j];
      if constexpr (k2p > 0) v = make_float2(v.x * wc + v.y * ws, v.y * wc - v.x * ws);          // * W (forward sign)
      // radix 4 over h, decimation in time: (a+c, b+d | a-c, b-d), -i on lane 3, then (X0, X2 | X1, X3)
      float2 t = make_float2(fmaf(s2, v.x, dpp_swap2(v.x)), fmaf(s2, v.y, dpp_swap2(v.y)));
      t = h3 ? make_float2(t.y, -t.x) : t;
      const float2 x = make_float2(fmaf(s1, t.x, dpp_swap1(t.x)), fmaf(s1, t.y, dpp_swap1(t.y)));
      float2 g = gp[gbase + gstep * k2p];
      if (a.conj_gate) g.y = -g.y;
      const bool edge = (k2p == 0) && (k1c == 0) && ((r & 1) == 0);     // DC (r = 0) and Nyquist (r = 2)
      if (edge) g.y = 0.f;
      if (upper) g.y = -g.y;
      g.x *= inv_n; g.y *= inv_n;
      float2 y = cmul(x, g);
      if constexpr (WITH_MEM) {                         // spectre.py:548-549
        const float4 m = *reinterpret_cast<const float4*>(a.mem + ((size_t)(mbase + gstep * k2p) * a.D + cb) * 2);
        float2 add;
        if (edge)       add = make_float2(m.x, m.z);
        else if (upper) add = make_float2(m.x + m.w, m.z - m.y);
        else            add = make_float2(m.x - m.w, m.y + m.z);
        y.x += add.x * inv_n; y.y += add.y * inv_n;
      }
      // inverse radix 4, decimation in frequency: (Y0+Y2, Y0-Y2 | Y1+Y3, Y1-Y3), +i on lane 3, then lane h = time index h
      float2 u = make_float2(fmaf(s1, y.x, dpp_swap1(y.x)), fmaf(s1, y.y, dpp_swap1(y.y)));
      u = h3 ? make_float2(-u.y, u.x) : u;
      u = make_float2(fmaf(s2, u.x, dpp_swap2(u.x)), fmaf(s2, u.y, dpp_swap2(u.y)));
      if constexpr (k2p > 0) u = make_float2(u.x * wc - u.y * ws, u.y * wc + u.x * ws);          // * conj(W)
      z[j] = u;
    });
    static_for<0, 8>([&](auto kac) { pin8<8 * decltype(kac)::value>(z); });
    static_for<0, 8>([&](auto kac) { fftB_stage1_group<8, 8, true, decltype(kac)::value>(z); });
    fftB_stage2<8, 8, true>(z);                        // C[n2 = 4 m + h] at position m
  }

  // ---- E2: (k1, n2 = 4m + h) -> row thread n2, slot sigma(k1) -----------------------------------------------------------
  {
    float* wbase = img + h * kQuadRow2 + pb * kQuadCol2 + quad_sigma(k1c);
    const float* rbase = img + n2 * kQuadRow2 + pa * kQuadCol2;
    if (bins) static_for<0, 64>([&](auto mc) { constexpr int m = decltype(mc)::value; wbase[m * 4 * kQuadRow2] = z[m].x; });
    __syncthreads();
    static_for<0, RF>([&](auto kc) { constexpr int kk = decltype(kc)::value; z[kk].x = rbase[quad_sigma(kk)]; });
    __syncthreads();
    if (bins) static_for<0, 64>([&](auto mc) { constexpr int m = decltype(mc)::value; wbase[m * 4 * kQuadRow2] = z[m].y; });
    __syncthreads();
    static_for<0, RF>([&](auto kc) { constexpr int kk = decltype(kc)::value; z[kk].y = rbase[quad_sigma(kk)]; });
  }

  // ---- conj twiddle, I2 over k1, store rows n2 + 256 n1 ----------------------------------------------------------------
  {
    float2 wa[RAF], wb[RBF];
    load_twiddle_bases(wa, wb);
    static_for<1, RF>([&](auto jc) {
      constexpr int j = decltype(jc)::value, ja = j % RAF, jb = j / RAF;
      if constexpr (ja > 0) z[j] = cmulc(z[j], wa[ja]);
      if constexpr (jb > 0) z[j] = cmulc(z[j], wb[jb]);
    });
    fft_ct<RF, true, IdentityMap, 64>(z);
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

// f32 -> f32 and bf16 -> bf16 only (mixed storage dtypes take the LDS path), three modes each
template <int RF>
inline hipError_t launch_regtile_quad(const RegtileArgs& a, bool in_bf16, bool out_bf16, int mode, hipStream_t stream) {
  if (in_bf16 != out_bf16) return hipErrorInvalidValue;
  const dim3 grid(a.n_wg), block(512);
  const size_t lds = quad_image_bytes(RF);
  const int key = (in_bf16 ? 4 : 0) | mode;
  static std::atomic<bool> lds_opt_in[16][8];
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
    case 0: return go(spectre_mix_regtile_quad<RF, false, false, 0>);
    case 1: return go(spectre_mix_regtile_quad<RF, false, false, 1>);
    case 2: return go(spectre_mix_regtile_quad<RF, false, false, 2>);
    case 4: return go(spectre_mix_regtile_quad<RF, true, true, 0>);
    case 5: return go(spectre_mix_regtile_quad<RF, true, true, 1>);
    case 6: return go(spectre_mix_regtile_quad<RF, true, true, 2>);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sfft
