// kernel_regtile_wide.h — the register-tile spectral mix with 32-CHANNEL tiles: a workgroup owns WHOLE 128-byte lines of every row
// (n_fft = RF*RS <= 1024; replaces /root/reference/spectre.py:506 + :542-553 exactly like kernel_regtile.h, fast mode only).
//
// Why (round 4, tools/store_lab.hip, profiles/r04_store_lab_half_line_stores.log): the 16-channel tile of kernel_regtile.h is HALF a line
// per row, and the L2 takes a half-line store at two thirds of the rate of a full-line one (3.4-3.6 against 5.3-5.5 TB/s; loads 5.3
// against 5.9) even though the two halves merge before they reach HBM; the kernels that use it sit exactly on the sum of their half-line
// load and store passes.  At n_fft = 4096 sixteen channels are all a CU can hold (256 KiB of registers); at n_fft <= 1024 thirty-two
// channels are 128 KiB (two workgroups per CU at 1024), and every request of the kernel is a full line.
//
// Same mathematics, same phases, same exchange code as kernel_regtile.h (F1 -> twiddle -> E1 -> F2 -> gate -> I1 -> E2 -> conj twiddle ->
// I2); what changes is the geometry:
//   tile      32 channels = 16 packed sequences x n_fft rows;  threads = 16 * RS;  lane = (p = lane & 15, row class = lane >> 4)
//   requests  a wave-wide 8-byte access = 4 rows x 128 contiguous bytes (bf16 rows: 4 bytes per lane, 64-byte half lines)
//   image     [row][column p][slot], column stride R + 4, row stride 16 (R + 4) floats — the row stride is a multiple of 64 dwords, so the
//             16 lanes of every ds_read_b128 group (which hold p = 0..15 once: {0-3}, {12-15}, {4-11 of the next row class}) hit
//             36 p mod 64 = 16 distinct 16-byte bank groups: conflict-free; the dword writes are 2-way (p and p + 8 share a bank), which
//             the LDS absorbs inside a ds_write2's transfer time (MI355X_MICROARCH.md)
//   tiles     one tile per workgroup, XCD-contiguous order; no pairing needed — nobody shares a line
// Conditions (spectre_hip.hip checks them and falls back to kernel_regtile.h otherwise): N_in >= n_fft, no memory_fft, d_g % 32 == 0,
// 8-byte aligned fp32 rows (4-byte aligned bf16 rows).
#pragma once
#include "kernel_regtile.h"
#include <cstdlib>

#ifndef SPECTRE_WIDE_MAP
#define SPECTRE_WIDE_MAP 0
#endif

namespace sfft {

typedef float wide_f32x2 __attribute__((ext_vector_type(2)));
constexpr int kPCW = 16;                     // pair-columns per wide tile: 32 channels, 128-byte fp32 row segments

template <int RF, int RS> constexpr int regtile_wide_threads() { return kPCW * RS; }
template <int RF, int RS> constexpr int regtile_wide_image_bytes() {
  return (RF * kPCW * (RS + 4) > RS * kPCW * (RF + 4) ? RF * kPCW * (RS + 4) : RS * kPCW * (RF + 4)) * 4;
}
template <int RF, int RS> constexpr int regtile_wide_lds_total() { return regtile_wide_image_bytes<RF, RS>() + regtile_gate_lds_bytes<RF, RS>(); }
// WPS = waves per SIMD the kernel is compiled for: 4 (128 registers: two 512-thread workgroups per CU at n_fft = 1024); the 64 x 32
// instantiation (n_fft = 2048: one 512-thread workgroup per CU, 155 KiB of LDS) takes 2
// PADDED: N_in < n_fft (spectre.py:506 zero-pads, :553 keeps the first N_in rows): rows >= N_in are the out-of-range case of buffer
// instructions — loads return 0, stores are dropped — so a padded sequence costs what a full one costs, without a predicate
// (kernel_regtile64p.h; the range check covers the VGPR offset only, so the row-block offset is added there).
template <int RF, int RS, bool IN_BF16, bool OUT_BF16, int WPS = (RF >= 64 ? 2 : 4), int NT = 0, bool PADDED = false>   // NT: 1 = non-temporal loads, 2 = stores, 3 = both
__global__ void __launch_bounds__(kPCW * RS, WPS)
spectre_mix_regtile_wide(const RegtileArgs a) {
  static_assert(RF == RS || RF == 2 * RS, "n_fft = RS*RS or 2*RS*RS");
  static_assert(RS % 4 == 0 && (64 / kPCW) * (kPCW * RS / 64) == RS, "row classes: 4 per wave");
  constexpr int N = RF * RS, NS = RF / RS;
  constexpr int RAF = FftCfg<RF>::RA, RBF = FftCfg<RF>::RB;
  constexpr int RAS = FftCfg<RS>::RA, RBS = FftCfg<RS>::RB;
  constexpr int ES_IN = IN_BF16 ? 2 : 4, ES_OUT = OUT_BF16 ? 2 : 4;
  constexpr float inv_n = 1.0f / (float)N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);
  float2* glds = reinterpret_cast<float2*>(smem + regtile_wide_image_bytes<RF, RS>());

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int p = lane & (kPCW - 1);
  const int u = (lane / kPCW) + (64 / kPCW) * wave;   // team index: n2 in F1/I2, k1 mod RS in F2/I1

  // tile of this workgroup.  The hardware hands workgroups to the XCDs round-robin in blockIdx order; nobody shares a line here, so any map is
  // correct.  SPECTRE_WIDE_MAP (compile time, A/B through tools/build_variant.sh): 0 = XCD-contiguous (every XCD walks through its own eighth
  // of the tensor), 1 = blockIdx order (the whole chip inside one window of ~21 batch elements, in address order), 2 = batch-major
  // (consecutive workgroups in different batch elements).
  const int tile = SPECTRE_WIDE_MAP == 1 ? (int)blockIdx.x
                 : SPECTRE_WIDE_MAP == 2 ? (int)((blockIdx.x % (unsigned)a.B) * a.tiles_per_row + blockIdx.x / (unsigned)a.B)
                                         : xcd_contiguous(blockIdx.x, a.n_wg);
  if (tile >= a.n_tiles) return;
  const long long v_sn = a.v_sn, out_sn = a.out_sn;
  const int b = tile / a.tiles_per_row;
  const int ct = tile - b * a.tiles_per_row;

  auto load_twiddle_bases = [&](float2 (&wa)[RAF], float2 (&wb)[RBF]) {
    static_for<1, RAF>([&](auto jc) { constexpr int j = decltype(jc)::value; wa[j] = a.tw[u * j]; });
    static_for<1, RBF>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = a.tw[u * RAF * j]; });
  };

  // ---- gate -> LDS: the N/2+1 bins of the tile's group, once per tile, pre-scaled by 1/N, Im(DC) and Im(Nyquist) dropped
  //      (spectre.py:551: irfft ignores them); issued before the tile's rows (VMEM returns in order)
  {
    const float2* gp = a.gate + ((size_t)b * a.G + (ct * (2 * kPCW)) / a.d_g) * a.F;
    for (int k = tid; k <= N / 2; k += kPCW * RS) {
      float2 g = gp[k];
      if (k == 0 || k == N / 2) g.y = 0.f;
      if (a.conj_gate) g.y = -g.y;
      glds[k] = make_float2(g.x * inv_n, g.y * inv_n);
    }
  }

  float2 z[RF];

  // ---- load: rows u + RS*q (spectre.py:506; N_in >= n_fft here), this lane's channel pair: 16 lanes x 8 bytes = one 128-byte line
  {
    const char* vb = reinterpret_cast<const char*>(a.v) + ((size_t)b * a.v_sb + (size_t)ct * (2 * kPCW)) * ES_IN;
    const uint32_t voff = (uint32_t)(((long long)u * v_sn + 2 * p) * ES_IN);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vb), 0, (int)((long long)a.rows_in * v_sn * ES_IN), kRsrcFlags);
    static_for<0, RF>([&](auto ic) {
      constexpr int q = (decltype(ic)::value / RAF) + RBF * (decltype(ic)::value % RAF);   // order of use in F1
      if constexpr (PADDED) {
        const uint32_t off = voff + (uint32_t)((long long)q * RS * v_sn * ES_IN);
        if constexpr (IN_BF16) {
          const uint32_t wv = __builtin_amdgcn_raw_buffer_load_b32(rs_in, off, 0, (NT & 1) ? 2 : 0);
          z[q] = make_float2(__uint_as_float(wv << 16), __uint_as_float(wv & 0xffff0000u));
        } else {
          const rt_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs_in, off, 0, (NT & 1) ? 2 : 0);
          z[q] = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
        }
        return;
      }
      const char* ptr = vb + (size_t)q * RS * v_sn * ES_IN + voff;
      if constexpr (IN_BF16) {
        uint32_t wv;
        if constexpr (NT & 1) wv = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(ptr)); else wv = *reinterpret_cast<const uint32_t*>(ptr);
        z[q] = make_float2(__uint_as_float(wv << 16), __uint_as_float(wv & 0xffff0000u));
      } else {
        if constexpr (NT & 1) { const wide_f32x2 t = __builtin_nontemporal_load(reinterpret_cast<const wide_f32x2*>(ptr)); z[q] = make_float2(t.x, t.y); }
        else z[q] = *reinterpret_cast<const float2*>(ptr);
      }
    });
  }

  // ---- F1: RF-point forward transform over n1, then W_N^(u*k1)
  {
    fftA_stage1<RAF, RBF, false>(z);
    float2 wa[RAF], wb[RBF];
    __builtin_amdgcn_sched_barrier(0);
    load_twiddle_bases(wa, wb);
    static_for<0, RAF>([&](auto kac) { fftA_stage2_group<RAF, RBF, false, decltype(kac)::value>(z); });
    static_for<1, RF>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int ka = j / RBF, kb = j % RBF;
      if constexpr (ka > 0) z[j] = cmul(z[j], wa[ka]);
      if constexpr (kb > 0) z[j] = cmul(z[j], wb[kb]);
    });
  }

  // ---- E1: position j (k1 = ka + RAF*kb) -> image row k1, column (p, u); thread u reads rows u + RS*t
  {
    constexpr int PS = RS + 4, RW = kPCW * PS;
    exchange_planes_b128_w2<RF, RAS, RBS, true, RF, 1, 0, RW>(z, img, p * PS + u,
        [](auto rc, auto) { constexpr int k1 = decltype(rc)::value; return std::integral_constant<int, RBF * (k1 % RAF) + k1 / RAF>{}; },
        [&](auto mc) { constexpr int m = decltype(mc)::value; constexpr int t = m / RS, n2 = m % RS;
                       return (u + RS * t) * RW + p * PS + n2; });
  }

  // ---- middle: F2 -> gate (spectre.py:545) -> I1, register group by register group (kernel_regtile.h)
  {
    static_for<0, NS>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int OFF = t * RS;
      const int k1 = u + RS * t;
      fftA_stage1<RAS, RBS, false, OFF, RF>(z);
      auto fetch_gate = [&](int k2, bool upper) -> float2 {
        float2 g = glds[(k2 >= RS / 2) ? RF * (RS - k2) - k1 : k1 + RF * k2];      // scaled, edges fixed
        if (upper) g.y = -g.y;                                                     // Hermitian extension above N/2
        return g;
      };
      float2 gcur[RBS];
      static_for<0, RBS>([&](auto kbc) {
        constexpr int k2 = RAS * decltype(kbc)::value;
        gcur[decltype(kbc)::value] = fetch_gate(k2, k2 >= RS / 2);
      });
      static_for<0, RAS>([&](auto kac) {
        constexpr int ka = decltype(kac)::value;
        fftA_stage2_group<RAS, RBS, false, ka, OFF, RF>(z);
        static_for<0, RBS>([&](auto kbc) {
          constexpr int kb = decltype(kbc)::value;
          z[OFF + RBS * ka + kb] = cmul(z[OFF + RBS * ka + kb], gcur[kb]);
        });
        if constexpr (ka + 1 < RAS) {
          static_for<0, RBS>([&](auto kbc) {
            constexpr int k2n = ka + 1 + RAS * decltype(kbc)::value;
            gcur[decltype(kbc)::value] = fetch_gate(k2n, k2n >= RS / 2);
          });
        }
        fftB_stage1_group<RAS, RBS, true, ka, OFF, RF>(z);
        __builtin_amdgcn_sched_barrier(0);           // keep the gate fetch one group deep (128-register budget)
      });
      fftB_stage2<RAS, RBS, true, OFF, RF>(z);
    });
  }

  // ---- E2: position t*RS + n2 -> image row n2, column (p, k1 = u + RS*t); thread u reads its row, slot k1
  {
    constexpr int PS = RF + 4, RW = kPCW * PS;
    exchange_planes_b128_w2<RF, RAF, RBF, false, RS, RF / RS, RS, RW>(z, img, p * PS + u,
        [](auto rc, auto tc) { return std::integral_constant<int, decltype(rc)::value + RS * decltype(tc)::value>{}; },
        [&](auto mc) { constexpr int m = decltype(mc)::value; return u * RW + p * PS + m; });
  }

  // ---- conj twiddle, I2, store (spectre.py:553; all n_fft rows exist here)
  {
    float2 wa[RAF], wb[RBF];
    load_twiddle_bases(wa, wb);
    static_for<1, RF>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int ja = j % RAF, jb = j / RAF;
      if constexpr (ja > 0) z[j] = cmulc(z[j], wa[ja]);
      if constexpr (jb > 0) z[j] = cmulc(z[j], wb[jb]);
    });
    fftA_stage1<RAF, RBF, true>(z);
  }
  {
    char* ob = reinterpret_cast<char*>(a.out) + ((size_t)b * a.out_sb + (size_t)ct * (2 * kPCW)) * ES_OUT;
    const uint32_t ooff = (uint32_t)(((long long)u * out_sn + 2 * p) * ES_OUT);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)((long long)a.rows_out * out_sn * ES_OUT), kRsrcFlags);
    static_for<0, RF>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr ((j % RBF) == 0) fftA_stage2_group<RAF, RBF, true, j / RBF>(z);
      constexpr int n1 = (j / RBF) + RAF * (j % RBF);
      if constexpr (PADDED) {
        const uint32_t off = ooff + (uint32_t)((long long)n1 * RS * out_sn * ES_OUT);
        if constexpr (OUT_BF16) {
          __builtin_amdgcn_raw_buffer_store_b32(f32x2_to_bf16x2_rne(z[j].x, z[j].y), rs_out, off, 0, (NT & 2) ? 2 : 0);
        } else {
          rt_u32x2 t;
          t.x = __float_as_uint(z[j].x); t.y = __float_as_uint(z[j].y);
          __builtin_amdgcn_raw_buffer_store_b64(t, rs_out, off, 0, (NT & 2) ? 2 : 0);
        }
        return;
      }
      char* ptr = ob + (size_t)n1 * RS * out_sn * ES_OUT + ooff;
      if constexpr (OUT_BF16) {
        const uint32_t w = f32x2_to_bf16x2_rne(z[j].x, z[j].y);
        if constexpr (NT & 2) __builtin_nontemporal_store(w, reinterpret_cast<uint32_t*>(ptr)); else *reinterpret_cast<uint32_t*>(ptr) = w;
      }
      else if constexpr (NT & 2) { wide_f32x2 t; t.x = z[j].x; t.y = z[j].y; __builtin_nontemporal_store(t, reinterpret_cast<wide_f32x2*>(ptr)); }
      else *reinterpret_cast<float2*>(ptr) = z[j];
    });
  }
}

template <int RF, int RS>
hipError_t launch_regtile_wide(const RegtileArgs& a, bool in_bf16, bool out_bf16, bool padded, hipStream_t stream);

#define SFFT_DEFINE_REGTILE_WIDE_LAUNCHER(RF_, RS_)                                                          \
  template <>                                                                                                \
  hipError_t launch_regtile_wide<RF_, RS_>(const RegtileArgs& a, bool in_bf16, bool out_bf16, bool padded, hipStream_t stream) { \
    const dim3 grid(a.n_wg), block(regtile_wide_threads<RF_, RS_>());                                        \
    const size_t lds = regtile_wide_lds_total<RF_, RS_>();                                                   \
    const int key = (in_bf16 ? 2 : 0) | (out_bf16 ? 1 : 0);                                                  \
    constexpr int W = (RF_ >= 64 ? 2 : 4);                                                                   \
    /* non-temporal accesses: whole-line requests that nobody else shares.  Measured (tools/wide_nt_ab.py, wide_nt_bf16_ab.py, one box,  */ \
    /* interleaved): fp32 rows, both directions nt: -4.8 % at 1024, -4.5 % at 512; bf16 -> fp32: -2.8 %; bf16 -> bf16 (64-byte halves    */ \
    /* shared with the neighbouring tile in BOTH directions): +3 %, stays plain.  SPECTRE_WIDE_NT (under SPECTRE_TUNING=1) overrides.     */ \
    static const int nt_env = [] { const char* t = getenv("SPECTRE_TUNING"); const char* e = getenv("SPECTRE_WIDE_NT"); return (t && atoi(t) == 1 && e) ? atoi(e) : -1; }(); \
    const int nt = nt_env >= 0 ? (nt_env & 3) : (key == 0 || key == 2 ? 3 : 0);                                         \
    static std::atomic<bool> lds_opt_in[16][32];                                                             \
    auto go = [&](auto kern) -> hipError_t {                                                                 \
      int dev = 0;                                                                                           \
      (void)hipGetDevice(&dev);                                                                              \
      if (dev < 0 || dev >= 16 || !lds_opt_in[dev][key * 4 + nt + (padded ? 16 : 0)]) {                                                   \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e != hipSuccess) return e;                                                                       \
        if (dev >= 0 && dev < 16) lds_opt_in[dev][key * 4 + nt + (padded ? 16 : 0)] = true;                                               \
      }                                                                                                      \
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                                 \
      return hipGetLastError();                                                                              \
    };                                                                                                       \
    if (padded) {   /* N_in < n_fft: buffer instructions with the rows that exist as their range */              \
      switch (key) {                                                                                         \
        case 0: return nt ? go(spectre_mix_regtile_wide<RF_, RS_, false, false, W, 3, true>) : go(spectre_mix_regtile_wide<RF_, RS_, false, false, W, 0, true>); \
        case 2: return nt ? go(spectre_mix_regtile_wide<RF_, RS_, true, false, W, 3, true>) : go(spectre_mix_regtile_wide<RF_, RS_, true, false, W, 0, true>);   \
        case 3: return go(spectre_mix_regtile_wide<RF_, RS_, true, true, W, 0, true>);                         \
        default: return hipErrorInvalidValue;                                                                \
      }                                                                                                      \
    }                                                                                                        \
    switch (key * 4 + nt) {                                                                                  \
      case 0: return go(spectre_mix_regtile_wide<RF_, RS_, false, false, W, 0>);                             \
      case 1: return go(spectre_mix_regtile_wide<RF_, RS_, false, false, W, 1>);                             \
      case 2: return go(spectre_mix_regtile_wide<RF_, RS_, false, false, W, 2>);                             \
      case 3: return go(spectre_mix_regtile_wide<RF_, RS_, false, false, W, 3>);                             \
      case 8: return go(spectre_mix_regtile_wide<RF_, RS_, true, false, W, 0>);                              \
      case 11: return go(spectre_mix_regtile_wide<RF_, RS_, true, false, W, 3>);                             \
      case 12: return go(spectre_mix_regtile_wide<RF_, RS_, true, true, W, 0>);                              \
      case 15: return go(spectre_mix_regtile_wide<RF_, RS_, true, true, W, 3>);                              \
      case 9: case 10: return go(spectre_mix_regtile_wide<RF_, RS_, true, false, W, 0>);                     \
      case 13: case 14: return go(spectre_mix_regtile_wide<RF_, RS_, true, true, W, 0>);                     \
      default: return hipErrorInvalidValue;   /* f32 -> bf16 stays with kernel_regtile.h */                   \
    }                                                                                                        \
  }

}  // namespace sfft
