// kernel_regtile_widep.h — PERSISTENT whole-line tiles for n_fft <= 1024: kernel_regtile_wide.h (32 channels = whole 128-byte lines per row)
// with the next tile's rows in flight while the current one is computed.  Replaces /root/reference/spectre.py:506 + :542-553, fast mode.
//
// At n_fft <= 1024 a 32-channel tile is at most 128 KiB = 64 registers per thread, so — unlike at 4096, where kernel_regtile64p.h needs
// LDS-DMA landing slots, deferred groups and hand-counted waits to keep requests in flight — the software pipeline is plain register
// double buffering: one workgroup per CU (n_fft = 1024: 8 waves, two per SIMD, up to 256 registers) walks through its tiles;
//     top of tile t:   the rows of tile t (requested a tile ago) become the working set; the rows of tile t+1 are REQUESTED into the second
//                      register set (unconditionally: an empty buffer range after the last tile), its gate bins into two registers
//     ...              F1, E1, F2 -> gate -> I1, E2, I2 exactly as in kernel_regtile_wide.h
//     end of tile t:   stores (non-temporal: nobody shares a whole line)
// The loads of tile t+1 are OLDER than the stores of tile t, so waiting for them at the top of tile t+1 does not wait for a store to be
// acknowledged (one in-order vmcnt); every barrier of the exchanges is an LDS-only barrier (s_waitcnt lgkmcnt(0) ; s_barrier): a
// __syncthreads() carries a fence that hipcc implements with vmcnt(0), which would drain the prefetch at the first barrier of every tile.
#pragma once
#include "../../fft_amd/csrc/kernel_regtile_wide.h"

namespace sfft {

__device__ __forceinline__ void widep_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// exchange_planes_b128_w2 of kernel_regtile.h with LDS-only barriers (same layout, same instruction forms)
template <int E, int RA, int RB, bool LAST_BARRIER, int ROWS, int NT, int TS, int RWF, class POS, class RD4>
__device__ __forceinline__ void exchange_planes_b128_w2_lds(float2 (&z)[E], float* img, int wbase, POS pos, RD4 rd4) {
  static_assert(ROWS * NT == E && ROWS % 4 == 0 && (2 * RWF * 4) % 256 == 0, "row pairs (r, r + 2) must be whole offset units apart");
  constexpr int UNITS2 = 2 * RWF * 4 / 256;                     // offset units between rows r and r + 2
  constexpr int WIN = ((255 / UNITS2) * 2 + 2) / 4 * 4;         // rows per window (multiple of 4): last pair starts at row WIN - 4 + {0,1}
  constexpr int NW = (ROWS + WIN - 1) / WIN;
  typedef __attribute__((address_space(3))) float lds_float;
  lds_float* wb[NT * 2 * NW];
  static_for<0, NT * 2 * NW>([&](auto ic) {
    constexpr int i = decltype(ic)::value, t = i / (2 * NW), par = (i / NW) % 2, w = i % NW;
    wb[i] = (lds_float*)(img + wbase) + t * TS + (w * WIN + par) * RWF;
    asm volatile("" : "+v"(wb[i]));                              // keep them apart: base + 16-bit offset would fold them back together
  });
  constexpr int SUB = RA * RB;
  auto write_plane = [&](auto is_im) {
    static_for<0, NT * (ROWS / 4)>([&](auto ic) {
      constexpr int t = decltype(ic)::value / (ROWS / 4), r0 = 4 * (decltype(ic)::value % (ROWS / 4)), w = r0 / WIN, rw = r0 % WIN;
      static_for<0, 2>([&](auto parc) {
        constexpr int par = decltype(parc)::value;
        constexpr int ja = decltype(pos(std::integral_constant<int, r0 + par>{}, std::integral_constant<int, t>{}))::value;
        constexpr int jb = decltype(pos(std::integral_constant<int, r0 + par + 2>{}, std::integral_constant<int, t>{}))::value;
        lds_float* b = wb[(t * 2 + par) * NW + w];
        if constexpr (decltype(is_im)::value) { b[rw * RWF] = z[ja].y; b[(rw + 2) * RWF] = z[jb].y; }
        else { b[rw * RWF] = z[ja].x; b[(rw + 2) * RWF] = z[jb].x; }
      });
    });
  };
  auto read_plane = [&](auto is_im) {
    static_for<0, E / 4>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int CPS = SUB / 4, CPR = (RB >= 4 ? RB / 4 : 1);
      constexpr int sub = i / CPS, ii = i % CPS;
      constexpr int chunk = (RB >= 4) ? (ii / RA) + CPR * (ii % RA) : ii;
      constexpr int m = sub * SUB + 4 * chunk;
      const float4 v = *reinterpret_cast<const float4*>(img + rd4(std::integral_constant<int, m>{}));
      if constexpr (decltype(is_im)::value) { z[m].y = v.x; z[m + 1].y = v.y; z[m + 2].y = v.z; z[m + 3].y = v.w; }
      else { z[m].x = v.x; z[m + 1].x = v.y; z[m + 2].x = v.z; z[m + 3].x = v.w; }
    });
  };
  write_plane(std::false_type{});
  widep_barrier();
  read_plane(std::false_type{});
  widep_barrier();
  write_plane(std::true_type{});
  widep_barrier();
  read_plane(std::true_type{});
  if constexpr (LAST_BARRIER) widep_barrier();
}

template <int RF, int RS, bool IN_BF16, bool OUT_BF16, int NT = 3>
__global__ void __launch_bounds__(kPCW * RS, 2)
spectre_mix_regtile_widep(const RegtileArgs a) {
  static_assert(RF == RS || RF == 2 * RS, "n_fft = RS*RS or 2*RS*RS");
  constexpr int N = RF * RS, NS = RF / RS, NTHR = kPCW * RS;
  constexpr int RAF = FftCfg<RF>::RA, RBF = FftCfg<RF>::RB;
  constexpr int RAS = FftCfg<RS>::RA, RBS = FftCfg<RS>::RB;
  constexpr int ES_IN = IN_BF16 ? 2 : 4, ES_OUT = OUT_BF16 ? 2 : 4;
  constexpr int GS = (N / 2 + 1 + NTHR - 1) / NTHR;            // gate bins per thread
  constexpr float inv_n = 1.0f / (float)N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);
  float2* glds = reinterpret_cast<float2*>(smem + regtile_wide_image_bytes<RF, RS>());

  const int tid0 = threadIdx.x;
  const int wg_lin = xcd_contiguous(blockIdx.x, a.n_wg);
  const int tile0 = wg_lin * a.tpw;                            // a contiguous run of tiles per workgroup
  if (tile0 >= a.n_tiles) return;

  float2 zn[RF];                                               // rows of the NEXT tile (in flight during the current one)
  float2 gn[GS];                                               // its gate bins, raw
  // requests of tile `tile` (live = false: an empty range — the request is issued all the same, so that every wave's vmcnt sees the same
  // number of operations on every path and hipcc's waits stay exact)
  auto request = [&](int tile, bool live, int p, int u, int tid) {
    const int b = tile / a.tiles_per_row, ct = tile - b * a.tiles_per_row;
    const char* vb = reinterpret_cast<const char*>(a.v) + ((size_t)b * a.v_sb + (size_t)ct * (2 * kPCW)) * ES_IN;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vb), 0, live ? 0x7fffffff : 0, kRsrcFlags);
    const uint32_t voff = (uint32_t)(((long long)u * a.v_sn + 2 * p) * ES_IN);
    constexpr int AUX = (NT & 1) ? 2 : 0;                      // nt
    static_for<0, RF>([&](auto ic) {
      constexpr int q = (decltype(ic)::value / RAF) + RBF * (decltype(ic)::value % RAF);   // order of use in F1
      const uint32_t so = (uint32_t)((long long)q * RS * a.v_sn * ES_IN);
      if constexpr (IN_BF16) {
        const uint32_t wv = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, so, AUX);
        zn[q] = make_float2(__uint_as_float(wv << 16), __uint_as_float(wv & 0xffff0000u));
      } else {
        const rt_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, so, AUX);
        zn[q] = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
      }
    });
    const float2* gp = a.gate + ((size_t)b * a.G + (ct * (2 * kPCW)) / a.d_g) * a.F;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(gp), 0, live ? (N / 2 + 1) * 8 : 0, kRsrcFlags);
    static_for<0, GS>([&](auto ic) {
      const rt_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rg, (uint32_t)((tid + NTHR * decltype(ic)::value) * 8), 0, 0);   // beyond bin N/2: 0
      gn[decltype(ic)::value] = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
    });
  };
  // the twiddle bases W_N^(u ka), W_N^(u RAF kb) depend on the thread only: loaded ONCE, before the first request (a global load issued
  // behind the requests of the next tile would have to wait for all of them: one in-order vmcnt), and kept in 2 (RAF + RBF - 2) registers
  float2 wa[RAF], wb[RBF];
  {
    const int lane = tid0 & 63;
    const int u = (lane / kPCW) + (64 / kPCW) * (tid0 >> 6);
    static_for<1, RAF>([&](auto jc) { constexpr int j = decltype(jc)::value; wa[j] = a.tw[u * j]; });
    static_for<1, RBF>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = a.tw[u * RAF * j]; });
    request(tile0, true, lane & (kPCW - 1), u, tid0);
    // RF stores into an EMPTY range: hipcc derives the waits at the top of the loop from what is guaranteed to be younger than the requests
    // on EVERY path into it; without these the prologue path has no stores behind the requests, the merged count comes out too small and
    // the steady state waits for most of the previous tile's stores to be acknowledged (seen in the ISA: vmcnt(9) instead of vmcnt(32))
    const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0, kRsrcFlags);
    static_for<0, RF>([&](auto ic) {                              // (distinct offsets: identical stores would be merged into one)
      constexpr uint32_t off = 64u * decltype(ic)::value;   // (not adjacent either: neighbours would be fused into wider stores)
      if constexpr (OUT_BF16) __builtin_amdgcn_raw_buffer_store_b32(0u, none, off, 0, 0);
      else { rt_u32x2 t; t.x = 0u; t.y = 0u; __builtin_amdgcn_raw_buffer_store_b64(t, none, off, 0, 0); }
    });
  }

  for (int it = 0; it < a.tpw; ++it) {
    const int tile = tile0 + it;
    if (tile >= a.n_tiles) break;                              // workgroup-uniform
    const bool more = (it + 1 < a.tpw) && (tile + 1 < a.n_tiles);
    int tid = tid0;
    asm volatile("" : "+v"(tid));                              // per-iteration copies: keeps the per-lane addresses out of LICM (kernel_regtile.h)
    const int lane = tid & 63, wave = tid >> 6;
    const int p = lane & (kPCW - 1);
    const int u = (lane / kPCW) + (64 / kPCW) * wave;
    long long out_sn = a.out_sn;
    asm volatile("" : "+s"(out_sn));
    const int b = tile / a.tiles_per_row, ct = tile - b * a.tiles_per_row;

    // the previous tile's last exchange left its image reads (and its gate reads) unfenced
    if (it > 0) widep_barrier();
    // ---- this tile's rows and gate bins arrive (requested a tile ago: older than every store since) -------------------------------
    float2 z[RF];
    static_for<0, RF>([&](auto ic) { z[decltype(ic)::value] = zn[decltype(ic)::value]; });
    static_for<0, GS>([&](auto ic) {
      const int k = tid + NTHR * decltype(ic)::value;
      float2 g = gn[decltype(ic)::value];
      if (k == 0 || k == N / 2) g.y = 0.f;                     // irfft ignores Im(DC), Im(Nyquist) (spectre.py:551)
      if (a.conj_gate) g.y = -g.y;
      if (k <= N / 2) glds[k] = make_float2(g.x * inv_n, g.y * inv_n);   // ordered before its first use by E1's barriers
    });
    __builtin_amdgcn_sched_barrier(0);
    // ---- the next tile is requested NOW: it travels while this one is computed ---------------------------------------------------
    request(tile + 1, more, p, u, tid);
    __builtin_amdgcn_sched_barrier(0);

    // ---- F1 + W_N^(u*k1)
    {
      fftA_stage1<RAF, RBF, false>(z);
      static_for<0, RAF>([&](auto kac) { fftA_stage2_group<RAF, RBF, false, decltype(kac)::value>(z); });
      static_for<1, RF>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int ka = j / RBF, kb = j % RBF;
        if constexpr (ka > 0) z[j] = cmul(z[j], wa[ka]);
        if constexpr (kb > 0) z[j] = cmul(z[j], wb[kb]);
      });
    }
    // ---- E1
    {
      constexpr int PS = RS + 4, RW = kPCW * PS;
      exchange_planes_b128_w2_lds<RF, RAS, RBS, true, RF, 1, 0, RW>(z, img, p * PS + u,
          [](auto rc, auto) { constexpr int k1 = decltype(rc)::value; return std::integral_constant<int, RBF * (k1 % RAF) + k1 / RAF>{}; },
          [&](auto mc) { constexpr int m = decltype(mc)::value; constexpr int t = m / RS, n2 = m % RS;
                         return (u + RS * t) * RW + p * PS + n2; });
    }
    // ---- F2 -> gate (spectre.py:545) -> I1
    {
      static_for<0, NS>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int OFF = t * RS;
        const int k1 = u + RS * t;
        fftA_stage1<RAS, RBS, false, OFF, RF>(z);
        auto fetch_gate = [&](int k2, bool upper) -> float2 {
          float2 g = glds[(k2 >= RS / 2) ? RF * (RS - k2) - k1 : k1 + RF * k2];
          if (upper) g.y = -g.y;
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
          __builtin_amdgcn_sched_barrier(0);
        });
        fftB_stage2<RAS, RBS, true, OFF, RF>(z);
      });
    }
    // ---- E2
    {
      constexpr int PS = RF + 4, RW = kPCW * PS;
      exchange_planes_b128_w2_lds<RF, RAF, RBF, false, RS, RF / RS, RS, RW>(z, img, p * PS + u,
          [](auto rc, auto tc) { return std::integral_constant<int, decltype(rc)::value + RS * decltype(tc)::value>{}; },
          [&](auto mc) { constexpr int m = decltype(mc)::value; return u * RW + p * PS + m; });
    }
    // ---- conj twiddle, I2, stores (spectre.py:553)
    {
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
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(ob, 0, 0x7fffffff, kRsrcFlags);
      const uint32_t ooff = (uint32_t)(((long long)u * out_sn + 2 * p) * ES_OUT);
      constexpr int AUXS = (NT & 2) ? 2 : 0;
      static_for<0, RF>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr ((j % RBF) == 0) fftA_stage2_group<RAF, RBF, true, j / RBF>(z);
        constexpr int n1 = (j / RBF) + RAF * (j % RBF);
        const uint32_t so = (uint32_t)((long long)n1 * RS * out_sn * ES_OUT);
        if constexpr (OUT_BF16) {
          __builtin_amdgcn_raw_buffer_store_b32(f32_to_bf16_rne(z[j].x) | (f32_to_bf16_rne(z[j].y) << 16), ro, ooff, so, AUXS);
        } else {
          rt_u32x2 t;
          t.x = __float_as_uint(z[j].x); t.y = __float_as_uint(z[j].y);
          __builtin_amdgcn_raw_buffer_store_b64(t, ro, ooff, so, AUXS);
        }
      });
    }
  }
}

template <int RF, int RS>
hipError_t launch_regtile_widep(const RegtileArgs& a, bool in_bf16, bool out_bf16, hipStream_t stream);

#define SFFT_DEFINE_REGTILE_WIDEP_LAUNCHER(RF_, RS_)                                                         \
  template <>                                                                                                \
  hipError_t launch_regtile_widep<RF_, RS_>(const RegtileArgs& a, bool in_bf16, bool out_bf16, hipStream_t stream) { \
    const dim3 grid(a.n_wg), block(regtile_wide_threads<RF_, RS_>());                                        \
    const size_t lds = regtile_wide_lds_total<RF_, RS_>();                                                   \
    const int key = (in_bf16 ? 2 : 0) | (out_bf16 ? 1 : 0);                                                  \
    static std::atomic<bool> lds_opt_in[16][4];                                                              \
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
      case 0: return go(spectre_mix_regtile_widep<RF_, RS_, false, false, 3>);                               \
      case 2: return go(spectre_mix_regtile_widep<RF_, RS_, true, false, 3>);                                \
      case 3: return go(spectre_mix_regtile_widep<RF_, RS_, true, true, 0>);                                 \
      default: return hipErrorInvalidValue;                                                                  \
    }                                                                                                        \
  }

}  // namespace sfft
