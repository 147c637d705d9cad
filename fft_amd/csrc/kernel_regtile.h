// kernel_regtile.h — register-resident spectral mix for n_fft = R*R (R = 16, 32, 64) on gfx950.
//
// One workgroup owns a tile of 16 adjacent channels (one 64-byte row segment in fp32) for ALL n_fft rows
// of one batch element; the tile never leaves the CU between the single HBM read and the single HBM write
// (replaces /root/reference/spectre.py:506 + :542-553, which make 8-9 HBM passes).
//
// Math.  Two real channels (c, c+1) of the same gate group are packed as one complex sequence
// z = x_c + i x_{c+1}.  Because the filter's impulse response irfft(gate) is REAL, filtering acts on Re and
// Im independently, so   y_c + i y_{c+1} = IDFT( Gf * DFT(z) + Mf ),   with Gf the Hermitian extension of the
// half-spectrum gate (Im dropped at DC and Nyquist — spectre.py:551's irfft ignores them) and
// Mf[k] = mem_c[k] + i mem_{c+1}[k] (k <= N/2), conj(mem_c[N-k]) + i conj(mem_{c+1}[N-k]) otherwise.
//
// The length-N complex DFT is the two-pass Cooley-Tukey split n = n2 + R n1, k = k1 + R k2:
//   F1  thread (p,u):  A[k1]  = sum_n1 z[u + R n1] W_R^(n1 k1)          (in registers, type A)
//                      A[k1] *= W_N^(u k1)                               (per-thread twiddle bases)
//   E1  exchange through LDS: value (u, k1) -> thread k1, slot u
//   F2  thread (p,s):  X[s + R k2] = sum_n2 A_n2[s] W_R^(n2 k2)          (type A; last stage fused with ..)
//       gate:          Y = X * Gf (+ Mf), 1/N folded in
//   I1                 C[n2]  = sum_k2 Y[s + R k2] W_R^(-n2 k2)          (.. the first stage of type B)
//                      C[n2] *= conj(W_N^(s n2))
//   E2  exchange: value (s, n2) -> thread n2, slot s
//   I2  thread (p,u):  y[u + R n1] = sum_k1 C_k1[u] W_R^(-n1 k1)         (type A, inverse)
//
// Geometry.  lane = (p = lane & 7 : pair-column, row class = lane >> 3), u = row class + 8 * wave, so a
// wave-wide 8-byte load covers 8 rows x 64 contiguous bytes.  64-byte segments reach the copy ceiling only
// when the neighbouring tile (other half of the 128-B line) is in flight in the same XCD's L2 at the same
// time: tiles are therefore dealt to workgroups XCD-contiguously (profiles/r01_segcopy_microbench.log).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_regs.h"

namespace sfft {

struct RegtileArgs {
  const void* v;
  const float2* gate;   // (B, G, F) complex64
  const float* mem;     // (F, D, 2) float or nullptr
  void* out;
  const float2* tw;     // exp(-2 pi i m / N), m = 0..N-1 (host-computed in double)
  int B, N_in, D, G, d_g, F;
  int tiles_per_row, n_tiles;
  long long v_sb, v_sn, out_sb, out_sn;   // element strides
};

template <int R> struct RegtileCfg;
template <> struct RegtileCfg<64> { static constexpr int RA = 8, RB = 8; };
template <> struct RegtileCfg<32> { static constexpr int RA = 4, RB = 8; };
template <> struct RegtileCfg<16> { static constexpr int RA = 4, RB = 4; };

template <int R> constexpr int regtile_threads() { return 8 * R; }
template <int R> constexpr int regtile_rowb() { return R * 32 + 32; }          // LDS bytes per destination index (one float plane)
template <int R> constexpr int regtile_lds_bytes() { return R * regtile_rowb<R>(); }

// Workgroup id -> tile.  Workgroup w is observed to run on XCD w % 8 (speed only, never correctness):
// give every XCD a contiguous run of tiles so tiles sharing 128-B lines meet in one L2.  Bijective for any n.
__device__ __forceinline__ int xcd_contiguous(int wg, int n) {
  const int nx = 8;
  const int q = n / nx, rem = n % nx;
  const int xcd = wg % nx, idx = wg / nx;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
}

__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
  uint32_t x = __float_as_uint(f);
  if ((x & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;   // NaN
  x += 0x7fffu + ((x >> 16) & 1u);
  return x >> 16;
}

// LDS exchange: the value at register position j goes to thread dest(j), which receives it in slot
// (sender's team index).  DIGREV: position j = RB*ka + kb carries index ka + RA*kb (type-A output);
// otherwise position j carries index j (type-B output).
// The tile (R*R*8 columns*8 B = 256 KiB at R = 64) does not fit the 160 KiB LDS, so real and imaginary
// parts go through the same R*R*8*4-byte image one after the other; this also keeps the live register
// set at R complex values (R re in + R im out) instead of 1.5 R for a two-round 8-byte exchange.
// Every ds_write_b32 is lane-linear (256 B per wave); every ds_read_b32 of a 32-lane group hits 32
// distinct banks thanks to the 32-byte pad per destination row.
template <int R, int RA, int RB, bool DIGREV>
__device__ __forceinline__ void exchange(float2 (&z)[R], char* smem, int p, int u) {
  constexpr int ROWB = regtile_rowb<R>();
  auto dest = [](int j) constexpr { return DIGREV ? (j / RB) + RA * (j % RB) : j; };
  float* wbase = reinterpret_cast<float*>(smem + u * 32 + p * 4);
  const float* rbase = reinterpret_cast<const float*>(smem + u * ROWB + p * 4);
  __syncthreads();                         // everyone done reading the previous exchange
  static_for<0, R>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    wbase[dest(j) * (ROWB / 4)] = z[j].x;
  });
  __syncthreads();
  static_for<0, R>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    z[m].x = rbase[m * 8];
  });
  __syncthreads();
  static_for<0, R>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    wbase[dest(j) * (ROWB / 4)] = z[j].y;
  });
  __syncthreads();
  static_for<0, R>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    z[m].y = rbase[m * 8];
  });
}

// GENERAL=false: N_in >= n_fft (no row predicates) and no memory_fft.  GENERAL=true: both handled.
template <int R, bool IN_BF16, bool OUT_BF16, bool GENERAL>
__global__ void __launch_bounds__(8 * R) spectre_mix_regtile(const RegtileArgs a) {
  constexpr int RA = RegtileCfg<R>::RA, RB = RegtileCfg<R>::RB, N = R * R;
  constexpr int ES_IN = IN_BF16 ? 2 : 4, ES_OUT = OUT_BF16 ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int p = lane & 7;
  const int u = (lane >> 3) + 8 * wave;          // team index: n2 in F1/I2, k1 = s in F2/I1

  const int tile = xcd_contiguous(blockIdx.x, a.n_tiles);
  const int b = tile / a.tiles_per_row;
  const int ct = tile - b * a.tiles_per_row;
  const int c = ct * 16 + 2 * p;                 // first channel of this lane's pair

  // per-thread twiddle bases: W_N^(u*ka) and W_N^(u*RA*kb); products give W_N^(u*j) for any j < R
  float2 wa[RA], wb[RB];
  static_for<1, RA>([&](auto jc) { constexpr int j = decltype(jc)::value; wa[j] = a.tw[u * j]; });
  static_for<1, RB>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = a.tw[u * RA * j]; });

  float2 z[R];

  // ---- load: rows u + R*q, q = 0..R-1 (spectre.py:506 zero-pads / truncates to n_fft) -------------
  {
    const char* vb = reinterpret_cast<const char*>(a.v) + ((size_t)b * a.v_sb + (size_t)ct * 16) * ES_IN;
    const uint32_t voff = (uint32_t)(((long long)u * a.v_sn + 2 * p) * ES_IN);
    static_for<0, R>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const char* ptr = vb + (size_t)q * R * a.v_sn * ES_IN + voff;
      bool ok = true;
      if constexpr (GENERAL) ok = (u + R * q) < a.N_in;
      if (ok) {
        if constexpr (IN_BF16) {
          const uint32_t wv = *reinterpret_cast<const uint32_t*>(ptr);
          z[q] = make_float2(__uint_as_float(wv << 16), __uint_as_float(wv & 0xffff0000u));
        } else {
          z[q] = *reinterpret_cast<const float2*>(ptr);
        }
      } else {
        z[q] = make_float2(0.f, 0.f);
      }
    });
  }

  // ---- F1 -------------------------------------------------------------------------------------------
  fftA<RA, RB, false>(z);
  static_for<1, R>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    constexpr int ka = j / RB, kb = j % RB;      // position j carries k1 = ka + RA*kb
    if constexpr (ka > 0) z[j] = cmul(z[j], wa[ka]);
    if constexpr (kb > 0) z[j] = cmul(z[j], wb[kb]);
  });

  exchange<R, RA, RB, true>(z, smem, p, u);

  // ---- F2 (stage 1), then per register group: F2 stage 2 -> gate -> I1 stage 1 --------------------
  fftA_stage1<RA, RB, false>(z);
  {
    const int grp = c / a.d_g;
    const float2* gp = a.gate + ((size_t)b * a.G + grp) * a.F;
    constexpr float inv_n = 1.0f / (float)N;
    // bin of register (ka, kb): k = u + R*k2, k2 = ka + RA*kb.  k2 >= R/2 means k > N/2 (or k == N/2 for
    // u == 0): the Hermitian extension reads conj(g[N - k]).
    auto gate_index = [&](int k2) { return (k2 >= R / 2) ? R * (R - k2) - u : u + R * k2; };
    float2 gcur[RB], gnxt[RB];
    static_for<0, RB>([&](auto kbc) { constexpr int kb = decltype(kbc)::value; gcur[kb] = gp[gate_index(RA * kb)]; });
    static_for<0, RA>([&](auto kac) {
      constexpr int ka = decltype(kac)::value;
      if constexpr (ka + 1 < RA) {   // software prefetch of the next group's 8 gate bins (16 VGPRs)
        static_for<0, RB>([&](auto kbc) {
          constexpr int kb = decltype(kbc)::value;
          gnxt[kb] = gp[gate_index(ka + 1 + RA * kb)];
        });
      }
      fftA_stage2_group<RA, RB, false, ka>(z);
      static_for<0, RB>([&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        constexpr int j = RB * ka + kb;
        constexpr int k2 = ka + RA * kb;
        constexpr bool upper = k2 >= R / 2;
        constexpr bool edge = (k2 == 0) || (k2 == R / 2);
        float2 g = gcur[kb];
        if constexpr (upper) g.y = -g.y;
        if constexpr (edge) { if (u == 0) g.y = 0.f; }   // irfft ignores Im(DC), Im(Nyquist)
        g.x *= inv_n; g.y *= inv_n;
        z[j] = cmul(z[j], g);
        if constexpr (GENERAL) {
          if (a.mem != nullptr) {
            const int idx = gate_index(k2);
            const float4 m = *reinterpret_cast<const float4*>(a.mem + ((size_t)idx * a.D + c) * 2);
            float2 add;
            if (edge && u == 0) add = make_float2(m.x, m.z);
            else if (upper)     add = make_float2(m.x + m.w, m.z - m.y);
            else                add = make_float2(m.x - m.w, m.y + m.z);
            z[j].x += add.x * inv_n; z[j].y += add.y * inv_n;
          }
        }
      });
      fftB_stage1_group<RA, RB, true, ka>(z);
      if constexpr (ka + 1 < RA) {
        static_for<0, RB>([&](auto kbc) { gcur[decltype(kbc)::value] = gnxt[decltype(kbc)::value]; });
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the gate prefetch one group deep (register budget)
    });
  }
  fftB_stage2<RA, RB, true>(z);
  static_for<1, R>([&](auto jc) {
    constexpr int j = decltype(jc)::value;       // position j carries n2 = j = ja + RA*jb
    constexpr int ja = j % RA, jb = j / RA;
    if constexpr (ja > 0) z[j] = cmulc(z[j], wa[ja]);
    if constexpr (jb > 0) z[j] = cmulc(z[j], wb[jb]);
  });

  exchange<R, RA, RB, false>(z, smem, p, u);

  // ---- I2 and store (spectre.py:553 keeps rows < min(N, n_fft)) --------------------------------------
  fftA<RA, RB, true>(z);
  {
    char* ob = reinterpret_cast<char*>(a.out) + ((size_t)b * a.out_sb + (size_t)ct * 16) * ES_OUT;
    const uint32_t ooff = (uint32_t)(((long long)u * a.out_sn + 2 * p) * ES_OUT);
    static_for<0, R>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int n1 = (j / RB) + RA * (j % RB);
      char* ptr = ob + (size_t)n1 * R * a.out_sn * ES_OUT + ooff;
      bool ok = true;
      if constexpr (GENERAL) ok = (u + R * n1) < a.N_in;
      if (ok) {
        if constexpr (OUT_BF16) {
          *reinterpret_cast<uint32_t*>(ptr) = f32_to_bf16_rne(z[j].x) | (f32_to_bf16_rne(z[j].y) << 16);
        } else {
          *reinterpret_cast<float2*>(ptr) = z[j];
        }
      }
    });
  }
}

// host-side launcher for one R (defined in regtile_r*.hip so the heavy kernels compile in parallel)
template <int R>
hipError_t launch_regtile(const RegtileArgs& a, bool in_bf16, bool out_bf16, bool general, hipStream_t stream);

#define SFFT_DEFINE_REGTILE_LAUNCHER(RR)                                                                     \
  template <>                                                                                                \
  hipError_t launch_regtile<RR>(const RegtileArgs& a, bool in_bf16, bool out_bf16, bool general,             \
                                hipStream_t stream) {                                                        \
    const dim3 grid(a.n_tiles), block(regtile_threads<RR>());                                                \
    const size_t lds = regtile_lds_bytes<RR>();                                                              \
    auto go = [&](auto kern) -> hipError_t {                                                                 \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);              \
      if (e != hipSuccess) return e;                                                                         \
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                                 \
      return hipGetLastError();                                                                              \
    };                                                                                                       \
    const int key = (in_bf16 ? 4 : 0) | (out_bf16 ? 2 : 0) | (general ? 1 : 0);                              \
    switch (key) {                                                                                           \
      case 0: return go(spectre_mix_regtile<RR, false, false, false>);                                       \
      case 1: return go(spectre_mix_regtile<RR, false, false, true>);                                        \
      case 2: return go(spectre_mix_regtile<RR, false, true, false>);                                        \
      case 3: return go(spectre_mix_regtile<RR, false, true, true>);                                         \
      case 4: return go(spectre_mix_regtile<RR, true, false, false>);                                        \
      case 5: return go(spectre_mix_regtile<RR, true, false, true>);                                         \
      case 6: return go(spectre_mix_regtile<RR, true, true, false>);                                         \
      default: return go(spectre_mix_regtile<RR, true, true, true>);                                         \
    }                                                                                                        \
  }

}  // namespace sfft
