// kernel_stockham.h — general spectral mix: mixed-radix Stockham auto-sort FFT staged in LDS, any n_fft.
//
// Covers every shape the register-resident kernel does not: non-power-of-two n_fft (radices 2,3,4,5,7,8,11,13
// and the merged 15, 16, 25 — e.g. 3000 = 8*15*25), prime or rough lengths through Bluestein's chirp-z
// (power-of-two convolution length M >= 2 n_fft - 1), odd group width d_g (no channel pairing), unaligned
// views, channel counts that are not a multiple of 16, n_fft = 8192.  Same math as kernel_regtile.h: a "slot" is
// one complex sequence z = x_c + i x_{c+1} (two channels of one gate group) or z = x_c (solo mode, d_g odd),
// filtered by the Hermitian extension of gate[b, c / d_g, :]  (spectre.py:506, :542-553).
//
// A workgroup keeps P slots x L points in ONE LDS buffer ([n][p], p fastest so global accesses of neighbouring
// lanes fall in one row segment).  A pass reads every butterfly's inputs into registers, synchronises, and writes
// the outputs back to the same buffer at their Stockham (auto-sort) positions: no ping-pong buffer, so twice as
// many slots fit (wider row segments), at the price of a second barrier per pass.  The inverse transform reuses the
// forward passes through conj(DFT(conj(.))).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_regs.h"
#include "kernel_regtile.h"   // f32_to_bf16_rne

namespace sfft {

constexpr int kStockhamMaxThreads = 1024;   // a workgroup that takes most of the LDS is alone on its CU and needs
                                            // all the waves it can get to hide LDS latency
constexpr int kMaxPasses = 16;

// butterflies one thread may hold in registers during a pass of radix R (all inputs of a pass are read before any
// output is written): about 24 complex values, at least one butterfly
__host__ __device__ constexpr int stockham_kmax(int R) { return R >= 24 ? 1 : (R >= 13 ? 2 : 24 / R); }

struct StockhamArgs {
  const void* v;
  const float2* gate;
  const float* mem;
  void* out;
  int B, N_in, N, D, G, d_g, F;
  int P;                 // slots per workgroup
  int S;                 // slots per batch element (D/2 paired, D solo)
  int solo;              // 1: one channel per slot (imaginary input zero, imaginary output dropped)
  int groups_per_batch;  // ceil(S / P)
  int in_bf16, out_bf16;
  long long v_sb, v_sn, out_sb, out_sn;
  // transform of length L (= N, or M for Bluestein)
  int L;
  int n_pass;
  unsigned long long radix_packed[2];   // 8 bits per pass (a kernarg array indexed at run time would be copied to scratch)
  const float2* tw;      // exp(-2 pi i m / L), m < L
  // Bluestein (M = L > N)
  int bluestein;
  const float2* chirp;   // w[n] = exp(-i pi n^2 / N), n < N
  const float2* bhat;    // DFT_M of the wrapped conj chirp
  int conj_gate;         // 1: filter with conj(gate) (adjoint w.r.t. v: dV = mix(dOut, conj(gate)))
  // gate-gradient kernel only
  const void* dout;      // (B, N_out, D), same dtype as v
  long long dout_sb, dout_sn;
  float2* ws;            // (B, G, N) complex accumulator, zeroed by the host
};

constexpr double kCos15[15] = {1, 0.91354545764260087, 0.66913060635885824, 0.30901699437494745, -0.10452846326765333, -0.49999999999999978, -0.80901699437494734, -0.97814760073380569, -0.97814760073380569, -0.80901699437494756, -0.50000000000000044, -0.10452846326765423, 0.30901699437494723, 0.66913060635885846, 0.91354545764260098};
constexpr double kSin15[15] = {0, 0.40673664307580015, 0.74314482547739413, 0.95105651629515353, 0.9945218953682734, 0.86602540378443871, 0.58778525229247325, 0.20791169081775931, -0.20791169081775907, -0.58778525229247303, -0.86602540378443837, -0.99452189536827329, -0.95105651629515364, -0.74314482547739402, -0.40673664307580015};
constexpr double kCos25[25] = {1, 0.96858316112863108, 0.87630668004386358, 0.72896862742141155, 0.53582679497899655, 0.30901699437494745, 0.062790519529313527, -0.1873813145857246, -0.42577929156507272, -0.63742398974868975, -0.80901699437494734, -0.92977648588825135, -0.99211470131447776, -0.99211470131447788, -0.92977648588825146, -0.80901699437494778, -0.63742398974868952, -0.42577929156507216, -0.18738131458572463, 0.062790519529312833, 0.30901699437494723, 0.53582679497899677, 0.72896862742141122, 0.87630668004386314, 0.96858316112863097};
constexpr double kSin25[25] = {0, 0.24868988716485479, 0.48175367410171532, 0.68454710592868862, 0.84432792550201508, 0.95105651629515353, 0.99802672842827156, 0.98228725072868872, 0.90482705246601947, 0.77051324277578925, 0.58778525229247325, 0.36812455268467814, 0.12533323356430454, -0.12533323356430429, -0.36812455268467792, -0.58778525229247269, -0.77051324277578936, -0.9048270524660198, -0.98228725072868872, -0.99802672842827156, -0.95105651629515364, -0.84432792550201496, -0.68454710592868895, -0.4817536741017161, -0.24868988716485535};

// ---- small DFTs on a register array: natural order in; output index m is left at position dft_pos<R>(m)
// (identity except for the two-factor 15/16/25, which skip the final un-permute to save R complex registers)
template <int R> __host__ __device__ constexpr int dft_pos(int m) {
  return R == 16 ? 4 * (m % 4) + m / 4 : R == 15 ? 5 * (m % 3) + m / 3 : R == 25 ? 5 * (m % 5) + m / 5 : m;
}
template <int R> __device__ __forceinline__ void small_dft(float2 (&v)[R], const float2* tw, int L);

template <> __device__ __forceinline__ void small_dft<2>(float2 (&v)[2], const float2*, int) { bfly2<false>(v[0], v[1]); }
template <> __device__ __forceinline__ void small_dft<4>(float2 (&v)[4], const float2*, int) { bfly4<false>(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void small_dft<8>(float2 (&v)[8], const float2*, int) {
  bfly8<false>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void dft3(float2& v0, float2& v1, float2& v2) {
  constexpr float s = 0.86602540378443865f;            // W3 = -1/2 - i sqrt(3)/2
  const float2 t = cadd(v1, v2);
  const float2 d = csub(v1, v2);
  const float2 m = make_float2(v0.x - 0.5f * t.x, v0.y - 0.5f * t.y);
  const float2 r = make_float2(s * d.y, -s * d.x);     // -i * s * d
  v0 = cadd(v0, t);
  v1 = cadd(m, r);
  v2 = csub(m, r);
}
__device__ __forceinline__ void dft5(float2& v0, float2& v1, float2& v2, float2& v3, float2& v4) {
  constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;   // cos(2pi/5), cos(4pi/5)
  constexpr float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;    // sin(2pi/5), sin(4pi/5)
  const float2 a1 = cadd(v1, v4), b1 = csub(v1, v4);
  const float2 a2 = cadd(v2, v3), b2 = csub(v2, v3);
  const float2 m1 = make_float2(v0.x + c1 * a1.x + c2 * a2.x, v0.y + c1 * a1.y + c2 * a2.y);
  const float2 m2 = make_float2(v0.x + c2 * a1.x + c1 * a2.x, v0.y + c2 * a1.y + c1 * a2.y);
  // -i * (s1 b1 + s2 b2)  and  -i * (s2 b1 - s1 b2)
  const float2 q1 = make_float2(s1 * b1.y + s2 * b2.y, -(s1 * b1.x + s2 * b2.x));
  const float2 q2 = make_float2(s2 * b1.y - s1 * b2.y, -(s2 * b1.x - s1 * b2.x));
  v0 = make_float2(v0.x + a1.x + a2.x, v0.y + a1.y + a2.y);
  v1 = cadd(m1, q1);
  v4 = csub(m1, q1);
  v2 = cadd(m2, q2);
  v3 = csub(m2, q2);
}
template <> __device__ __forceinline__ void small_dft<3>(float2 (&v)[3], const float2*, int) { dft3(v[0], v[1], v[2]); }
template <> __device__ __forceinline__ void small_dft<5>(float2 (&v)[5], const float2*, int) { dft5(v[0], v[1], v[2], v[3], v[4]); }

// 16 = 4 x 4 through the type-A in-register transform, then back to natural order
template <> __device__ __forceinline__ void small_dft<16>(float2 (&v)[16], const float2*, int) {
  fftA<4, 4, false>(v);                    // output k = ka + 4*kb sits at position 4*ka + kb (dft_pos<16>)
}
// 15 = 3 x 5 and 25 = 5 x 5: q = RB*q1 + q0, k = ka + RA*kb (same split as fft_regs.h type A), compile-time twiddles
template <int M> __device__ __forceinline__ float2 twid15(float2 a) {
  if constexpr (M % 15 == 0) return a;
  constexpr float c = (float)kCos15[M % 15], s = (float)kSin15[M % 15];
  return make_float2(a.x * c + a.y * s, a.y * c - a.x * s);
}
template <int M> __device__ __forceinline__ float2 twid25(float2 a) {
  if constexpr (M % 25 == 0) return a;
  constexpr float c = (float)kCos25[M % 25], s = (float)kSin25[M % 25];
  return make_float2(a.x * c + a.y * s, a.y * c - a.x * s);
}
template <> __device__ __forceinline__ void small_dft<15>(float2 (&v)[15], const float2*, int) {
  // RA = 3 (over q1, stride 5), RB = 5
  static_for<0, 5>([&](auto q0c) {
    constexpr int q0 = decltype(q0c)::value;
    dft3(v[q0], v[5 + q0], v[10 + q0]);                               // ka at 5*ka + q0
    v[5 + q0] = twid15<q0 * 1>(v[5 + q0]);
    v[10 + q0] = twid15<q0 * 2>(v[10 + q0]);
  });
  static_for<0, 3>([&](auto kac) {
    constexpr int ka = decltype(kac)::value;
    dft5(v[5 * ka], v[5 * ka + 1], v[5 * ka + 2], v[5 * ka + 3], v[5 * ka + 4]);   // kb at 5*ka + kb
  });
}
template <> __device__ __forceinline__ void small_dft<25>(float2 (&v)[25], const float2*, int) {
  static_for<0, 5>([&](auto q0c) {
    constexpr int q0 = decltype(q0c)::value;
    dft5(v[q0], v[5 + q0], v[10 + q0], v[15 + q0], v[20 + q0]);
    v[5 + q0] = twid25<q0 * 1>(v[5 + q0]);
    v[10 + q0] = twid25<q0 * 2>(v[10 + q0]);
    v[15 + q0] = twid25<q0 * 3>(v[15 + q0]);
    v[20 + q0] = twid25<q0 * 4>(v[20 + q0]);
  });
  static_for<0, 5>([&](auto kac) {
    constexpr int ka = decltype(kac)::value;
    dft5(v[5 * ka], v[5 * ka + 1], v[5 * ka + 2], v[5 * ka + 3], v[5 * ka + 4]);
  });
}
// odd primes 7, 11, 13: direct O(R^2) sum with W_R^(r m) read from the length-L table (R | L)
template <int R> __device__ __forceinline__ void small_dft(float2 (&v)[R], const float2* tw, int L) {
  float2 o[R];
  const int step = L / R;
#pragma unroll
  for (int m = 0; m < R; ++m) {
    float2 acc = v[0];
#pragma unroll
    for (int r = 1; r < R; ++r) acc = cadd(acc, cmul(v[r], tw[((r * m) % R) * step]));
    o[m] = acc;
  }
#pragma unroll
  for (int m = 0; m < R; ++m) v[m] = o[m];
}

// one in-place Stockham pass of radix R over P interleaved sequences of length L (host guarantees
// (L/R)*P <= blockDim.x * stockham_kmax(R), so every butterfly is register-resident between the two barriers)
template <int R>
__device__ __attribute__((noinline)) void stockham_pass(float2* __restrict__ buf, int L, int P, int Ns, const float2* __restrict__ tw) {
  constexpr int KM = stockham_kmax(R);
  const int nb = L / R;                 // butterflies per sequence
  const int tstep = L / (Ns * R);       // W_(Ns R)^(k r) = tw[k r tstep]
  const int total = nb * P;
  float2 v[KM][R];
#pragma unroll
  for (int q = 0; q < KM; ++q) {
    const int wi = threadIdx.x + q * blockDim.x;
    if (wi < total) {
      const int j = wi / P, pp = wi - j * P;
      const int k = j % Ns;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        v[q][r] = buf[(j + r * nb) * P + pp];
        if (r > 0 && Ns > 1) v[q][r] = cmul(v[q][r], tw[k * r * tstep]);
      }
    }
  }
  __syncthreads();                      // every input of the pass is in registers: the buffer may be overwritten
#pragma unroll
  for (int q = 0; q < KM; ++q) {
    const int wi = threadIdx.x + q * blockDim.x;
    if (wi < total) {
      const int j = wi / P, pp = wi - j * P;
      const int k = j % Ns;
      small_dft<R>(v[q], tw, L);
      const int j0 = (j - k) * R + k;   // (j / Ns) * Ns * R + k
#pragma unroll
      for (int m = 0; m < R; ++m) buf[(j0 + m * Ns) * P + pp] = v[q][dft_pos<R>(m)];
    }
  }
  __syncthreads();
}

// forward DFT_L of the P sequences in buf, natural order in and out
__device__ __forceinline__ void stockham_fft(float2* buf, int L, int P, unsigned long long r0, unsigned long long r1, int n_pass,
                                             const float2* tw) {
  int Ns = 1;
  for (int s = 0; s < n_pass; ++s) {
    const int R = (int)(((s < 8 ? r0 : r1) >> (8 * (s & 7))) & 0xffu);
    switch (R) {
      case 2: stockham_pass<2>(buf, L, P, Ns, tw); break;
      case 3: stockham_pass<3>(buf, L, P, Ns, tw); break;
      case 4: stockham_pass<4>(buf, L, P, Ns, tw); break;
      case 5: stockham_pass<5>(buf, L, P, Ns, tw); break;
      case 7: stockham_pass<7>(buf, L, P, Ns, tw); break;
      case 8: stockham_pass<8>(buf, L, P, Ns, tw); break;
      case 11: stockham_pass<11>(buf, L, P, Ns, tw); break;
      case 13: stockham_pass<13>(buf, L, P, Ns, tw); break;
      case 15: stockham_pass<15>(buf, L, P, Ns, tw); break;
      case 16: stockham_pass<16>(buf, L, P, Ns, tw); break;
      default: stockham_pass<25>(buf, L, P, Ns, tw); break;
    }
    Ns *= R;
  }
}

// forward DFT_N of the data in buf (entries n < N valid), smooth N or Bluestein, in place
__device__ __forceinline__ void dft_n(float2* buf, const StockhamArgs& a) {
  const int P = a.P;
  if (!a.bluestein) {
    stockham_fft(buf, a.L, P, a.radix_packed[0], a.radix_packed[1], a.n_pass, a.tw);
    return;
  }
  const int N = a.N, M = a.L;
  // a[n] = x[n] w[n] (n < N), 0 (N <= n < M)
  for (int i = threadIdx.x; i < M * P; i += blockDim.x) {
    const int n = i / P;
    buf[i] = (n < N) ? cmul(buf[i], a.chirp[n]) : make_float2(0.f, 0.f);
  }
  __syncthreads();
  stockham_fft(buf, M, P, a.radix_packed[0], a.radix_packed[1], a.n_pass, a.tw);
  // circular convolution with the conj chirp: multiply spectra; inverse FFT as conj(FFT(conj(.))) / M
  for (int i = threadIdx.x; i < M * P; i += blockDim.x) {
    const float2 t = cmul(buf[i], a.bhat[i / P]);
    buf[i] = make_float2(t.x, -t.y);
  }
  __syncthreads();
  stockham_fft(buf, M, P, a.radix_packed[0], a.radix_packed[1], a.n_pass, a.tw);
  const float inv_m = 1.0f / (float)M;
  for (int i = threadIdx.x; i < N * P; i += blockDim.x) {
    const float2 t = make_float2(buf[i].x * inv_m, -buf[i].y * inv_m);
    buf[i] = cmul(t, a.chirp[i / P]);
  }
  __syncthreads();
}

__global__ void __launch_bounds__(kStockhamMaxThreads) spectre_mix_stockham(const StockhamArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float2* buf = reinterpret_cast<float2*>(smem_raw);
  const int P = a.P, N = a.N;
  // neighbouring channel groups share 128-byte lines: keep them on one XCD (same L2), as the register kernel does —
  // in launch order the 48-byte segments of N = 3000 were fetched 3.4x over (rocprofv3 FETCH_SIZE)
  const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
  const int b = wg / a.groups_per_batch;
  const int slot0 = (wg - b * a.groups_per_batch) * P;
  const int n_out = a.N_in < N ? a.N_in : N;

  // ---- load (zero-pad / truncate to n_fft, spectre.py:506) ----------------------------------------
  // (unrolled so that several global loads are in flight per thread: one tile is ~18 of these iterations)
#pragma unroll 4
  for (int i = threadIdx.x; i < N * P; i += blockDim.x) {
    const int n = i / P, pp = i - n * P;
    const int slot = slot0 + pp;
    float2 val = make_float2(0.f, 0.f);
    if (slot < a.S && n < a.N_in) {
      const int c0 = a.solo ? slot : 2 * slot;
      const size_t off = (size_t)b * a.v_sb + (size_t)n * a.v_sn + c0;
      if (a.in_bf16) {
        const uint16_t* pv = reinterpret_cast<const uint16_t*>(a.v) + off;
        val.x = __uint_as_float((uint32_t)pv[0] << 16);
        if (!a.solo) val.y = __uint_as_float((uint32_t)pv[1] << 16);
      } else {
        const float* pv = reinterpret_cast<const float*>(a.v) + off;
        val.x = pv[0];
        if (!a.solo) val.y = pv[1];
      }
    }
    buf[i] = val;
  }
  __syncthreads();

  dft_n(buf, a);

  // ---- filter: Y[k] = Gf[k] X[k] + Mf[k]; stored conjugated for the conj-trick inverse ------------
  const bool even = (N % 2) == 0;
#pragma unroll 4
  for (int i = threadIdx.x; i < N * P; i += blockDim.x) {
    const int k = i / P, pp = i - k * P;
    const int slot = slot0 + pp;
    float2 y = make_float2(0.f, 0.f);
    if (slot < a.S) {
      const int c0 = a.solo ? slot : 2 * slot;
      const int grp = c0 / a.d_g;
      const bool upper = 2 * k > N;                    // k > N/2: conj(g[N-k])
      const int idx = upper ? N - k : k;
      const bool edge = (k == 0) || (even && 2 * k == N);
      float2 g = a.gate[((size_t)b * a.G + grp) * a.F + idx];
      if (a.conj_gate) g.y = -g.y;
      if (upper) g.y = -g.y;
      if (edge) g.y = 0.f;                             // irfft ignores Im(DC), Im(Nyquist)
      y = cmul(buf[i], g);
      if (a.mem != nullptr) {
        const float* mp = a.mem + ((size_t)idx * a.D + c0) * 2;
        const float m0r = mp[0], m0i = mp[1];
        float m1r = 0.f, m1i = 0.f;
        if (!a.solo) { m1r = mp[2]; m1i = mp[3]; }
        if (edge)       { y.x += m0r;        y.y += m1r; }
        else if (upper) { y.x += m0r + m1i;  y.y += m1r - m0i; }
        else            { y.x += m0r - m1i;  y.y += m0i + m1r; }
      }
    }
    buf[i] = make_float2(y.x, -y.y);
  }
  __syncthreads();

  dft_n(buf, a);   // = conj(N * y)

  // ---- store rows < min(N_in, n_fft) (spectre.py:553) -----------------------------------------------
  const float inv_n = 1.0f / (float)N;
#pragma unroll 4
  for (int i = threadIdx.x; i < n_out * P; i += blockDim.x) {
    const int n = i / P, pp = i - n * P;
    const int slot = slot0 + pp;
    if (slot >= a.S) continue;
    const int c0 = a.solo ? slot : 2 * slot;
    const float yr = buf[i].x * inv_n, yi = -buf[i].y * inv_n;
    const size_t off = (size_t)b * a.out_sb + (size_t)n * a.out_sn + c0;
    if (a.out_bf16) {
      uint16_t* po = reinterpret_cast<uint16_t*>(a.out) + off;
      po[0] = (uint16_t)f32_to_bf16_rne(yr);
      if (!a.solo) po[1] = (uint16_t)f32_to_bf16_rne(yi);
    } else {
      float* po = reinterpret_cast<float*>(a.out) + off;
      po[0] = yr;
      if (!a.solo) po[1] = yi;
    }
  }
}

// ---- gradient w.r.t. the gate ----------------------------------------------------------
// dgate[b,g,k] = (w_k / N) * sum_{c in g} conj(X_c[k]) R_c[k],  X = rfft(v), R = rfft(dOut zero-padded), w_k = 2
// (1 at DC / Nyquist).  With two channels per complex sequence (Z = X_c + i X_{c+1}, W = R_c + i R_{c+1})
//   P[k] = conj(Z[k]) W[k] = S[k] + i T[k],   P[N-k] = conj(S[k]) + i conj(T[k])   =>   S[k] = (P[k] + conj(P[N-k])) / 2
// so the kernel only accumulates P over the channel pairs of a group for ALL N bins (atomics into ws), and the k <-> N-k
// pairing happens once per (b, g) in spectre_gate_grad_finish.  Slots 0..P-1 of the LDS buffer hold the v sequences,
// slots P..2P-1 the dOut sequences; one Stockham run transforms all 2P.  The host picks P | (slots per group).
__global__ void __launch_bounds__(kStockhamMaxThreads) spectre_gate_grad_stockham(const StockhamArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float2* buf = reinterpret_cast<float2*>(smem_raw);
  const int P = a.P, P2 = 2 * a.P, N = a.N;
  const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
  const int b = wg / a.groups_per_batch;
  const int slot0 = (wg - b * a.groups_per_batch) * P;
  const int n_dy = a.N_in < N ? a.N_in : N;        // rows of dOut (= rows the forward kept)

#pragma unroll 2
  for (int i = threadIdx.x; i < N * P2; i += blockDim.x) {
    const int n = i / P2, pp = i - n * P2;
    const bool is_dy = pp >= P;
    const int slot = slot0 + (is_dy ? pp - P : pp);
    float2 val = make_float2(0.f, 0.f);
    if (slot < a.S && n < (is_dy ? n_dy : a.N_in)) {
      const int c0 = a.solo ? slot : 2 * slot;
      const size_t off = is_dy ? (size_t)b * a.dout_sb + (size_t)n * a.dout_sn + c0
                               : (size_t)b * a.v_sb + (size_t)n * a.v_sn + c0;
      const void* base = is_dy ? a.dout : a.v;
      if (a.in_bf16) {
        const uint16_t* pv = reinterpret_cast<const uint16_t*>(base) + off;
        val.x = __uint_as_float((uint32_t)pv[0] << 16);
        if (!a.solo) val.y = __uint_as_float((uint32_t)pv[1] << 16);
      } else {
        const float* pv = reinterpret_cast<const float*>(base) + off;
        val.x = pv[0];
        if (!a.solo) val.y = pv[1];
      }
    }
    buf[i] = val;
  }
  __syncthreads();

  StockhamArgs a2 = a;          // the transform sees 2P interleaved sequences
  a2.P = P2;
  dft_n(buf, a2);

  const int c_first = a.solo ? slot0 : 2 * slot0;
  float2* wsp = a.ws + ((size_t)b * a.G + c_first / a.d_g) * N;
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    float2 acc = make_float2(0.f, 0.f);
    for (int pp = 0; pp < P; ++pp) {
      if (slot0 + pp < a.S) acc = cadd(acc, cmulc(buf[k * P2 + P + pp], buf[k * P2 + pp]));   // W * conj(Z)
    }
    atomicAdd(&wsp[k].x, acc.x);
    atomicAdd(&wsp[k].y, acc.y);
  }
}

// dgate[b,g,k] = (P[k] + conj(P[N-k])) / N for 0 < k < N/2 (w_k = 2, times 1/2), Re(P[k]) / N at DC / Nyquist
__global__ void __launch_bounds__(256) spectre_gate_grad_finish(const float2* __restrict__ ws, float2* __restrict__ dgate, int BG, int N, int F) {
  const long long total = (long long)BG * F;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % F);
    const long long bg = i / F;
    const float2 pk = ws[bg * N + k];
    const float2 pm = ws[bg * N + (k == 0 ? 0 : N - k)];
    const float inv_n = 1.0f / (float)N;
    const bool edge = (k == 0) || (2 * k == N);
    dgate[i] = edge ? make_float2(pk.x * inv_n, 0.f) : make_float2((pk.x + pm.x) * inv_n, (pk.y - pm.y) * inv_n);
  }
}

}  // namespace sfft
