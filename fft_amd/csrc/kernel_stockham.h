// kernel_stockham.h — general spectral mix: mixed-radix Stockham auto-sort FFT staged in LDS, any n_fft.
//
// Covers every shape the register-resident kernel does not: non-square / non-power-of-two n_fft
// (radices 2,3,4,5,7,8,11,13 — e.g. 3000 = 8*3*5*5*5), prime or rough lengths through Bluestein's
// chirp-z (power-of-two convolution length M >= 2 n_fft - 1), odd group width d_g (no channel pairing),
// unaligned views, partial channel tiles.  Same math as kernel_regtile.h: a "slot" is one complex sequence
// z = x_c + i x_{c+1} (two channels of one gate group) or z = x_c (solo mode, d_g odd), filtered by the
// Hermitian extension of gate[b, c / d_g, :]  (spectre.py:506, :542-553).
//
// A workgroup keeps P slots x L points in two LDS buffers ([n][p], p fastest so global accesses of
// neighbouring lanes fall in one row segment) and ping-pongs the Stockham passes between them; the inverse
// transform reuses the forward passes through conj(DFT(conj(.))).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_regs.h"
#include "kernel_regtile.h"   // f32_to_bf16_rne

namespace sfft {

constexpr int kStockhamMaxThreads = 1024;   // block size is chosen at launch (256..1024): a workgroup that takes most of the
                                            // LDS is alone on its CU and needs all the waves it can get to hide LDS latency
constexpr int kMaxPasses = 16;

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
  int radix[kMaxPasses];
  const float2* tw;      // exp(-2 pi i m / L), m < L
  // Bluestein (M = L > N)
  int bluestein;
  const float2* chirp;   // w[n] = exp(-i pi n^2 / N), n < N
  const float2* bhat;    // DFT_M of the wrapped conj chirp
};

// ---- small DFTs on a register array (natural order in/out) ---------------------------------------
template <int R> __device__ __forceinline__ void small_dft(float2 (&v)[R], const float2* tw, int L);

template <> __device__ __forceinline__ void small_dft<2>(float2 (&v)[2], const float2*, int) { bfly2<false>(v[0], v[1]); }
template <> __device__ __forceinline__ void small_dft<4>(float2 (&v)[4], const float2*, int) { bfly4<false>(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void small_dft<8>(float2 (&v)[8], const float2*, int) {
  bfly8<false>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void small_dft<3>(float2 (&v)[3], const float2*, int) {
  // W3 = -1/2 - i sqrt(3)/2
  constexpr float s = 0.86602540378443865f;
  const float2 t = cadd(v[1], v[2]);
  const float2 d = csub(v[1], v[2]);
  const float2 m = make_float2(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
  const float2 r = make_float2(s * d.y, -s * d.x);     // -i * s * d
  v[0] = cadd(v[0], t);
  v[1] = cadd(m, r);
  v[2] = csub(m, r);
}
template <> __device__ __forceinline__ void small_dft<5>(float2 (&v)[5], const float2*, int) {
  constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;   // cos(2pi/5), cos(4pi/5)
  constexpr float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;    // sin(2pi/5), sin(4pi/5)
  const float2 a1 = cadd(v[1], v[4]), b1 = csub(v[1], v[4]);
  const float2 a2 = cadd(v[2], v[3]), b2 = csub(v[2], v[3]);
  const float2 m1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
  const float2 m2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
  // -i * (s1 b1 + s2 b2)  and  -i * (s2 b1 - s1 b2)
  const float2 q1 = make_float2(s1 * b1.y + s2 * b2.y, -(s1 * b1.x + s2 * b2.x));
  const float2 q2 = make_float2(s2 * b1.y - s1 * b2.y, -(s2 * b1.x - s1 * b2.x));
  v[0] = make_float2(v[0].x + a1.x + a2.x, v[0].y + a1.y + a2.y);
  v[1] = cadd(m1, q1);
  v[4] = csub(m1, q1);
  v[2] = cadd(m2, q2);
  v[3] = csub(m2, q2);
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

// one Stockham pass of radix R over P interleaved sequences of length L
template <int R>
__device__ __forceinline__ void stockham_pass(const float2* __restrict__ in, float2* __restrict__ out, int L, int P, int Ns,
                                              const float2* __restrict__ tw) {
  const int nb = L / R;                 // butterflies per sequence
  const int tstep = L / (Ns * R);       // W_(Ns R)^(k r) = tw[k r tstep]
  for (int wi = threadIdx.x; wi < nb * P; wi += blockDim.x) {
    const int j = wi / P, pp = wi - j * P;
    const int k = j % Ns;
    float2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[r] = in[(j + r * nb) * P + pp];
      if (r > 0 && Ns > 1) v[r] = cmul(v[r], tw[k * r * tstep]);
    }
    small_dft<R>(v, tw, L);
    const int j0 = (j - k) * R + k;     // (j / Ns) * Ns * R + k
#pragma unroll
    for (int m = 0; m < R; ++m) out[(j0 + m * Ns) * P + pp] = v[m];
  }
}

// forward DFT_L of the P sequences in `cur`; returns the buffer holding the natural-order result
__device__ __forceinline__ float2* stockham_fft(float2* cur, float2* oth, int L, int P, const int* radix, int n_pass,
                                                const float2* tw) {
  int Ns = 1;
  for (int s = 0; s < n_pass; ++s) {
    const int R = radix[s];
    switch (R) {
      case 2: stockham_pass<2>(cur, oth, L, P, Ns, tw); break;
      case 3: stockham_pass<3>(cur, oth, L, P, Ns, tw); break;
      case 4: stockham_pass<4>(cur, oth, L, P, Ns, tw); break;
      case 5: stockham_pass<5>(cur, oth, L, P, Ns, tw); break;
      case 7: stockham_pass<7>(cur, oth, L, P, Ns, tw); break;
      case 8: stockham_pass<8>(cur, oth, L, P, Ns, tw); break;
      case 11: stockham_pass<11>(cur, oth, L, P, Ns, tw); break;
      default: stockham_pass<13>(cur, oth, L, P, Ns, tw); break;
    }
    Ns *= R;
    __syncthreads();
    float2* t = cur; cur = oth; oth = t;
  }
  return cur;
}

// forward DFT_N of the data in `cur` (entries n < N valid), smooth N or Bluestein; *other is updated
__device__ __forceinline__ float2* dft_n(float2* cur, float2** other, const StockhamArgs& a) {
  float2* oth = *other;
  const int P = a.P;
  if (!a.bluestein) {
    float2* r = stockham_fft(cur, oth, a.L, P, a.radix, a.n_pass, a.tw);
    *other = (r == cur) ? oth : cur;
    return r;
  }
  const int N = a.N, M = a.L;
  // a[n] = x[n] w[n] (n < N), 0 (N <= n < M)
  for (int i = threadIdx.x; i < M * P; i += blockDim.x) {
    const int n = i / P;
    cur[i] = (n < N) ? cmul(cur[i], a.chirp[n]) : make_float2(0.f, 0.f);
  }
  __syncthreads();
  float2* r = stockham_fft(cur, oth, M, P, a.radix, a.n_pass, a.tw);
  float2* o = (r == cur) ? oth : cur;
  // circular convolution with the conj chirp: multiply spectra; inverse FFT as conj(FFT(conj(.))) / M
  for (int i = threadIdx.x; i < M * P; i += blockDim.x) {
    const float2 t = cmul(r[i], a.bhat[i / P]);
    r[i] = make_float2(t.x, -t.y);
  }
  __syncthreads();
  float2* r2 = stockham_fft(r, o, M, P, a.radix, a.n_pass, a.tw);
  float2* o2 = (r2 == r) ? o : r;
  const float inv_m = 1.0f / (float)M;
  for (int i = threadIdx.x; i < N * P; i += blockDim.x) {
    const float2 t = make_float2(r2[i].x * inv_m, -r2[i].y * inv_m);
    r2[i] = cmul(t, a.chirp[i / P]);
  }
  __syncthreads();
  *other = o2;
  return r2;
}

__global__ void __launch_bounds__(kStockhamMaxThreads) spectre_mix_stockham(const StockhamArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float2* buf0 = reinterpret_cast<float2*>(smem_raw);
  float2* buf1 = buf0 + (size_t)a.L * a.P;
  const int P = a.P, N = a.N;
  const int b = blockIdx.x / a.groups_per_batch;
  const int slot0 = (blockIdx.x - b * a.groups_per_batch) * P;
  const int n_out = a.N_in < N ? a.N_in : N;

  // ---- load (zero-pad / truncate to n_fft, spectre.py:506) ----------------------------------------
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
    buf0[i] = val;
  }
  __syncthreads();

  float2* oth = buf1;
  float2* X = dft_n(buf0, &oth, a);

  // ---- filter: Y[k] = Gf[k] X[k] + Mf[k]; stored conjugated for the conj-trick inverse ------------
  const bool even = (N % 2) == 0;
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
      if (upper) g.y = -g.y;
      if (edge) g.y = 0.f;                             // irfft ignores Im(DC), Im(Nyquist)
      y = cmul(X[i], g);
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
    X[i] = make_float2(y.x, -y.y);
  }
  __syncthreads();

  float2* Yt = dft_n(X, &oth, a);   // = conj(N * y)

  // ---- store rows < min(N_in, n_fft) (spectre.py:553) -----------------------------------------------
  const float inv_n = 1.0f / (float)N;
  for (int i = threadIdx.x; i < n_out * P; i += blockDim.x) {
    const int n = i / P, pp = i - n * P;
    const int slot = slot0 + pp;
    if (slot >= a.S) continue;
    const int c0 = a.solo ? slot : 2 * slot;
    const float yr = Yt[i].x * inv_n, yi = -Yt[i].y * inv_n;
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

}  // namespace sfft
