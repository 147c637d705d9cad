// kernel_wavelet.h — the Haar round trip of the reference's WaveletRefinement (spectre.py:819-887) as one launch.
//
// What the reference does per batch element that its coin flip switched on (spectre.py:841, :853-872): transpose the (N, d) slab, run the
// multi-level Haar analysis `dwt_decompose` (:288-312: log2(N) levels of `HaarDWT.forward` :190-219 = circular left pad by one, two-tap
// correlation, stride 2) and the synthesis `dwt_reconstruct` (:315-328: `HaarIDWT.forward` :246-272 = two-tap transposed convolution,
// stride 2), then `v + (v_ref.detach() * gate) * on_mask` (:884-886).  Because of the one-sample pad the analysis pairs (x[2j-1], x[2j])
// while the synthesis writes (y[2j], y[2j+1]), so the round trip R is NOT the identity (one level maps 0..7 to 0,7,2,1,4,3,6,5) — it is a
// fixed linear operator along the sequence, and a trained `gate_mlp` has learnt against exactly that operator.  This kernel applies it:
//
//   level l (length L = N >> l, samples at rows i << l):   lo[j] = (x[2j-1] + x[2j]) / sqrt 2,  hi[j] = (x[2j] - x[2j-1]) / sqrt 2
//       in place: lo[j] takes x[2j]'s row, hi[j] takes x[2j-1]'s row (row L-1 for j = 0), so level l+1 finds its input at rows i << (l+1)
//   back up:                                                 y[2j] = (lo'[j] + hi[j]) / sqrt 2,  y[2j+1] = (lo'[j] - hi[j]) / sqrt 2
//       (y[2j+1] lands on the row that still holds hi[j+1]: every level reads all its pairs into registers, barrier, then writes)
//
// Layout: a workgroup owns C channels x all N rows of ONE batch element in the LDS (N * C * 4 bytes <= 128 KiB: C = 8 at N = 4096), reads
// its mask byte first and leaves at once when the element is off — so the launch costs the bytes of the switched-on elements only (10 % at
// the reference's default rate), needs no host-side gather and no device-to-host synchronisation (the reference's `on_mask.any()` is one).
// HBM-bound integer-free streaming work: v is read twice (tile load + the final add, the second read from the L2), out written once.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bfloat16.h>
#include <stdint.h>

namespace sfft {

struct WaveletArgs {
  const void* v;              // (B, N, D) f32 | bf16
  void* out;                  // (B, N, D), may alias v (in place)
  void* vref;                 // optional (B, N, D): R(v) of the switched-on elements (for the gate gradient); rows of the others untouched
  const unsigned char* mask;  // (B) one byte per batch element, non-zero = on
  const float* gate;          // (B, D) f32
  int B, N, D, C, levels;
  long long v_sb, v_sn, out_sb, out_sn, ref_sb, ref_sn;   // element strides
};

constexpr int kWaveletThreads = 512;
constexpr int kWaveletMaxFloats = 32768;                       // N * C, 128 KiB of the 160-KiB LDS
constexpr int kWaveletPairs = kWaveletMaxFloats / 2 / kWaveletThreads;   // pairs a thread holds at level 0 (32)

template <bool BF16> __device__ inline float wv_load(const void* p, long long i) {
  if constexpr (BF16) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(p)[i] << 16);
  else return reinterpret_cast<const float*>(p)[i];
}
template <bool BF16> __device__ inline void wv_store(void* p, long long i, float x) {
  if constexpr (BF16) {                                        // round to nearest even, NaN kept quiet
    unsigned u = __float_as_uint(x);
    u = (x != x) ? 0x7fc00000u : u + 0x7fffu + ((u >> 16) & 1u);
    reinterpret_cast<unsigned short*>(p)[i] = (unsigned short)(u >> 16);
  } else reinterpret_cast<float*>(p)[i] = x;
}

template <bool BF16>
__global__ __launch_bounds__(kWaveletThreads) void spectre_wavelet_refine_kernel(WaveletArgs a) {
  extern __shared__ float wx[];                                // [N][C]
  const int tid = threadIdx.x, b = blockIdx.y, C = a.C, N = a.N;
  const int c0 = blockIdx.x * C, cw = min(C, a.D - c0);
  const int cs = __ffs(C) - 1;                                 // C is a power of two
  const int total = N << cs;
  const long long vb = (long long)b * a.v_sb + c0, ob = (long long)b * a.out_sb + c0;
  if (!a.mask[b]) {
    if (a.out != a.v)
      for (int i = tid; i < total; i += kWaveletThreads) {
        const int n = i >> cs, c = i & (C - 1);
        if (c < cw) wv_store<BF16>(a.out, ob + (long long)n * a.out_sn + c, wv_load<BF16>(a.v, vb + (long long)n * a.v_sn + c));
      }
    return;
  }
  for (int i = tid; i < total; i += kWaveletThreads) {
    const int n = i >> cs, c = i & (C - 1);
    wx[i] = c < cw ? wv_load<BF16>(a.v, vb + (long long)n * a.v_sn + c) : 0.f;
  }
  __syncthreads();
  const float s = 0.70710678118654752440f;
  // analysis, in place
  for (int l = 0; l < a.levels; ++l) {
    const int L = N >> l, work = (L >> 1) << cs;
    for (int i = tid; i < work; i += kWaveletThreads) {
      const int j = i >> cs, c = i & (C - 1);
      const int ra = ((2 * j - 1) & (L - 1)) << l, rb = (2 * j) << l;
      const float xa = wx[(ra << cs) + c], xb = wx[(rb << cs) + c];
      wx[(rb << cs) + c] = (xa + xb) * s;
      wx[(ra << cs) + c] = (xb - xa) * s;
    }
    __syncthreads();
  }
  // synthesis: all pairs of a level into registers, then out again
  for (int l = a.levels - 1; l >= 0; --l) {
    const int L = N >> l, work = (L >> 1) << cs;
    float lo[kWaveletPairs], hi[kWaveletPairs];
#pragma unroll
    for (int k = 0; k < kWaveletPairs; ++k) {
      const int i = tid + k * kWaveletThreads;
      if (i < work) {
        const int j = i >> cs, c = i & (C - 1);
        lo[k] = wx[(((2 * j) << l) << cs) + c];
        hi[k] = wx[((((2 * j - 1) & (L - 1)) << l) << cs) + c];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWaveletPairs; ++k) {
      const int i = tid + k * kWaveletThreads;
      if (i < work) {
        const int j = i >> cs, c = i & (C - 1);
        wx[(((2 * j) << l) << cs) + c] = (lo[k] + hi[k]) * s;
        wx[(((2 * j + 1) << l) << cs) + c] = (lo[k] - hi[k]) * s;
      }
    }
    __syncthreads();
  }
  const long long rb0 = (long long)b * a.ref_sb + c0;
  for (int i = tid; i < total; i += kWaveletThreads) {
    const int n = i >> cs, c = i & (C - 1);
    if (c < cw) {
      const float r = wx[i];
      const float x = wv_load<BF16>(a.v, vb + (long long)n * a.v_sn + c);
      wv_store<BF16>(a.out, ob + (long long)n * a.out_sn + c, x + r * a.gate[(long long)b * a.D + c0 + c]);
      if (a.vref) wv_store<BF16>(a.vref, rb0 + (long long)n * a.ref_sn + c, r);
    }
  }
}

// d/d(gate) of v + (R(v).detach() * gate) * on_mask (spectre.py:884-886): dgate[b, c] = on[b] * sum_n dout[b, n, c] * vref[b, n, c].
// A workgroup = 64 channels x 8 row phases of one batch element; switched-off elements write their zeros and leave.
struct WaveletGradArgs {
  const void* dout;           // (B, N, D) f32 | bf16
  const void* vref;           // (B, N, D) as written by the forward launch
  const unsigned char* mask;
  float* dgate;               // (B, D) f32
  int B, N, D;
  long long d_sb, d_sn, ref_sb, ref_sn;
};

template <bool BF16>
__global__ __launch_bounds__(512) void spectre_wavelet_gate_grad_kernel(WaveletGradArgs a) {
  __shared__ float part[8][64];
  const int b = blockIdx.y, lane = threadIdx.x & 63, ph = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
  const bool live = c < a.D;
  if (!a.mask[b]) {
    if (ph == 0 && live) a.dgate[(long long)b * a.D + c] = 0.f;
    return;
  }
  float acc = 0.f;
  if (live) {
    const long long db = (long long)b * a.d_sb + c, rb = (long long)b * a.ref_sb + c;
#pragma unroll 4
    for (int n = ph; n < a.N; n += 8)
      acc = fmaf(wv_load<BF16>(a.dout, db + (long long)n * a.d_sn), wv_load<BF16>(a.vref, rb + (long long)n * a.ref_sn), acc);
  }
  part[ph][lane] = acc;
  __syncthreads();
  if (ph == 0 && live) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += part[k][lane];
    a.dgate[(long long)b * a.D + c] = t;
  }
}

}  // namespace sfft
