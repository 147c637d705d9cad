// kernel_gate_grad_twopass.h — gate gradient for transform lengths whose two LDS slots (V and dOut side by side) do not fit
// the CU (n_fft = 12288, 16384 and Bluestein lengths above 8192): the spectra X = rfft(V) and R = rfft(dOut) are produced by
// the half-spectrum kernel (spectre_rfft_stockham, one LDS slot per channel pair) into a scratch buffer, and this kernel
// forms  dgate[b,g,k] = (w_k / n) * sum_{c in group g} conj(X[b,k,c]) * R[b,k,c],  w_k = 2 (1 at DC and Nyquist), imaginary part
// dropped at DC / Nyquist (they do not reach the output: spectre.py:551) — what autograd derives through spectre.py:506,:542-553.
// Slower than the register-tile gradients (two extra passes over the spectra); it exists so that spectre_mix_bwd never refuses
// a length spectre_mix_fwd accepts.
#pragma once
#include <hip/hip_runtime.h>

namespace sfft {

// grid (F, G, Bc), one wave per (bin, group, batch element); X, R: (Bc, F, D) complex64
__global__ void __launch_bounds__(64) spectre_gate_grad_reduce(const float2* __restrict__ X, const float2* __restrict__ R,
                                                               float2* __restrict__ dgate, int F, int D, int G, int d_g, int n) {
  const int k = blockIdx.x, g = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  const size_t row = ((size_t)b * F + k) * D + (size_t)g * d_g;
  float2 acc = make_float2(0.f, 0.f);
  for (int c = lane; c < d_g; c += 64) {
    const float2 x = X[row + c], r = R[row + c];
    acc.x += x.x * r.x + x.y * r.y;      // conj(x) * r
    acc.y += x.x * r.y - x.y * r.x;
  }
  for (int off = 32; off > 0; off >>= 1) {
    acc.x += __shfl_down(acc.x, off, 64);
    acc.y += __shfl_down(acc.y, off, 64);
  }
  if (lane == 0) {
    const bool edge = (k == 0) || ((n % 2 == 0) && k == n / 2);
    const float w = (edge ? 1.0f : 2.0f) / (float)n;
    dgate[((size_t)b * G + g) * F + k] = make_float2(acc.x * w, edge ? 0.f : acc.y * w);
  }
}

}  // namespace sfft
