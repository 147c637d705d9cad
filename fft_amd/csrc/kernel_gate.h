// kernel_gate.h — gate producer tail fused into one launch:
//   cubic resample of the K anchors of every (batch, group) to F bins   /root/reference/spectre.py:38-61, :518-524
//   -> complex modReLU with one bias per (group, bin)                   spectre.py:109-121, :530-531
//   -> optional positional phase                                        spectre.py:534-536
// The reference runs this as ~12 ATen launches on a (B, G, F) tensor (grid_sample + abs/sqrt/relu/div/mul); they are
// launch-latency bound (~0.1 ms together), which is 5-20 % of the fused spectral mix they feed.
//
// The resample restates F.grid_sample(mode="bicubic", padding_mode="border", align_corners=True) on a height-1 image
// sampled at y = 0, x = linspace(-1, 1, F): the y taps have weights (0, 1, 0, 0) exactly, so it is the 1-D cubic
// convolution (A = -0.75) over the anchors with clamped indices; x is formed with the same float32 operations
// (linspace's two-sided formula, then ((x + 1) / 2) * (K - 1)) so that floor() and the fractional part agree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfft {

struct GateArgs {
  const float2* anchors;   // (B, G, K)
  const float* bias;       // (G * F)
  const float2* phase;     // (F) or (B, F), or nullptr
  float2* gate;            // (B, G, F)
  int B, G, K, F;
  long long phase_sb;      // 0 or F
  float eps;
  int decode_m, decode_n;  // decode_m != 0: multiply by exp(1j * 2*pi * k * decode_m / decode_n) evaluated as the reference does in
                           // float32 (spectre.py:593-596); decode_m = t - t % n_fft
};

__device__ __forceinline__ float cubic_conv1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic_conv2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__global__ void __launch_bounds__(256) spectre_gate_producer(const GateArgs a) {
  const long long total = (long long)a.B * a.G * a.F;
  const float step = 2.0f / (float)(a.F - 1);              // torch.linspace(-1, 1, F): (end - start) / (steps - 1)
  const int half = a.F / 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % a.F);
    const long long bg = i / a.F;
    const int g = (int)(bg % a.G);
    const long long b = bg / a.G;
    float xs = (k < half) ? -1.0f + step * (float)k : 1.0f - step * (float)(a.F - k - 1);
    if (a.F == 1) xs = -1.0f;
    const float ix = ((xs + 1.f) / 2.f) * (float)(a.K - 1);  // grid_sampler_unnormalize, align_corners=True
    const float fl = floorf(ix);
    const float t = ix - fl;
    const int i0 = (int)fl;
    constexpr float A = -0.75f;
    const float w0 = cubic_conv2(t + 1.f, A), w1 = cubic_conv1(t, A);
    const float w2 = cubic_conv1(1.f - t, A), w3 = cubic_conv2((1.f - t) + 1.f, A);
    // Which anchor plane feeds which output plane: the reference stacks (real, imag) on dim 1 — (B, 2, G, K) — and then
    // RESHAPES that to (B*G, 2, 1, K) (spectre.py:43), so row g' of a batch element takes its "real" channel from flat
    // plane 2g' and its "imag" channel from flat plane 2g'+1 of the [part][group] block: plane s = (part s / G, group
    // s % G).  A drop-in has to reproduce that pairing (for G = 1 it is the identity).
    const float* af = reinterpret_cast<const float*>(a.anchors) + (size_t)b * a.G * a.K * 2;
    auto plane = [&](int s, int j) { j = j < 0 ? 0 : (j > a.K - 1 ? a.K - 1 : j);            // padding_mode="border"
                                     return af[((size_t)(s % a.G) * a.K + j) * 2 + (s / a.G)]; };
    const int s0 = 2 * g, s1 = 2 * g + 1;
    float re = plane(s0, i0 - 1) * w0 + plane(s0, i0) * w1 + plane(s0, i0 + 1) * w2 + plane(s0, i0 + 2) * w3;
    float im = plane(s1, i0 - 1) * w0 + plane(s1, i0) * w1 + plane(s1, i0 + 1) * w2 + plane(s1, i0 + 2) * w3;
    // modReLU: z * relu(|z| + b) / sqrt(|z|^2 + eps^2)
    const float mag = hypotf(re, im);
    const float scale = fmaxf(mag + a.bias[(size_t)g * a.F + k], 0.f) / sqrtf(mag * mag + a.eps * a.eps);
    re *= scale; im *= scale;
    if (a.phase) {
      const float2 ph = a.phase[b * a.phase_sb + k];
      const float r2 = re * ph.x - im * ph.y;
      im = re * ph.y + im * ph.x;
      re = r2;
    }
    if (a.decode_m != 0) {
      // `1j * 2 * math.pi * k * (t - j) / N` on a complex64 tensor: ((float32(2 pi) * k) * m) * float32(1 / N) — ATen's
      // complex division multiplies by the rounded reciprocal of the real divisor — then cos / sin
      const float x = ((6.283185307179586f * (float)k) * (float)a.decode_m) * (1.0f / (float)a.decode_n);
      float sn, cs;
      sincosf(x, &sn, &cs);
      const float r2 = re * cs - im * sn;
      im = re * sn + im * cs;
      re = r2;
    }
    a.gate[i] = make_float2(re, im);
  }
}

}  // namespace sfft
