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

// ---- backward of the fused tail (training path of row N2) -------------------------------------------------------------------
// gate = phase * modReLU(r, bias), r = cubic resample of the anchors.  With G = dL/dgate in PyTorch's convention (real and imaginary
// part = derivative w.r.t. the real and imaginary part), per (b, g, k):
//   dy = G conj(phase),   dphase += G conj(y)
//   dot = <dy, r>,  on = (|r| + bias > 0):   dbias += on * dot / den,   den = sqrt(|r|^2 + eps^2)
//   dr = scale dy + [on * (1/den - act |r| / den^3) * dot / |r|] r          (second term 0 at r = 0, as torch.abs' gradient)
// and the resample is linear: danchor plane (s, j) = sum over the bins whose 4 taps touch j (clamped) of weight * dr component,
// with the reference's plane pairing (see spectre_gate_producer).  Everything is summed in a fixed order (no atomics).
struct GateBwdArgs {
  const float2* anchors;   // (B, G, K)
  const float* bias;       // (G * F)
  const float2* phase;     // (F) or (B, F) or nullptr
  const float2* dgate;     // (B, G, F) upstream gradient
  float2* dr;              // workspace (B, G, F): gradient w.r.t. the resampled anchors
  float2* danchors;        // (B, G, K) out
  float* dbias;            // (G * F) out
  float2* dphase;          // same shape as phase, or nullptr
  int B, G, K, F;
  long long phase_sb;
  float eps;
};

__device__ __forceinline__ void gate_taps(int k, int K, int F, int& i0, float (&w)[4]) {
  const float step = 2.0f / (float)(F - 1);
  const int half = F / 2;
  float xs = (k < half) ? -1.0f + step * (float)k : 1.0f - step * (float)(F - k - 1);
  if (F == 1) xs = -1.0f;
  const float ix = ((xs + 1.f) / 2.f) * (float)(K - 1);
  const float fl = floorf(ix);
  const float t = ix - fl;
  i0 = (int)fl;
  constexpr float A = -0.75f;
  w[0] = cubic_conv2(t + 1.f, A); w[1] = cubic_conv1(t, A); w[2] = cubic_conv1(1.f - t, A); w[3] = cubic_conv2((1.f - t) + 1.f, A);
}

// one thread per (g, k), loop over the batch: dbias and a shared dphase need no cross-thread reduction over b
__global__ void __launch_bounds__(256) spectre_gate_bwd_elem(const GateBwdArgs a) {
  const int gk = blockIdx.x * blockDim.x + threadIdx.x;
  if (gk >= a.G * a.F) return;
  const int g = gk / a.F, k = gk - g * a.F;
  int i0; float w[4];
  gate_taps(k, a.K, a.F, i0, w);
  const float bias = a.bias[gk];
  float db = 0.f;
  float2 dph = make_float2(0.f, 0.f);
  for (int b = 0; b < a.B; ++b) {
    const float* af = reinterpret_cast<const float*>(a.anchors) + (size_t)b * a.G * a.K * 2;
    auto plane = [&](int s, int j) { j = j < 0 ? 0 : (j > a.K - 1 ? a.K - 1 : j); return af[((size_t)(s % a.G) * a.K + j) * 2 + (s / a.G)]; };
    const int s0 = 2 * g, s1 = 2 * g + 1;
    const float re = plane(s0, i0 - 1) * w[0] + plane(s0, i0) * w[1] + plane(s0, i0 + 1) * w[2] + plane(s0, i0 + 2) * w[3];
    const float im = plane(s1, i0 - 1) * w[0] + plane(s1, i0) * w[1] + plane(s1, i0 + 1) * w[2] + plane(s1, i0 + 2) * w[3];
    const float mag = hypotf(re, im);
    const float den = sqrtf(mag * mag + a.eps * a.eps);
    const float act = fmaxf(mag + bias, 0.f);
    const bool on = (mag + bias) > 0.f;
    const float scale = act / den;
    const size_t idx = ((size_t)b * a.G + g) * a.F + k;
    const float2 G_ = a.dgate[idx];
    float2 dy = G_;
    if (a.phase) {
      const float2 ph = a.phase[b * a.phase_sb + k];
      dy = make_float2(G_.x * ph.x + G_.y * ph.y, G_.y * ph.x - G_.x * ph.y);            // G conj(phase)
      if (a.dphase) {
        const float2 y = make_float2(re * scale, im * scale);
        const float2 c = make_float2(G_.x * y.x + G_.y * y.y, G_.y * y.x - G_.x * y.y);   // G conj(y)
        if (a.phase_sb) { float2* d = a.dphase + (size_t)b * a.phase_sb + k; atomicAdd(&d->x, c.x); atomicAdd(&d->y, c.y); }   // (B, F): sum over the G groups
        else { dph.x += c.x; dph.y += c.y; }
      }
    }
    const float dot = dy.x * re + dy.y * im;
    float coef = 0.f;
    if (on) {
      db += dot / den;
      if (mag > 0.f) coef = (1.f / den - act * mag / (den * den * den)) * dot / mag;
    }
    a.dr[idx] = make_float2(scale * dy.x + coef * re, scale * dy.y + coef * im);
  }
  a.dbias[gk] = db;
  if (a.phase && a.dphase && !a.phase_sb) { atomicAdd(&a.dphase[k].x, dph.x); atomicAdd(&a.dphase[k].y, dph.y); }   // (F): sum over the G groups
}

// one wave per (b, plane s, anchor j): the bins whose taps can touch j form a short contiguous range
__global__ void __launch_bounds__(64) spectre_gate_bwd_anchors(const GateBwdArgs a) {
  const int j = blockIdx.x, s = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  const int g = s >> 1, comp = s & 1;                       // output group / component this plane feeds (s0 = 2g -> re, s1 = 2g+1 -> im)
  // i0(k) = floor(k (K-1) / (F-1)) up to rounding: taps i0-1 .. i0+2 -> i0 in [j-2, j+1]; one bin of margin on both sides
  const float inv = (a.K > 1) ? (float)(a.F - 1) / (float)(a.K - 1) : 0.f;
  int k_lo = (int)floorf((float)(j - 2) * inv) - 1, k_hi = (int)ceilf((float)(j + 2) * inv) + 1;
  if (a.K == 1) { k_lo = 0; k_hi = a.F - 1; }
  k_lo = k_lo < 0 ? 0 : k_lo; k_hi = k_hi > a.F - 1 ? a.F - 1 : k_hi;
  float acc = 0.f;
  for (int k = k_lo + lane; k <= k_hi; k += 64) {
    int i0; float w[4];
    gate_taps(k, a.K, a.F, i0, w);
    const float2 d = a.dr[((size_t)b * a.G + g) * a.F + k];
    const float val = comp ? d.y : d.x;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      int idx = i0 - 1 + jj;
      idx = idx < 0 ? 0 : (idx > a.K - 1 ? a.K - 1 : idx);
      if (idx == j) acc += w[jj] * val;
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) reinterpret_cast<float*>(a.danchors)[(((size_t)b * a.G + (s % a.G)) * a.K + j) * 2 + (s / a.G)] = acc;
}

}  // namespace sfft
