// kernel_decode.h — prefill and single-token decode.
//
//   spectre_rfft_stockham     PrefixFFTCache.prefill: rfft of the zero-padded prompt        /root/reference/spectre.py:769-783
//   spectre_decode_step       PrefixFFTCache.decode_step (sliding-window spectrum update)   spectre.py:786-814
//                             fused with SpectreHead.decode_step's filter multiply          spectre.py:597-603
//                             and pruned_irfft_single (one output row of the inverse)       spectre.py:614-655
//   spectre_decode_finish     sum of the per-chunk partial results, 1/n
//
// The reference forms every phase from float32 products of large arguments (omega * k * t with t growing without
// bound); the kernel evaluates them in the same order, without fused multiply-adds, so that the spectrum it maintains
// tracks the reference's over many steps.  The Nyquist term of pruned_irfft_single is multiplied by (-1)^pos a second
// time in the reference (:650, on top of cos(pi pos) inside `contrib`); a drop-in reproduces that.
#pragma once
#include "kernel_stockham.h"

namespace sfft {

// ---- prefill: half spectrum of every channel, (B, F, D) complex64 -----------------------------------------------------
// Uses the slot machinery of kernel_stockham.h (a.out = the spectrum); paired slots are untangled on the way out:
//   X_c[k] = (Z[k] + conj(Z[N-k])) / 2,   X_{c+1}[k] = (Z[k] - conj(Z[N-k])) / (2i).
__global__ void __launch_bounds__(kStockhamMaxThreads) spectre_rfft_stockham(const StockhamArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float2* buf = reinterpret_cast<float2*>(smem_raw);
  const int P = a.P, N = a.N;
  const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
  const int b = wg / a.groups_per_batch;
  const int slot0 = (wg - b * a.groups_per_batch) * P;
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
  float2* spec = reinterpret_cast<float2*>(a.out) + (size_t)b * a.F * a.D;
  for (int i = threadIdx.x; i < a.F * P; i += blockDim.x) {
    const int k = i / P, pp = i - k * P;
    const int slot = slot0 + pp;
    if (slot >= a.S) continue;
    const float2 zk = buf[k * P + pp];
    if (a.solo) {
      spec[(size_t)k * a.D + slot] = zk;
    } else {
      const float2 zm = buf[(k == 0 ? 0 : N - k) * P + pp];
      const float2 x0 = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
      const float2 x1 = make_float2(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
      spec[(size_t)k * a.D + 2 * slot] = x0;
      spec[(size_t)k * a.D + 2 * slot + 1] = x1;
    }
  }
}

// ---- decode ---------------------------------------------------------------------------------------------------------------
struct DecodeArgs {
  float2* prefix;        // (F, d) complex64, updated in place
  const float* v_old;    // (d): V_buf[j] before this step
  const float* v_new;    // (d)
  const float2* gate;    // (G, F) complex64 (after modReLU and phase), or nullptr: state update only
  float* partial;        // (chunks, d)
  int n, F, d, d_g;
  int t, j, evict;       // absolute step, ring position t % n, t >= n
  int chunk;             // bins per workgroup
};

constexpr int kDecodeChunk = 64;

#pragma clang fp contract(off)
__global__ void __launch_bounds__(256) spectre_decode_step(const DecodeArgs a) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lane;
  const int k0 = blockIdx.x * a.chunk;
  const int k1 = min(k0 + a.chunk, a.F);
  const bool live = c < a.d;
  const float omega = (float)(-2.0 * 3.14159265358979323846 / (double)a.n);     // self.omega (spectre.py:767) as float32
  const float two_pi = (float)(2.0 * 3.14159265358979323846);
  const float vo = live ? a.v_old[c] : 0.f, vn = live ? a.v_new[c] : 0.f;
  const float2* grow = a.gate ? a.gate + (size_t)((live ? c : 0) / a.d_g) * a.F : nullptr;
  const int pos = a.j;
  float acc = 0.f;
  for (int k = k0 + w; k < k1; k += 4) {
    const float kf = (float)k;
    float2 x = live ? a.prefix[(size_t)k * a.d + c] : make_float2(0.f, 0.f);
    if (a.evict) {                                   // prefix -= exp(1j * omega * k * j) * v_old      (:798-801)
      const float arg = (omega * kf) * (float)a.j;
      float s, co;
      sincosf(arg, &s, &co);
      x.x = x.x - co * vo;
      x.y = x.y - s * vo;
    }
    {                                                // prefix += exp(1j * omega * k * t) * v_t        (:803-805)
      const float arg = (omega * kf) * (float)a.t;
      float s, co;
      sincosf(arg, &s, &co);
      x.x = x.x + co * vn;
      x.y = x.y + s * vn;
    }
    if (live) a.prefix[(size_t)k * a.d + c] = x;
    if (grow) {                                      // gate * prefix, then one output of the inverse (:603, :614-655)
      const float2 g = grow[k];
      const float mr = g.x * x.x - g.y * x.y, mi = g.x * x.y + g.y * x.x;
      const float ph = ((two_pi * kf) * (float)pos) / (float)a.n;
      float s, co;
      sincosf(ph, &s, &co);
      float contrib = mr * co - mi * s;
      if (k == 0) {
      } else if ((a.n % 2 == 0) && k == a.F - 1) {
        contrib = (pos & 1) ? -contrib : contrib;    // "* ((-1) ** pos)" (:650)
      } else {
        contrib = 2.f * contrib;
      }
      acc += contrib;
    }
  }
  if (!a.gate) return;
  red[w][lane] = acc;
  __syncthreads();
  if (w == 0 && live) a.partial[(size_t)blockIdx.x * a.d + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// sum of the per-chunk partials, 1/n; optionally also the ring-buffer writes V_buf[j] = v_t, Q_buf[j] = q_t (:807-810),
// which must follow the main kernel's read of the evicted row
__global__ void __launch_bounds__(256) spectre_decode_finish(const float* __restrict__ partial, float* __restrict__ out, int chunks, int d,
                                                             int n, float* v_row, const float* v_new, float* q_row, const float* q_new) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  float s = 0.f;
  for (int i = 0; i < chunks; ++i) s += partial[(size_t)i * d + c];
  out[c] = s / (float)n;
  if (v_row) v_row[c] = v_new[c];
  if (q_row) q_row[c] = q_new[c];
}
#pragma clang fp contract(fast)

// ---- gate descriptor of a decode step (spectre.py:575-580): running query sum -> LayerNorm -> Linear -> GELU -> Linear ----
// One workgroup: d and the MLP are small (d x 256 + 256 x 2BG weights, < 1 MB); a wave per output row, lanes along the
// reduction.  sum_q is updated in place with the reference's own expression (see fft_amd/decode.py on `q_old`).
struct DecodeMlpArgs {
  float* sum_q;            // (d) in/out
  const float* q_t;        // (d)
  const float *ln_w, *ln_b;
  const float *w1, *b1;    // (h1, d), (h1)
  const float *w2, *b2;    // (o, h1), (o)
  float* anchors;          // (o) = (G, K, 2)
  int d, h1, o, n, evict;
  float ln_eps;
};

__device__ __forceinline__ float wave_sum(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ void __launch_bounds__(1024) spectre_decode_mlp(const DecodeMlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* descr = reinterpret_cast<float*>(smem_raw);          // d
  float* hidden = descr + a.d;                                 // h1
  __shared__ float red[16];
  __shared__ float stats[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  // sum_q += q_t - (q_old if evict else 0.0), with q_old read AFTER the ring slot was overwritten in the reference
  float part = 0.f;
  for (int c = tid; c < a.d; c += blockDim.x) {
    const float q = a.q_t[c];
    const float s = a.sum_q[c] + (a.evict ? (q - q) : (q - 0.0f));
    a.sum_q[c] = s;
    const float x = s / (float)a.n;                            // (sum_q / cache.N)
    descr[c] = x;
    part += x;
  }
  part = wave_sum(part);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  if (tid == 0) { float t = 0.f; for (int i = 0; i < nw; ++i) t += red[i]; stats[0] = t / (float)a.d; }
  __syncthreads();
  const float mean = stats[0];
  part = 0.f;
  for (int c = tid; c < a.d; c += blockDim.x) { const float dv = descr[c] - mean; part += dv * dv; }
  part = wave_sum(part);
  __syncthreads();
  if (lane == 0) red[wave] = part;
  __syncthreads();
  if (tid == 0) { float t = 0.f; for (int i = 0; i < nw; ++i) t += red[i]; stats[1] = 1.0f / sqrtf(t / (float)a.d + a.ln_eps); }
  __syncthreads();
  const float rstd = stats[1];
  for (int c = tid; c < a.d; c += blockDim.x) descr[c] = (descr[c] - mean) * rstd * a.ln_w[c] + a.ln_b[c];
  __syncthreads();
  for (int j = wave; j < a.h1; j += nw) {                      // hidden = GELU(W1 descr + b1), exact erf form (nn.GELU())
    const float* w = a.w1 + (size_t)j * a.d;
    float acc = 0.f;
    for (int c = lane; c < a.d; c += 64) acc += w[c] * descr[c];
    acc = wave_sum(acc);
    if (lane == 0) { const float x = acc + a.b1[j]; hidden[j] = 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
  }
  __syncthreads();
  for (int i = wave; i < a.o; i += nw) {
    const float* w = a.w2 + (size_t)i * a.h1;
    float acc = 0.f;
    for (int c = lane; c < a.h1; c += 64) acc += w[c] * hidden[c];
    acc = wave_sum(acc);
    if (lane == 0) a.anchors[i] = acc + a.b2[i];
  }
}

}  // namespace sfft
