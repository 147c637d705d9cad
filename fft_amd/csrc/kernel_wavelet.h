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
// HBM-bound streaming work.  Two forms: level 0 in registers (`spectre_wavelet_refine_regs_kernel`: v read once, 64-byte row segments at
// N = 4096) for whole aligned tiles, and the general LDS-only form below (any power-of-two N <= 32768, ragged channel counts, v read twice).
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
  int tiles, gang;            // channel tiles per batch element (the launch is tiles * B workgroups, one-dimensional); tiles per 128-byte line
  long long v_sb, v_sn, out_sb, out_sn, ref_sb, ref_sn;   // element strides
};

// Workgroup -> (batch element, channel tile).  The hardware deals workgroup ids round-robin over the 8 XCDs, each with an L2 of its own; the
// `gang` tiles that share a row's 128-byte line must meet in ONE L2 and close in time, or every line is fetched once per XCD that holds a piece
// of it and its pieces are written back separately (tools/fold_lab.hip: a half line costs a line then).  A gang's members are therefore the ids
// x, x + 8, x + 16, ... (congruent mod 8 = one XCD, neighbours in dispatch order), and consecutive gangs go round the XCDs, so that the tiles of
// a switched-on batch element spread over the whole chip whatever the mask looks like (one XCD per element made the default rate slower).
__device__ inline void wv_tile(const WaveletArgs& a, int& b, int& ct) {
  const int n = gridDim.x, id = blockIdx.x, span = 8 * a.gang;
  int w = id;
  if (n % span == 0) {
    const int grp = id / span, r = id - grp * span;
    w = a.gang * (grp * 8 + (r & 7)) + (r >> 3);
  }
  b = w / a.tiles;
  ct = w - b * a.tiles;
}

constexpr int kWaveletThreads = 512;
constexpr int kWaveletMaxFloats = 32768;                       // N * C, 128 KiB of the 160-KiB LDS

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

// I/O of a tile in bursts: kWaveletBurst requests per lane are issued before the first one is consumed (a plain loop would wait for every
// load in turn: 64 round trips per tile).  VEC = 4: 16-byte (fp32) / 8-byte (bf16) accesses, needs C >= 4, whole tiles (D % C == 0) and strides
// / bases that keep them aligned — the host checks; VEC = 1 is the general path.
constexpr int kWaveletBurst = 8;

template <bool BF16, int VEC> struct WvPack { float x[VEC]; };

template <bool BF16, int VEC> __device__ inline WvPack<BF16, VEC> wv_load_pack(const void* p, long long i) {
  WvPack<BF16, VEC> r;
  if constexpr (VEC == 1) r.x[0] = wv_load<BF16>(p, i);
  else if constexpr (BF16) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + i);
    r.x[0] = __uint_as_float(u.x << 16); r.x[1] = __uint_as_float(u.x & 0xffff0000u);
    r.x[2] = __uint_as_float(u.y << 16); r.x[3] = __uint_as_float(u.y & 0xffff0000u);
  } else {
    const float4 f = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
    r.x[0] = f.x; r.x[1] = f.y; r.x[2] = f.z; r.x[3] = f.w;
  }
  return r;
}
__device__ inline unsigned wv_bf16_bits(float x) {
  const unsigned u = __float_as_uint(x);
  return ((x != x) ? 0x7fc00000u : u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
template <bool BF16, int VEC> __device__ inline void wv_store_pack(void* p, long long i, const WvPack<BF16, VEC>& r) {
  if constexpr (VEC == 1) wv_store<BF16>(p, i, r.x[0]);
  else if constexpr (BF16)
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p) + i) =
        make_uint2(wv_bf16_bits(r.x[0]) | (wv_bf16_bits(r.x[1]) << 16), wv_bf16_bits(r.x[2]) | (wv_bf16_bits(r.x[3]) << 16));
  else *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i) = make_float4(r.x[0], r.x[1], r.x[2], r.x[3]);
}

// The round trip of an [n][C] array in the LDS, in place (see the header comment); every thread of the workgroup calls it.
// Work items (pair j, channel c) are dealt round-robin, kWaveletChunk of them in registers at a time.  Analysis: a pair touches only its own
// two cells, so chunks need no barrier between them.  Synthesis: pair j writes the cell pair j + 1 reads (y[2j+1] lands on hi[j+1]) — the
// chunks run from the HIGHEST pairs down, each one `read all, barrier, write all, barrier`, so that a cell is read before a lower pair
// overwrites it; the one exception is the wrap (pair 0's hi lives in row L - 1, which the top pair overwrites): it is read first.
constexpr int kWaveletChunk = 8;

template <int T = kWaveletThreads>
__device__ inline void wv_pyramid(float* wx, const int N, const int levels, const int cs, const int tid) {
  const int C = 1 << cs;
  const float s = 0.70710678118654752440f;
  for (int l = 0; l < levels; ++l) {                           // analysis, in place
    const int L = N >> l, work = (L >> 1) << cs;
    for (int i0 = tid; i0 < work; i0 += T * kWaveletChunk) {
      float xa[kWaveletChunk], xb[kWaveletChunk];
#pragma unroll
      for (int u = 0; u < kWaveletChunk; ++u) {
        const int i = i0 + u * T, j = i >> cs, c = i & (C - 1);
        if (i < work) {
          xa[u] = wx[((((2 * j - 1) & (L - 1)) << l) << cs) + c];
          xb[u] = wx[(((2 * j) << l) << cs) + c];
        }
      }
#pragma unroll
      for (int u = 0; u < kWaveletChunk; ++u) {
        const int i = i0 + u * T, j = i >> cs, c = i & (C - 1);
        if (i < work) {
          wx[(((2 * j) << l) << cs) + c] = (xa[u] + xb[u]) * s;
          wx[((((2 * j - 1) & (L - 1)) << l) << cs) + c] = (xb[u] - xa[u]) * s;
        }
      }
    }
    __syncthreads();
  }
  for (int l = levels - 1; l >= 0; --l) {                      // synthesis
    const int L = N >> l, work = (L >> 1) << cs;
    const int chunks = (work + T * kWaveletChunk - 1) / (T * kWaveletChunk);
    float hi_wrap = 0.f;
    if (chunks > 1 && tid < C) hi_wrap = wx[(((L - 1) << l) << cs) + tid];      // item (pair 0, channel tid) is this thread's first item
    for (int ch = chunks - 1; ch >= 0; --ch) {
      const int i0 = tid + ch * T * kWaveletChunk;
      float lo[kWaveletChunk], hi[kWaveletChunk];
#pragma unroll
      for (int u = 0; u < kWaveletChunk; ++u) {
        const int i = i0 + u * T, j = i >> cs, c = i & (C - 1);
        if (i < work) {
          lo[u] = wx[(((2 * j) << l) << cs) + c];
          hi[u] = wx[((((2 * j - 1) & (L - 1)) << l) << cs) + c];
        }
      }
      if (chunks > 1 && ch == 0 && tid < C) hi[0] = hi_wrap;
      __syncthreads();
#pragma unroll
      for (int u = 0; u < kWaveletChunk; ++u) {
        const int i = i0 + u * T, j = i >> cs, c = i & (C - 1);
        if (i < work) {
          wx[(((2 * j) << l) << cs) + c] = (lo[u] + hi[u]) * s;
          wx[(((2 * j + 1) << l) << cs) + c] = (lo[u] - hi[u]) * s;
        }
      }
      __syncthreads();
    }
  }
}

template <bool BF16, int VEC>
__global__ __launch_bounds__(kWaveletThreads) void spectre_wavelet_refine_kernel(WaveletArgs a) {
  extern __shared__ float wx[];                                // [N][C]
  const int tid = threadIdx.x, C = a.C, N = a.N;
  int b, ct;
  wv_tile(a, b, ct);
  const int c0 = ct * C, cw = min(C, a.D - c0);
  const int cs = __ffs(C) - 1;                                 // C is a power of two
  const int total = N << cs, packs = total / VEC;              // (VEC = 4: C >= 4, so a pack never crosses a row)
  const long long vb = (long long)b * a.v_sb + c0, ob = (long long)b * a.out_sb + c0, rb0 = (long long)b * a.ref_sb + c0;
  const bool on = a.mask[b] != 0;
  if (!on && a.out == a.v) return;
  // tile -> LDS (switched-off elements of an out-of-place call: tile -> out, a plain copy).  Every load is issued unconditionally — lanes
  // beyond the tile re-read its first element — because hipcc puts a predicated load behind a wait of its own (fft_amd/isa_lint.py).
  const int c = (tid * VEC) & (C - 1);                         // the same channels in every round: kWaveletThreads * VEC is a multiple of C
  const bool c_ok = c < cw;
  for (int p0 = tid; p0 < packs; p0 += kWaveletThreads * kWaveletBurst) {
    WvPack<BF16, VEC> r[kWaveletBurst];
#pragma unroll
    for (int k = 0; k < kWaveletBurst; ++k) {
      const int i = (p0 + k * kWaveletThreads) * VEC, n = i >> cs;
      r[k] = wv_load_pack<BF16, VEC>(a.v, (i < total && c_ok) ? vb + (long long)n * a.v_sn + c : vb);
    }
#pragma unroll
    for (int k = 0; k < kWaveletBurst; ++k) {
      const int i = (p0 + k * kWaveletThreads) * VEC, n = i >> cs;
      if (i < total) {
        if (on) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) wx[i + e] = c_ok ? r[k].x[e] : 0.f;       // (channels beyond D: zeros, never stored)
        } else if (c_ok) wv_store_pack<BF16, VEC>(a.out, ob + (long long)n * a.out_sn + c, r[k]);
      }
    }
  }
  if (!on) return;
  __syncthreads();
  wv_pyramid(wx, N, a.levels, cs, tid);
  // out = v + R(v) * gate (v a second time: from the L2 where it survived), vref = R(v)
  float g[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) g[e] = a.gate[(long long)b * a.D + (c_ok ? c0 + c + e : c0)];
  for (int p0 = tid; p0 < packs; p0 += kWaveletThreads * kWaveletBurst) {
    WvPack<BF16, VEC> r[kWaveletBurst];
#pragma unroll
    for (int k = 0; k < kWaveletBurst; ++k) {
      const int i = (p0 + k * kWaveletThreads) * VEC, n = i >> cs;
      r[k] = wv_load_pack<BF16, VEC>(a.v, (i < total && c_ok) ? vb + (long long)n * a.v_sn + c : vb);
    }
#pragma unroll
    for (int k = 0; k < kWaveletBurst; ++k) {
      const int i = (p0 + k * kWaveletThreads) * VEC, n = i >> cs;
      if (i < total && c_ok) {
        WvPack<BF16, VEC> rt;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          rt.x[e] = wx[i + e];
          r[k].x[e] = fmaf(rt.x[e], g[e], r[k].x[e]);
        }
        wv_store_pack<BF16, VEC>(a.out, ob + (long long)n * a.out_sn + c, r[k]);
        if (a.vref) wv_store_pack<BF16, VEC>(a.vref, rb0 + (long long)n * a.ref_sn + c, rt);
      }
    }
  }
}

// Level 0 in registers (the fast form; the host takes it for whole, aligned tiles with 256 <= N <= 16384).  A thread owns P consecutive
// pairs (x[2j-1], x[2j]) of one channel QUAD — 2P rows x 16 bytes, loaded in one burst —, forms lo / hi there and sends only the
// approximation band to the LDS: half the image, so the tile is TWICE as wide (16 channels = 64-byte row segments at N = 4096, a 256-KiB
// tile) and the deeper levels move half the bytes.  The detail band never leaves the registers, v is read from HBM ONCE (out = v + ... adds
// to the registers' copy; the one row per thread that belongs to the next thread's first pair is fetched a second time), and the output
// rows (2j, 2j+1) leave as two 16-byte stores per pair.
template <bool BF16, int P, int T = kWaveletThreads>
__global__ __launch_bounds__(T) void spectre_wavelet_refine_regs_kernel(WaveletArgs a) {
  extern __shared__ float wx[];                                // approximation band [N / 2][C]
  const int tid = threadIdx.x, C = a.C, N = a.N;
  int b, ct;
  wv_tile(a, b, ct);
  if (!a.mask[b]) {
    if (a.out != a.v) {                                        // out of place: the switched-off element's tile is copied
      const int c0o = ct * C, cso = __ffs(C) - 1, packs = (N << cso) >> 2, co = (tid * 4) & (C - 1);
      const long long vb = (long long)b * a.v_sb + c0o + co, ob = (long long)b * a.out_sb + c0o + co;
      for (int p0 = tid; p0 < packs; p0 += T * kWaveletBurst) {
        WvPack<BF16, 4> r[kWaveletBurst];
#pragma unroll
        for (int k = 0; k < kWaveletBurst; ++k) {
          const int i = (p0 + k * T) * 4;
          r[k] = wv_load_pack<BF16, 4>(a.v, i < (N << cso) ? vb + (long long)(i >> cso) * a.v_sn : vb);
        }
#pragma unroll
        for (int k = 0; k < kWaveletBurst; ++k) {
          const int i = (p0 + k * T) * 4;
          if (i < (N << cso)) wv_store_pack<BF16, 4>(a.out, ob + (long long)(i >> cso) * a.out_sn, r[k]);
        }
      }
    }
    return;
  }
  const int cs = __ffs(C) - 1, Q = C >> 2;                     // quads per row
  const int tq = tid & (Q - 1), tb = tid / Q;                  // (Q is a power of two)
  const int c = 4 * tq, j0 = tb * P;
  const long long vb = (long long)b * a.v_sb + ct * C + c, ob = (long long)b * a.out_sb + ct * C + c,
                  rb0 = (long long)b * a.ref_sb + ct * C + c;
  const float s = 0.70710678118654752440f;
  WvPack<BF16, 4> xa[P], xb[P], edge;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const int rb = 2 * (j0 + q), ra = (rb - 1) & (N - 1);
    xa[q] = wv_load_pack<BF16, 4>(a.v, vb + (long long)ra * a.v_sn);
    xb[q] = wv_load_pack<BF16, 4>(a.v, vb + (long long)rb * a.v_sn);
  }
  edge = wv_load_pack<BF16, 4>(a.v, vb + (long long)(2 * (j0 + P) - 1) * a.v_sn);     // x[2j+1] of the last pair = the next thread's first row
  float g[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) g[e] = a.gate[(long long)b * a.D + ct * C + c + e];
#pragma unroll
  for (int q = 0; q < P; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) wx[((j0 + q) << cs) + c + e] = (xa[q].x[e] + xb[q].x[e]) * s;
  __syncthreads();
  wv_pyramid<T>(wx, N >> 1, a.levels - 1, cs, tid);
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const int rb = 2 * (j0 + q);
    WvPack<BF16, 4> y0, y1, o0, o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = wx[((j0 + q) << cs) + c + e], hi = (xb[q].x[e] - xa[q].x[e]) * s;
      y0.x[e] = (lo + hi) * s;
      y1.x[e] = (lo - hi) * s;
      o0.x[e] = fmaf(y0.x[e], g[e], xb[q].x[e]);
      o1.x[e] = fmaf(y1.x[e], g[e], q + 1 < P ? xa[q + 1 < P ? q + 1 : q].x[e] : edge.x[e]);
    }
    wv_store_pack<BF16, 4>(a.out, ob + (long long)rb * a.out_sn, o0);
    wv_store_pack<BF16, 4>(a.out, ob + (long long)(rb + 1) * a.out_sn, o1);
    if (a.vref) {
      wv_store_pack<BF16, 4>(a.vref, rb0 + (long long)rb * a.ref_sn, y0);
      wv_store_pack<BF16, 4>(a.vref, rb0 + (long long)(rb + 1) * a.ref_sn, y1);
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
  const int cblocks = (a.D + 63) >> 6;                       // (one-dimensional launch: cblocks * B workgroups)
  const int b = blockIdx.x / cblocks, lane = threadIdx.x & 63, ph = threadIdx.x >> 6, c = (blockIdx.x - b * cblocks) * 64 + lane;
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
