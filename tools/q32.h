// q32.h — round 6 experiment: the n_fft = 4096 spectral mix with SIXTEEN waves per workgroup, 32 points per thread
// (<= 128 VGPRs: four waves per SIMD instead of the two of kernel_regtile64p.h).  Same 256-KiB tile (16 channels x 4096 rows of one
// batch element = 8 packed complex sequences), same path (/root/reference/spectre.py:506, :542-553 in one launch).
//
// 4096 = 128 x 32:  n = n2 + 128 n1 (n1 < 32),  k = k1 + 32 k2 (k2 < 128).
//   rows role  thread (p, n2):     F1 = 32-point transform over n1 in registers (type A), twiddle W_N^(n2 k1), ... I2, loads and stores
//   bins role  thread (p, k1, L):  the 128-point transform over n2 is shared by the FOUR lanes of a quad (L = lane & 3): lane L reads
//              n2 = m + 32 L (m < 32) out of the exchange image, a radix-4 butterfly ACROSS the quad (DPP quad_perm, decimation in
//              frequency:  C_q[m] = sum_L x[m + 32 L] (-i)^(L q)), the twiddle W_128^(m q) from a per-lane table, and a 32-point
//              transform over m in registers leave lane L with the bins k2 = 4 k2'' + q_L, q_L = (0, 2, 1, 3); the inverse runs the same
//              steps backwards and leaves lane L with n2 = m + 32 L again.
//   The signs the cross-lane butterflies leave behind (a lane can only compute own + tau * partner) are absorbed by the per-lane table.
//
// Exchange image (one float plane at a time, like kernel_regtile64p.h): [k1: 32 rows][p: 8 columns of 132 floats][n2: 128 slots with a
// 2-float gap in the middle].  E1: scattered ds_write_b32 by the rows role (bank = 4 p + (n2 & 3): conflict-free), contiguous ds_read_b64
// by the bins role (32 lanes = (L, p): dword pairs 16 L + (L >> 1) + 2 p (mod 32): conflict-free).  E2: the same addresses the other way
// round (contiguous ds_write_b64, scattered ds_read_b32).
//
// Version 0 (this file): ONE tile per workgroup, no software pipeline — measures the instruction stream of the 16-wave form with and
// without its traffic against the 8-wave kernel's (tools/q32_lab.hip).
#pragma once
#include "../fft_amd/csrc/kernel_regtile.h"

namespace sfft {

constexpr int kQ32PS = 132, kQ32RW = 8 * kQ32PS;                  // column / row strides of the image in floats
constexpr int kQ32ImageBytes = 32 * kQ32RW * 4;                   // 135168
constexpr int kQ32GateOff = kQ32ImageBytes;                       // half-spectrum gate, entry k at k + (k >> 5): 2113 float2
constexpr int kQ32TwOff = (kQ32GateOff + 2113 * 8 + 15) & ~15;    // rows-role twiddle bases: 128 x 10 float2
constexpr int kQ32LaneTwOff = kQ32TwOff + 128 * 10 * 8;           // per-lane table: 4 x (32 + 2 pad) float2
constexpr int kQ32LdsTotal = kQ32LaneTwOff + 4 * 34 * 8;
static_assert(kQ32LdsTotal <= 160 * 1024, "LDS budget");

__host__ __device__ constexpr int q32_slot(int n2) { return n2 + 2 * (n2 >> 6); }

template <int CTRL>
__device__ __forceinline__ float q32_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
constexpr int kQuadX1 = 0xB1;    // quad_perm [1,0,3,2]: lane ^ 1
constexpr int kQuadX2 = 0x4E;    // quad_perm [2,3,0,1]: lane ^ 2

__device__ __forceinline__ void q32_lane16_swap(float& x, float& y) {   // x of the odd 16-lane rows <-> y of the even rows
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void q32_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef unsigned int q32_u32x4 __attribute__((ext_vector_type(4)));

// acc + tau * (acc of the lane ^ 1 / lane ^ 2 of the quad): ONE v_fmac_f32 with a DPP operand (hipcc leaves a v_mov_b32_dpp in front of a
// plain v_fmac when the same thing is written with __builtin_amdgcn_update_dpp: 256 extra moves per thread and tile in version 0)
__device__ __forceinline__ float q32_fmac_x1(float acc, float tau) {
  asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(acc), "v"(tau));
  return acc;
}
__device__ __forceinline__ float q32_fmac_x2(float acc, float tau) {
  asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(acc), "v"(tau));
  return acc;
}

template <int BASE, int STRIDE, int CNT>
__device__ __forceinline__ void q32_pin(float2 (&z)[32]) {
  static_assert(CNT == 4 || CNT == 8, "pin 4 or 8 values");
  asm volatile("" : "+v"(z[BASE].x), "+v"(z[BASE].y), "+v"(z[BASE + STRIDE].x), "+v"(z[BASE + STRIDE].y),
                    "+v"(z[BASE + 2 * STRIDE].x), "+v"(z[BASE + 2 * STRIDE].y), "+v"(z[BASE + 3 * STRIDE].x), "+v"(z[BASE + 3 * STRIDE].y));
  if constexpr (CNT == 8)
    asm volatile("" : "+v"(z[BASE + 4 * STRIDE].x), "+v"(z[BASE + 4 * STRIDE].y), "+v"(z[BASE + 5 * STRIDE].x), "+v"(z[BASE + 5 * STRIDE].y),
                      "+v"(z[BASE + 6 * STRIDE].x), "+v"(z[BASE + 6 * STRIDE].y), "+v"(z[BASE + 7 * STRIDE].x), "+v"(z[BASE + 7 * STRIDE].y));
}
// the 4 x 8 in-register transforms of fft_regs.h with a scheduling fence behind every butterfly (kernel_regtile64p.h: hipcc otherwise
// interleaves all butterflies of a stage and parks the temporaries in scratch)
template <bool INV>
__device__ __forceinline__ void q32_fftA(float2 (&z)[32]) {
  static_for<0, 8>([&](auto q0c) { constexpr int q0 = decltype(q0c)::value; bfly_plain<4, INV, q0, 8, 32>(z); q32_pin<q0, 8, 4>(z); __builtin_amdgcn_sched_barrier(0); });
  static_for<0, 4>([&](auto kac) { constexpr int ka = decltype(kac)::value; fftA_stage2_group<4, 8, INV, ka>(z); q32_pin<8 * ka, 1, 8>(z); __builtin_amdgcn_sched_barrier(0); });
}
template <bool INV>
__device__ __forceinline__ void q32_fftB(float2 (&z)[32]) {
  static_for<0, 4>([&](auto kac) { constexpr int ka = decltype(kac)::value; fftB_stage1_group<4, 8, INV, ka>(z); q32_pin<8 * ka, 1, 8>(z); __builtin_amdgcn_sched_barrier(0); });
  static_for<0, 8>([&](auto nc) { constexpr int nlo = decltype(nc)::value; fftB_stage2_group<4, 8, INV, nlo>(z); q32_pin<nlo, 8, 4>(z); __builtin_amdgcn_sched_barrier(0); });
}

// VARIANT bit 0: the cross-lane steps as single v_fmac_f32_dpp (0: through __builtin_amdgcn_update_dpp, version 0's code);
// ablations (wrong results, timing only): bit 1 = no LDS exchange traffic (the barriers stay), bit 2 = no in-register transforms / twiddles /
// gate (the exchanges, the cross-lane steps' data flow and every request stay)
// Persistent: workgroup w (XCD-contiguous, pairs on adjacent tiles like kernel_regtile64p.h) walks through a.tpw tiles; NOT pipelined:
// every tile is loaded, transformed and stored before the next one is requested (the next tile's gate bins travel in 3 registers).
template <int VARIANT = 0>
__global__ void __launch_bounds__(1024) spectre_mix_q32(const RegtileArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);
  float2* glds = reinterpret_cast<float2*>(smem + kQ32GateOff);
  float2* twl = reinterpret_cast<float2*>(smem + kQ32TwOff);
  float2* ltw = reinterpret_cast<float2*>(smem + kQ32LaneTwOff);
  constexpr float inv_n = 1.0f / 4096.0f;
  const int tid = threadIdx.x;
  // Only threadIdx.x stays live across the tile loop; the lane coordinates are re-derived from an opaque copy per tile (otherwise LICM
  // hoists every per-lane address out of the loop and the allocator spills them: 48 spilled registers in the first persistent build)
  int lane, wave, pp, h, p, n2, L, pb, k1, qL;
  float* wr; float* rd;
  float tA, tB; bool l3;
  auto coords = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    lane = t & 63; wave = t >> 6;
    pp = lane & 3; h = (lane >> 4) & 1; p = 2 * pp + h;                     // rows role
    n2 = ((lane >> 2) & 3) + 4 * (lane >> 5) + 8 * wave;
    L = lane & 3; pb = (lane >> 2) & 7; k1 = (lane >> 5) + 2 * wave;        // bins role
    qL = L == 0 ? 0 : L == 1 ? 2 : L == 2 ? 1 : 3;
    wr = img + p * kQ32PS + q32_slot(n2);                                   // + k1 * RW
    rd = img + k1 * kQ32RW + pb * kQ32PS + 32 * L + 2 * (L >> 1);           // + m
    tA = L < 2 ? 1.0f : -1.0f; tB = (L & 1) ? -1.0f : 1.0f; l3 = L == 3;
  };

  // ---- tables (once per workgroup)
  for (int i = tid; i < 128 * 10; i += 1024) {
    const int nn = i / 10, e = i - 10 * nn;
    twl[i] = a.tw[e < 3 ? nn * (e + 1) : 4 * nn * (e - 2)];
  }
  if (tid < 128) {
    const int l = tid >> 5, m = tid & 31, q = l == 0 ? 0 : l == 1 ? 2 : l == 2 ? 1 : 3;
    float2 w = a.tw[(32 * m * q) & 4095];
    if (l == 1 || l == 2) { w.x = -w.x; w.y = -w.y; }
    ltw[l * 34 + m] = w;
  }
  __syncthreads();

  const int wg_lin = xcd_contiguous(blockIdx.x, a.n_wg);
  const int pair_base = (wg_lin / 2) * a.tpw * 2 + (wg_lin % 2);
  if (pair_base >= a.n_tiles) return;
  auto tile_ptrs = [&](int tile, const char*& vb, char*& ob, const float2*& gp) {
    const int b = tile / a.tiles_per_row, ct = tile - b * a.tiles_per_row;
    vb = reinterpret_cast<const char*>(a.v) + ((size_t)b * a.v_sb + (size_t)ct * 16) * 4;
    ob = reinterpret_cast<char*>(a.out) + ((size_t)b * a.out_sb + (size_t)ct * 16) * 4;
    gp = a.gate + ((size_t)b * a.G + (ct * 16) / a.d_g) * a.F;
  };
  float2 gstage[3];
  auto gate_fetch = [&](const float2* gp) {
    static_for<0, 3>([&](auto ic) { constexpr int i = decltype(ic)::value; const int k = tid + 1024 * i; gstage[i] = gp[i < 2 ? k : (k <= 2048 ? k : 2048)]; });
  };
  auto gate_commit = [&]() {
    static_for<0, 3>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int k = tid + 1024 * i;
      float2 g = gstage[i];
      if (k == 0 || k == 2048) g.y = 0.f;          // irfft ignores Im(DC), Im(Nyquist) (spectre.py:551)
      if (a.conj_gate) g.y = -g.y;
      if (i < 2 || k <= 2048) glds[k + (k >> 5)] = make_float2(g.x * inv_n, g.y * inv_n);
    });
  };
  {
    const char* vb; char* ob; const float2* gp;
    tile_ptrs(pair_base, vb, ob, gp);
    gate_fetch(gp);
  }

  float2 z[32];
  auto write_rows = [&](auto is_im) {
    static_for<0, 4>([&](auto kac) {
      static_for<0, 8>([&](auto kbc) {
        constexpr int ka = decltype(kac)::value, kb = decltype(kbc)::value, j = 8 * ka + kb, kk = ka + 4 * kb;
        wr[kk * kQ32RW] = decltype(is_im)::value ? z[j].y : z[j].x;
      });
    });
  };
  auto read_rows = [&](auto is_im) {
    static_for<0, 4>([&](auto kac) {
      static_for<0, 8>([&](auto kbc) {
        constexpr int ka = decltype(kac)::value, kb = decltype(kbc)::value, j = 8 * ka + kb, kk = ka + 4 * kb;
        if constexpr (decltype(is_im)::value) z[j].y = wr[kk * kQ32RW]; else z[j].x = wr[kk * kQ32RW];
      });
    });
  };
  auto read_run = [&](auto is_im) {
    static_for<0, 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const float2 v = *reinterpret_cast<const float2*>(rd + 2 * i);
      if constexpr (decltype(is_im)::value) { z[2 * i].y = v.x; z[2 * i + 1].y = v.y; } else { z[2 * i].x = v.x; z[2 * i + 1].x = v.y; }
    });
  };
  auto write_run = [&](auto is_im) {
    static_for<0, 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      *reinterpret_cast<float2*>(rd + 2 * i) = decltype(is_im)::value ? make_float2(z[2 * i].y, z[2 * i + 1].y) : make_float2(z[2 * i].x, z[2 * i + 1].x);
    });
  };

  for (int it = 0; it < a.tpw; ++it) {
    const int tile = pair_base + 2 * it;
    if (tile >= a.n_tiles) break;
    const bool more = (it + 1 < a.tpw) && (tile + 2 < a.n_tiles);
    coords();
    long long v_sn = a.v_sn, out_sn = a.out_sn;
    asm volatile("" : "+s"(v_sn), "+s"(out_sn));
    const char* vb; char* ob; const float2* gp;
    tile_ptrs(tile, vb, ob, gp);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vb), 0, (int)((long long)a.rows_in * v_sn * 4), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)((long long)a.rows_out * out_sn * 4), kRsrcFlags);

    // ---- loads: instruction r fetches row n2 + 128 (2 r + h), this lane's 4 channels = sequences 2 pp and 2 pp + 1
    const uint32_t voff = (uint32_t)(((long long)(n2 + 128 * h) * v_sn + 4 * pp) * 4);
    static_for<0, 16>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      const q32_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff + (uint32_t)((long long)(256 * r) * v_sn * 4), 0, 0);
      z[2 * r] = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
      z[2 * r + 1] = make_float2(__uint_as_float(t.z), __uint_as_float(t.w));
    });
    gate_commit();                                   // (glds is free: the previous tile's gate was last read in front of its E2)
    if (more) { const char* vbn; char* obn; const float2* gpn; tile_ptrs(tile + 2, vbn, obn, gpn); gate_fetch(gpn); }
    // position 2 r holds row n1 = 2 r (h = 0 lanes loaded it), 2 r + 1 holds n1 = 2 r + 1, both of sequence p
    static_for<0, 16>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      q32_lane16_swap(z[2 * r].x, z[2 * r + 1].x);
      q32_lane16_swap(z[2 * r].y, z[2 * r + 1].y);
    });

    // ---- F1: 32-point forward over n1 (type A: k1 = ka + 4 kb at position 8 ka + kb), then W_N^(n2 k1)
    if constexpr (!(VARIANT & 4)) q32_fftA<false>(z);
    if constexpr (!(VARIANT & 4)) {
      float2 wa[4], wb[8];
      static_for<1, 4>([&](auto jc) { wa[decltype(jc)::value] = twl[n2 * 10 + decltype(jc)::value - 1]; });
      static_for<1, 8>([&](auto jc) { wb[decltype(jc)::value] = twl[n2 * 10 + 2 + decltype(jc)::value]; });
      static_for<0, 4>([&](auto kac) {
        static_for<0, 8>([&](auto kbc) {
          constexpr int ka = decltype(kac)::value, kb = decltype(kbc)::value, j = 8 * ka + kb;
          if constexpr (ka > 0) z[j] = cmul(z[j], wa[ka]);
          if constexpr (kb > 0) z[j] = cmul(z[j], wb[kb]);
        });
      });
    }
    // ---- E1
    q32_barrier();                                   // the image is free (E2 of the previous tile has been read), the gate is in place
    if constexpr (!(VARIANT & 2)) write_rows(std::false_type{});
    q32_barrier();
    if constexpr (!(VARIANT & 2)) read_run(std::false_type{});
    q32_barrier();
    if constexpr (!(VARIANT & 2)) write_rows(std::true_type{});
    q32_barrier();
    if constexpr (!(VARIANT & 2)) read_run(std::true_type{});
    // (the image stays busy until the barrier in front of E2's first write)

    // ---- F2: radix-4 across the quad (input lane L holds n2 = m + 32 L), per-lane twiddle, 32-point forward over m
    static_for<0, 32>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      float x = z[m].x, y = z[m].y;
      if constexpr ((VARIANT & 1) == 0) {
        x = __builtin_fmaf(q32_dpp<kQuadX2>(x), tA, x);               // L0: s0, L1: s1, L2: -d0, L3: -d1
        y = __builtin_fmaf(q32_dpp<kQuadX2>(y), tA, y);
      } else { x = q32_fmac_x2(x, tA); y = q32_fmac_x2(y, tA); }
      float rx = l3 ? y : x, ry = l3 ? -x : y;                        // lane 3: * (-i)
      if constexpr ((VARIANT & 1) == 0) {
        x = __builtin_fmaf(q32_dpp<kQuadX1>(rx), tB, rx);             // L0: C0, L1: -C2, L2: -C1, L3: C3
        y = __builtin_fmaf(q32_dpp<kQuadX1>(ry), tB, ry);
      } else { x = q32_fmac_x1(rx, tB); y = q32_fmac_x1(ry, tB); }
      z[m] = cmul(make_float2(x, y), ltw[L * 34 + m]);                // (signs included in the table)
      if constexpr (m % 8 == 7) __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (!(VARIANT & 4)) q32_fftA<false>(z);                      // bins k2 = 4 k2'' + q_L, k2'' = ka + 4 kb at position 8 ka + kb

    // ---- gate (spectre.py:545): k = k1 + 32 q_L + 128 k2''; above N/2 the Hermitian extension
    {
      const int kq = k1 + 32 * qL;
      const float2* glo = glds + kq + qL;
      const float2* ghi = glds + (4096 - kq) + ((4096 - kq) >> 5);
      static_for<0, 4>([&](auto kac) {
        static_for<0, 8>([&](auto kbc) {
          constexpr int ka = decltype(kac)::value, kb = decltype(kbc)::value, j = 8 * ka + kb, k2 = ka + 4 * kb;
          float2 g;
          if constexpr (k2 < 16) g = glo[132 * k2];
          else { g = ghi[-132 * k2]; g.y = -g.y; }
          z[j] = cmul(z[j], g);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    }

    // ---- I1: 32-point inverse over k2'' (type B: m at position m), conj twiddle, radix-4 across the quad -> lane L holds n2 = m + 32 L
    if constexpr (!(VARIANT & 4)) q32_fftB<true>(z);
    static_for<0, 32>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const float2 t = cmulc(z[m], ltw[L * 34 + m]);                // L0: P0, L1: -P2, L2: -P1, L3: P3
      float x = t.x, y = t.y;
      if constexpr ((VARIANT & 1) == 0) {
        x = __builtin_fmaf(q32_dpp<kQuadX1>(x), -tB, x);              // L0: a, L1: b, L2: -c, L3: -e
        y = __builtin_fmaf(q32_dpp<kQuadX1>(y), -tB, y);
      } else { x = q32_fmac_x1(x, -tB); y = q32_fmac_x1(y, -tB); }
      float rx = l3 ? -y : x, ry = l3 ? x : y;                        // lane 3: * (+i)
      if constexpr ((VARIANT & 1) == 0) {
        x = __builtin_fmaf(q32_dpp<kQuadX2>(rx), -tA, rx);            // y[m + 32 L]
        y = __builtin_fmaf(q32_dpp<kQuadX2>(ry), -tA, ry);
      } else { x = q32_fmac_x2(rx, -tA); y = q32_fmac_x2(ry, -tA); }
      z[m] = make_float2(x, y);
      if constexpr (m % 8 == 7) __builtin_amdgcn_sched_barrier(0);
    });

    // ---- E2: the same image addresses the other way round
    q32_barrier();                                                   // every wave has finished E1's reads
    if constexpr (!(VARIANT & 2)) write_run(std::false_type{});
    q32_barrier();
    if constexpr (!(VARIANT & 2)) read_rows(std::false_type{});
    q32_barrier();
    if constexpr (!(VARIANT & 2)) write_run(std::true_type{});
    q32_barrier();
    if constexpr (!(VARIANT & 2)) read_rows(std::true_type{});

    // ---- conj twiddle, I2 (type B inverse: n1 at position n1), stores (spectre.py:553)
    {
      float2 wa[4], wb[8];
      static_for<1, 4>([&](auto jc) { wa[decltype(jc)::value] = twl[n2 * 10 + decltype(jc)::value - 1]; });
      static_for<1, 8>([&](auto jc) { wb[decltype(jc)::value] = twl[n2 * 10 + 2 + decltype(jc)::value]; });
      static_for<0, 4>([&](auto kac) {
        static_for<0, 8>([&](auto kbc) {
          constexpr int ka = decltype(kac)::value, kb = decltype(kbc)::value, j = 8 * ka + kb;
          if constexpr (ka > 0) z[j] = cmulc(z[j], wa[ka]);
          if constexpr (kb > 0) z[j] = cmulc(z[j], wb[kb]);
        });
      });
    }
    if constexpr (!(VARIANT & 4)) q32_fftB<true>(z);
    const uint32_t ooff = (uint32_t)(((long long)(n2 + 128 * h) * out_sn + 4 * pp) * 4);
    static_for<0, 16>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      q32_lane16_swap(z[2 * r].x, z[2 * r + 1].x);
      q32_lane16_swap(z[2 * r].y, z[2 * r + 1].y);
      q32_u32x4 t;
      t.x = __float_as_uint(z[2 * r].x); t.y = __float_as_uint(z[2 * r].y); t.z = __float_as_uint(z[2 * r + 1].x); t.w = __float_as_uint(z[2 * r + 1].y);
      __builtin_amdgcn_raw_buffer_store_b128(t, rs_out, ooff + (uint32_t)((long long)(256 * r) * out_sn * 4), 0, 0);
    });
  }
}

}  // namespace sfft
