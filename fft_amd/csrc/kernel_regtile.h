// kernel_regtile.h — register-resident spectral mix for n_fft = RF*RS, RF in {RS, 2*RS}, RS in {16, 32, 64}
// (n_fft = 256, 512, 1024, 2048, 4096) on gfx950.
//
// One workgroup owns a tile of 16 adjacent channels (one 64-byte row segment in fp32) for ALL n_fft rows of one
// batch element; the tile never leaves the CU between the single HBM read and the single HBM write (replaces
// /root/reference/spectre.py:506 + :542-553, which make 8-9 HBM passes).
//
// Math.  Two real channels (c, c+1) of the same gate group are packed as one complex sequence z = x_c + i x_{c+1}.
// Because the filter's impulse response irfft(gate) is REAL, filtering acts on Re and Im independently, so
//   y_c + i y_{c+1} = IDFT( Gf * DFT(z) + Mf ),
// with Gf the Hermitian extension of the half-spectrum gate (Im dropped at DC and Nyquist — spectre.py:551's irfft
// ignores them) and Mf[k] = mem_c[k] + i mem_{c+1}[k] (k <= N/2), conj(mem_c[N-k]) + i conj(mem_{c+1}[N-k]) otherwise.
//
// The length-N complex DFT is the two-pass Cooley-Tukey split n = n2 + RS*n1 (n1 < RF), k = k1 + RF*k2 (k2 < RS):
//   F1  thread (p,u):  A[k1]  = sum_n1 z[u + RS n1] W_RF^(n1 k1)        (RF-point, in registers, type A)
//                      A[k1] *= W_N^(u k1)                               (per-thread twiddle bases)
//   E1  exchange through LDS: value (u, k1) -> thread k1 mod RS, set k1 div RS, slot u
//   F2  thread (p,s), each set t (k1 = s + RS t):  X[k1 + RF k2] = sum_n2 A_n2[k1] W_RS^(n2 k2)   (type A)
//       gate:          Y = X * Gf (+ Mf), 1/N folded in       } fused per register group with the last stage of F2
//   I1                 C[n2]  = sum_k2 Y[k1 + RF k2] W_RS^(-n2 k2)   (type B)   } and the first stage of I1
//   E2  exchange: value (s, t, n2) -> thread n2, slot k1
//                      C[k1] *= conj(W_N^(u k1))                         (same bases as F1)
//   I2  thread (p,u):  y[u + RS n1] = sum_k1 C[k1] W_RF^(-n1 k1)         (type A, inverse)
//
// Geometry.  lane = (p = lane & 7 : pair-column, row class = lane >> 3), u = row class + 8 * wave, so a wave-wide
// 8-byte load covers 8 rows x 64 contiguous bytes.  64-byte segments reach the copy ceiling only when the
// neighbouring tile (other half of the 128-B line) is in flight in the same XCD's L2 at the same time: tiles are
// therefore dealt to workgroups XCD-contiguously (profiles/r01_segcopy_microbench.log).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "fft_regs.h"

// LDS image layout per (RF, RS): 1 = [row][column][slot] with 16-byte reads, 0 = [row][slot][column] with 4-byte reads.
// The 16-byte layout needs 4 floats of padding per column instead of 8 per row; at n_fft = 2048 that pushes the image
// from 74 KiB to 82 KiB per workgroup, i.e. from two workgroups per CU to one (1.17 ms instead of 0.71 ms), so that size
// keeps the 4-byte layout.  Everywhere else the 16-byte reads win 2-3 % (profiles/r01_ablation.log).
#define SFFT_EXCHANGE_B128(RF, RS) (((RF) == 64 && (RS) == 32) ? 0 : 1)

namespace sfft {

typedef unsigned int rt_u32x2 __attribute__((ext_vector_type(2)));
constexpr int kRsrcFlags = 0x00020000;       // raw buffer resource, dword 3 (gfx90a / gfx942 / gfx950)

struct RegtileArgs {
  const void* v;
  const float2* gate;   // (B, G, F) complex64
  const float* mem;     // (F, D, 2) float or nullptr
  void* out;
  const float2* tw;     // exp(-2 pi i m / N), m = 0..N-1 (host-computed in double)
  int B, N_in, D, G, d_g, F;
  int tiles_per_row, n_tiles;
  long long v_sb, v_sn, out_sb, out_sn;   // element strides
  int tpw;              // tiles per workgroup (>= 1)
  int n_wg;             // workgroups launched = 2 * ceil(n_tiles / (2 * tpw))
  int conj_gate;        // 1: filter with conj(gate) — the adjoint w.r.t. v (dV = mix(dOut, conj(gate)))
  int rows_in, rows_out;   // kernel_regtile64p.h: input rows that exist (min(N_in, n_fft); the rest is rfft's zero padding) and output rows
                           // that are written (spectre.py:553) = the extents of its buffer resources
  unsigned* tickets;       // kernel_regtile64p.h TICKETS: this launch's slice of the plan's ticket ring (uncached device memory, zeroed on
                           // the launch's stream): counter | gang mailboxes | one claim bit per tile.  nullptr: static tile map
};

constexpr int kPC = 8;                       // pair-columns per tile: 16 channels, 64-byte fp32 row segments

template <int RF, int RS> constexpr int regtile_threads() { return kPC * RS; }
// LDS exchange image, one float plane: E1 is [RF rows][RS sources][8 columns], E2 is [RS rows][RF sources][8];
// every row is padded by 32 bytes so that the 4 team indices of a 32-lane group read from distinct banks
template <int RF, int RS, int XV = SFFT_EXCHANGE_B128(RF, RS)> constexpr int regtile_image_bytes() {
  return XV ? (RF * kPC * (RS + 4) > RS * kPC * (RF + 4) ? RF * kPC * (RS + 4) : RS * kPC * (RF + 4)) * 4
            : RF * RS * kPC * 4 + (RF > RS ? RF : RS) * kPC * 4;
}
template <int RF, int RS> constexpr int regtile_gate_lds_bytes() { return (RF * RS / 2 + 1) * 8; }   // half-spectrum gate
template <int RF, int RS, int XV = SFFT_EXCHANGE_B128(RF, RS)> constexpr int regtile_lds_total() { return regtile_image_bytes<RF, RS, XV>() + regtile_gate_lds_bytes<RF, RS>(); }

// Workgroup id -> position.  Workgroup w is observed to run on XCD w % 8 (speed only, never correctness): give
// every XCD a contiguous run of tiles so tiles sharing 128-B lines meet in one L2.  Bijective for any n.
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
// Two values -> one packed dword (a in the low half), the same rounding: gfx950 has an instruction for it (v_cvt_pk_bf16_f32, round to
// nearest even, overflow to infinity) where the bit arithmetic above costs ~12 VALU instructions per pair — 700 of the ~5 400 per thread and
// 4096-row tile in the kernels that write bf16 rows (round 5).  Bit-identical to the scalar form (and to torch's conversion) for every
// input that is not a NaN.  NaNs (round 6): the instruction returns a QUIET NaN that keeps the input's sign and upper payload bits
// (0x7fc1 for 0x7fc10000, 0xffc0 for 0xffc00000, 0x7fc0 for the signalling 0x7f800001; tools/gpu_jobs/r06_nanfix.sh), where the scalar
// form and torch return the canonical 0x7fc0 for every NaN.  Rounds 4-5 patched the canonical pattern in behind the instruction —
// two compares, two selects, and / or and the wait states of the VCC hazard per pair: 900 instructions per thread and tile, 3.3 % of the
// bf16 -> bf16 launch (1.121 -> 1.084 ms, profiles/r06_nanfix_ab.log) — for a value that is "not a number" either way.  Not any more:
// a NaN result is a quiet NaN, its sign and payload are unspecified.  -DSPECTRE_BF16_CANONICAL_NAN restores the patch.
typedef __bf16 rt_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rt_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f32x2_to_bf16x2_rne(float a, float b) {
  const rt_f32x2 v = {a, b};
  uint32_t r = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rt_bf16x2));
#ifdef SPECTRE_BF16_CANONICAL_NAN
  if (a != a) r = (r & 0xffff0000u) | 0x7fc0u;
  if (b != b) r = (r & 0x0000ffffu) | 0x7fc00000u;
#endif
  return r;
}

// LDS exchange of one float plane at a time (real parts, then imaginary parts): the tile (RF*RS*8 columns*8 B =
// 256 KiB at 4096) does not fit the 160 KiB LDS, and this keeps the live register set at RF complex values (re in +
// im out) instead of 1.5x for a two-round 8-byte exchange, which spills.  wr(j) / rd(m) give the float index inside
// the image for register position j (write) and m (read).  Every ds_write_b32 is lane-linear (256 B per wave);
// every ds_read_b32 of a 32-lane group hits 32 distinct banks thanks to the 32-byte pad per row.
// The image is free on entry: every exchange ENDS with a barrier, so the first writes can be scheduled into the
// code that produces z[j].  Reads follow the order in which the next butterfly stage consumes them.
template <int E, int RA, int RB, bool LAST_BARRIER = true, class WR, class RD>
__device__ __forceinline__ void exchange_planes(float2 (&z)[E], float* img, WR wr, RD rd) {
  constexpr int SUB = RA * RB;             // consumer works on sub-arrays of SUB values (E / SUB sets)
  static_for<0, E>([&](auto jc) { constexpr int j = decltype(jc)::value; img[wr(jc)] = z[j].x; });
  __syncthreads();
  static_for<0, E>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int m = (i / SUB) * SUB + ((i % SUB) / RA) + RB * ((i % SUB) % RA);
    z[m].x = img[rd(std::integral_constant<int, m>{})];
  });
  __syncthreads();
  static_for<0, E>([&](auto jc) { constexpr int j = decltype(jc)::value; img[wr(jc)] = z[j].y; });
  __syncthreads();
  static_for<0, E>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int m = (i / SUB) * SUB + ((i % SUB) / RA) + RB * ((i % SUB) % RA);
    z[m].y = img[rd(std::integral_constant<int, m>{})];
  });
  if constexpr (LAST_BARRIER) __syncthreads();   // image free again for the next exchange
}

// Variant with 16-byte reads: image laid out [row][column p][slot], slot fastest, every column padded by 4 floats.
// A thread's slots of one row are then contiguous, so each plane is read with E/4 ds_read_b128 (256 B/clk) instead of
// E/2 ds_read2_b32 (128 B/clk); the ds_write_b32 side stays conflict-free (bank = 4p + row class), as are the b128
// reads (checked for every (RF, RS) with the lane-group table of MI355X_MICROARCH.md).  wr(j) -> float index of
// position j; rd4(m4) -> float index of the 4 slots m4..m4+3 (16-byte aligned).
template <int E, int RA, int RB, bool LAST_BARRIER = true, class WR, class RD4>
__device__ __forceinline__ void exchange_planes_b128(float2 (&z)[E], float* img, WR wr, RD4 rd4) {
  constexpr int SUB = RA * RB;             // consumer works on sub-arrays of SUB values
  auto read_plane = [&](auto is_im) {
    static_for<0, E / 4>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      // chunk order: within each sub-array, chunks holding slots {RB*q1 + 0..3} first (q1 = 0..RA-1), then +4..7, ...
      constexpr int CPS = SUB / 4, CPR = (RB >= 4 ? RB / 4 : 1);     // chunks per sub-array, chunks per RB-run
      constexpr int sub = i / CPS, ii = i % CPS;
      constexpr int chunk = (RB >= 4) ? (ii / RA) + CPR * (ii % RA) : ii;
      constexpr int m = sub * SUB + 4 * chunk;
      const float4 v = *reinterpret_cast<const float4*>(img + rd4(std::integral_constant<int, m>{}));
      if constexpr (decltype(is_im)::value) { z[m].y = v.x; z[m + 1].y = v.y; z[m + 2].y = v.z; z[m + 3].y = v.w; }
      else { z[m].x = v.x; z[m + 1].x = v.y; z[m + 2].x = v.z; z[m + 3].x = v.w; }
    });
  };
  static_for<0, E>([&](auto jc) { constexpr int j = decltype(jc)::value; img[wr(jc)] = z[j].x; });
  __syncthreads();
  read_plane(std::false_type{});
  __syncthreads();
  static_for<0, E>([&](auto jc) { constexpr int j = decltype(jc)::value; img[wr(jc)] = z[j].y; });
  __syncthreads();
  read_plane(std::true_type{});
  if constexpr (LAST_BARRIER) __syncthreads();   // image free again for the next exchange
}

// The same exchange with the E scattered dword writes of a plane issued as E/2 ds_write2st64_b32.  The LDS takes a store's address
// and data registers at 2 cycles per dword: two ds_write_b32 cost 8 cycles, one ds_write2 with two data dwords 6 (MI355X_MICROARCH.md),
// and the writes are ~80 % of an exchange's LDS time.  Position j = pos(row, t) is written to float index
// wbase + t * TS + row * RWF; rows r and r + 2 are 2 * RWF * 4 bytes apart, a multiple of the instruction's 256-byte offset unit for
// every image layout here (RWF = 8 (R + 4), R a multiple of 4).  Its 8-bit offsets reach WIN rows, hence one opaque base address per
// (t, row parity, window) instead of a single base with 16-bit offsets.
// HOOK (optional): called with std::integral_constant<int, 0 / 1 / 2> behind the three inner barriers — a caller that keeps global loads
// in flight across the exchange (kernel_regtile_grad.h, prefetch form) issues more of them there; such a caller passes LDS_BAR = true:
// __syncthreads() is a fence + s_barrier, and hipcc implements the fence with s_waitcnt vmcnt(0), which would drain those loads.
struct ExchangeNoHook { template <class T> __device__ __forceinline__ void operator()(T) const {} };
template <int E, int RA, int RB, bool LAST_BARRIER, int ROWS, int NT, int TS, int RWF, bool LDS_BAR = false, class POS, class RD4, class HOOK = ExchangeNoHook>
__device__ __forceinline__ void exchange_planes_b128_w2(float2 (&z)[E], float* img, int wbase, POS pos, RD4 rd4, HOOK hook = HOOK{}) {
  auto bar = [] { if constexpr (LDS_BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads(); };
  static_assert(ROWS * NT == E && ROWS % 4 == 0 && (2 * RWF * 4) % 256 == 0, "row pairs (r, r + 2) must be whole offset units apart");
  constexpr int UNITS2 = 2 * RWF * 4 / 256;                     // offset units between rows r and r + 2
  constexpr int WIN = ((255 / UNITS2) * 2 + 2) / 4 * 4;         // rows per window (multiple of 4): last pair starts at row WIN - 4 + {0,1}
  constexpr int NW = (ROWS + WIN - 1) / WIN;
  typedef __attribute__((address_space(3))) float lds_float;
  lds_float* wb[NT * 2 * NW];
  static_for<0, NT * 2 * NW>([&](auto ic) {
    constexpr int i = decltype(ic)::value, t = i / (2 * NW), par = (i / NW) % 2, w = i % NW;
    wb[i] = (lds_float*)(img + wbase) + t * TS + (w * WIN + par) * RWF;
    asm volatile("" : "+v"(wb[i]));                              // keep them apart: base + 16-bit offset would fold them back together
  });
  constexpr int SUB = RA * RB;
  auto write_plane = [&](auto is_im) {
    static_for<0, NT * (ROWS / 4)>([&](auto ic) {
      constexpr int t = decltype(ic)::value / (ROWS / 4), r0 = 4 * (decltype(ic)::value % (ROWS / 4)), w = r0 / WIN, rw = r0 % WIN;
      static_for<0, 2>([&](auto parc) {
        constexpr int par = decltype(parc)::value;
        constexpr int ja = decltype(pos(std::integral_constant<int, r0 + par>{}, std::integral_constant<int, t>{}))::value;
        constexpr int jb = decltype(pos(std::integral_constant<int, r0 + par + 2>{}, std::integral_constant<int, t>{}))::value;
        lds_float* b = wb[(t * 2 + par) * NW + w];
        if constexpr (decltype(is_im)::value) { b[rw * RWF] = z[ja].y; b[(rw + 2) * RWF] = z[jb].y; }
        else { b[rw * RWF] = z[ja].x; b[(rw + 2) * RWF] = z[jb].x; }
      });
    });
  };
  auto read_plane = [&](auto is_im) {
    static_for<0, E / 4>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int CPS = SUB / 4, CPR = (RB >= 4 ? RB / 4 : 1);
      constexpr int sub = i / CPS, ii = i % CPS;
      constexpr int chunk = (RB >= 4) ? (ii / RA) + CPR * (ii % RA) : ii;
      constexpr int m = sub * SUB + 4 * chunk;
      const float4 v = *reinterpret_cast<const float4*>(img + rd4(std::integral_constant<int, m>{}));
      if constexpr (decltype(is_im)::value) { z[m].y = v.x; z[m + 1].y = v.y; z[m + 2].y = v.z; z[m + 3].y = v.w; }
      else { z[m].x = v.x; z[m + 1].x = v.y; z[m + 2].x = v.z; z[m + 3].x = v.w; }
    });
  };
  write_plane(std::false_type{});
  bar();
  hook(std::integral_constant<int, 0>{});
  read_plane(std::false_type{});
  bar();
  hook(std::integral_constant<int, 1>{});
  write_plane(std::true_type{});
  bar();
  hook(std::integral_constant<int, 2>{});
  read_plane(std::true_type{});
  if constexpr (LAST_BARRIER) bar();
}

// MODE 0 (fast): N_in >= n_fft (no row predicates), no memory_fft, every tile inside one gate group (gate staged in LDS).
// MODE 1 (general): row predicates, any even d_g (gate read from global memory).  MODE 2: general + memory_fft.
// MODE 3: row predicates only (N_in < n_fft, the padded-sequence case) with the gate still staged in LDS.
// MODE 4: MODE 3 + memory_fft.
template <int RF, int RS, bool IN_BF16, bool OUT_BF16, int MODE, int XV = SFFT_EXCHANGE_B128(RF, RS)>
__global__ void __launch_bounds__(kPC * RS, 2)   // at least two waves per SIMD (VGPR + AGPR budget 256): two 64x32 workgroups per CU
spectre_mix_regtile(const RegtileArgs a) {
  constexpr bool GENERAL = MODE != 0, WITH_MEM = MODE == 2 || MODE == 4, GATE_LDS = MODE == 0 || MODE == 3 || MODE == 4;
  static_assert(RF == RS || RF == 2 * RS, "n_fft = RS*RS or 2*RS*RS");
  constexpr int N = RF * RS, NS = RF / RS;                  // NS sets of RS values per thread in the middle phase
  constexpr int RAF = FftCfg<RF>::RA, RBF = FftCfg<RF>::RB; // RF-point transforms (F1, I2)
  constexpr int RAS = FftCfg<RS>::RA, RBS = FftCfg<RS>::RB; // RS-point transforms (F2, I1)
  constexpr int ROW1 = RS * kPC + kPC, ROW2 = RF * kPC + kPC;   // image row lengths in floats (E1, E2)
  constexpr int ES_IN = IN_BF16 ? 2 : 4, ES_OUT = OUT_BF16 ? 2 : 4;
  constexpr float inv_n = 1.0f / (float)N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);
  float2* glds = reinterpret_cast<float2*>(smem + regtile_image_bytes<RF, RS, XV>());

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int p0 = lane & (kPC - 1);
  const int u0 = (lane / kPC) + (64 / kPC) * wave;   // team index: n2 in F1/I2, k1 mod RS in F2/I1

  // Workgroup w handles tpw tiles.  Workgroups 2m and 2m+1 (same XCD) walk through ADJACENT tiles in step, so the
  // two 64-byte halves of every 128-byte line are requested within the same few microseconds and merge in that L2.
  const int wg_lin = xcd_contiguous(blockIdx.x, a.n_wg);
  const int pair_base = (wg_lin >> 1) * a.tpw * 2 + (wg_lin & 1);

  for (int it = 0; it < a.tpw; ++it) {
  const int tile = pair_base + 2 * it;
  if (tile >= a.n_tiles) break;                  // workgroup-uniform
  if (it > 0) __syncthreads();                   // the previous tile's last exchange left its image reads unfenced
  // Opaque per-iteration copies of the lane coordinates and row strides: otherwise LICM hoists every per-lane
  // address (twiddle table, gate offsets, 2*RF row offsets) out of the tile loop and the allocator spills them.
  int p = p0, u = u0;
  asm volatile("" : "+v"(p), "+v"(u));
  long long v_sn = a.v_sn, out_sn = a.out_sn;
  asm volatile("" : "+s"(v_sn), "+s"(out_sn));
  const int b = tile / a.tiles_per_row;
  const int ct = tile - b * a.tiles_per_row;
  const int c = ct * (2 * kPC) + 2 * p;          // first channel of this lane's pair
  bool cvalid = true;                            // the last tile of a row is ragged when D % 16 != 0 (general modes only)
  if constexpr (GENERAL) cvalid = c < a.D;

  // per-thread twiddle bases W_N^(u*ka), W_N^(u*RAF*kb): products give W_N^(u*k1) for any k1 < RF.  They are
  // (re)loaded from the L1/L2-resident table where they are used, twice per tile: holding 2*(RAF+RBF-2) registers
  // across the whole tile pushes the 64-point kernel past 256 VGPRs into scratch.
  auto load_twiddle_bases = [&](float2 (&wa)[RAF], float2 (&wb)[RBF]) {
    static_for<1, RAF>([&](auto jc) { constexpr int j = decltype(jc)::value; wa[j] = a.tw[u * j]; });
    static_for<1, RBF>([&](auto jc) { constexpr int j = decltype(jc)::value; wb[j] = a.tw[u * RAF * j]; });
  };

  // ---- gate -> LDS (behind the exchange image): each of the N/2+1 bins is fetched from global memory once per
  // tile instead of once per wave, pre-scaled by 1/N, with Im(DC) and Im(Nyquist) already dropped.  Compile-time
  // variant: the host picks it only when all 16 channels of a tile share one gate group (d_g % 16 == 0); E1's
  // barriers order fill and use.  Issued BEFORE the tile loads: VMEM returns in order.
  if constexpr (GATE_LDS) {
    const float2* gp = a.gate + ((size_t)b * a.G + (ct * (2 * kPC)) / a.d_g) * a.F;
    for (int k = tid; k <= N / 2; k += kPC * RS) {
      float2 g = gp[k];
      if (k == 0 || k == N / 2) g.y = 0.f;         // irfft ignores Im(DC), Im(Nyquist)
      if (a.conj_gate) g.y = -g.y;
      glds[k] = make_float2(g.x * inv_n, g.y * inv_n);
    }
  }

  float2 z[RF];

  // ---- load: rows u + RS*q, q = 0..RF-1 (spectre.py:506 zero-pads / truncates to n_fft) -------------
  {
    const char* vb = reinterpret_cast<const char*>(a.v) + ((size_t)b * a.v_sb + (size_t)ct * (2 * kPC)) * ES_IN;
    const uint32_t voff = (uint32_t)(((long long)u * v_sn + 2 * p) * ES_IN);
    // General modes: rows >= N_in (rfft's zero padding) and the lanes of a ragged last tile beyond D are the out-of-range case of the
    // buffer instructions — loads return 0, stores are dropped — instead of 2 * RF predicates, pointer selects and their spills.
    // The range check covers the VGPR offset only, so the row-block offset is added there; an invalid lane starts at 2^31.
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs_in;
    [[maybe_unused]] uint32_t voff_c = voff;
    if constexpr (GENERAL) {
      const int rows = a.N_in < N ? a.N_in : N;
      rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vb), 0, (int)((long long)rows * v_sn * ES_IN), kRsrcFlags);
      voff_c = cvalid ? voff : 0x80000000u;
    }
    static_for<0, RF>([&](auto ic) {
      // issue order = order of use: stage 1 of F1 works on {q0 + RBF*q1}, q0 = 0, 1, ..., so its first butterflies
      // start while the tail of the tile is still in flight
      constexpr int q = (decltype(ic)::value / RAF) + RBF * (decltype(ic)::value % RAF);
      if constexpr (GENERAL) {
        const uint32_t off = voff_c + (uint32_t)((long long)q * RS * v_sn * ES_IN);
        if constexpr (IN_BF16) {
          const uint32_t wv = __builtin_amdgcn_raw_buffer_load_b32(rs_in, off, 0, 0);
          z[q] = make_float2(__uint_as_float(wv << 16), __uint_as_float(wv & 0xffff0000u));
        } else {
          const rt_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs_in, off, 0, 0);
          z[q] = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
        }
      } else {
        const char* ptr = vb + (size_t)q * RS * v_sn * ES_IN + voff;
        if constexpr (IN_BF16) {
          const uint32_t wv = *reinterpret_cast<const uint32_t*>(ptr);
          z[q] = make_float2(__uint_as_float(wv << 16), __uint_as_float(wv & 0xffff0000u));
        } else {
          z[q] = *reinterpret_cast<const float2*>(ptr);
        }
      }
    });
  }

  // ---- F1: RF-point forward transform over n1, then W_N^(u*k1) ---------------------------------------
  {
    fftA_stage1<RAF, RBF, false>(z);
    float2 wa[RAF], wb[RBF];
    __builtin_amdgcn_sched_barrier(0);           // keep the base loads (and their registers) out of stage 1
    load_twiddle_bases(wa, wb);
    static_for<0, RAF>([&](auto kac) { fftA_stage2_group<RAF, RBF, false, decltype(kac)::value>(z); });
    static_for<1, RF>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int ka = j / RBF, kb = j % RBF;  // position j carries k1 = ka + RAF*kb
      if constexpr (ka > 0) z[j] = cmul(z[j], wa[ka]);
      if constexpr (kb > 0) z[j] = cmul(z[j], wb[kb]);
    });
  }

  // ---- E1: position j (k1 = ka + RAF*kb) -> image row k1, column (u, p); thread s reads rows s + RS*t ----
  {
    if constexpr (XV) {
      constexpr int PS = RS + 4, RW = kPC * PS;      // column stride, row stride (floats)
      exchange_planes_b128_w2<RF, RAS, RBS, true, RF, 1, 0, RW>(z, img, p * PS + u,
          [](auto rc, auto) { constexpr int k1 = decltype(rc)::value; return std::integral_constant<int, RBF * (k1 % RAF) + k1 / RAF>{}; },   // row k1 <- position
          [&](auto mc) { constexpr int m = decltype(mc)::value; constexpr int t = m / RS, n2 = m % RS;
                         return (u + RS * t) * RW + p * PS + n2; });
    } else {
      exchange_planes<RF, RAS, RBS>(z, img,
          [&](auto jc) { constexpr int j = decltype(jc)::value; constexpr int k1 = (j / RBF) + RAF * (j % RBF);
                         return k1 * ROW1 + u * kPC + p; },
          [&](auto mc) { constexpr int m = decltype(mc)::value; constexpr int t = m / RS, n2 = m % RS;
                         return (u + RS * t) * ROW1 + n2 * kPC + p; });
    }
  }

  // ---- middle: per set t (k1 = u + RS*t):  F2 stage 1, then per register group  F2 stage 2 -> gate -> I1 stage 1,
  //      then I1 stage 2.  Bin of register (ka, kb) of set t: k = k1 + RF*k2, k2 = ka + RAS*kb.  k2 >= RS/2 means
  //      k > N/2 (or k == N/2 when k1 == 0): the Hermitian extension reads conj(g[N - k]).
  {
    const int cg = cvalid ? c : 0;               // lanes beyond D compute on zeros; keep their addresses in range
    const int grp = cg / a.d_g;
    const float2* gp = a.gate + ((size_t)b * a.G + grp) * a.F;
    static_for<0, NS>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int OFF = t * RS;
      const int k1 = u + RS * t;
      fftA_stage1<RAS, RBS, false, OFF, RF>(z);
      auto gate_index = [&](int k2) { return (k2 >= RS / 2) ? RF * (RS - k2) - k1 : k1 + RF * k2; };
      auto fetch_gate = [&](int k2, bool upper, bool edge) -> float2 {
        float2 g;
        if constexpr (GATE_LDS) {
          g = glds[gate_index(k2)];                          // already scaled, edges fixed
        } else {
          g = gp[gate_index(k2)];
          if (a.conj_gate) g.y = -g.y;
          if (edge && k1 == 0) g.y = 0.f;                    // irfft ignores Im(DC), Im(Nyquist)
          g.x *= inv_n; g.y *= inv_n;
        }
        if (upper) g.y = -g.y;
        return g;
      };
      float2 gcur[RBS], gnxt[RBS];
      static_for<0, RBS>([&](auto kbc) {
        constexpr int k2 = RAS * decltype(kbc)::value;
        gcur[decltype(kbc)::value] = fetch_gate(k2, k2 >= RS / 2, k2 == 0 || k2 == RS / 2);
      });
      static_for<0, RAS>([&](auto kac) {
        constexpr int ka = decltype(kac)::value;
        if constexpr (ka + 1 < RAS && !WITH_MEM) {   // software prefetch of the next group's gate bins
          static_for<0, RBS>([&](auto kbc) {
            constexpr int k2n = ka + 1 + RAS * decltype(kbc)::value;
            gnxt[decltype(kbc)::value] = fetch_gate(k2n, k2n >= RS / 2, k2n == 0 || k2n == RS / 2);
          });
        }
        float4 mcur[WITH_MEM ? RBS : 1];
        if constexpr (WITH_MEM) {                    // memory_fft rows of this group: issued ahead of the butterfly that hides them
          static_for<0, RBS>([&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
            mcur[kb] = *reinterpret_cast<const float4*>(a.mem + ((size_t)gate_index(ka + RAS * kb) * a.D + cg) * 2);
          });
        }
        fftA_stage2_group<RAS, RBS, false, ka, OFF, RF>(z);
        static_for<0, RBS>([&](auto kbc) {
          constexpr int kb = decltype(kbc)::value;
          constexpr int j = OFF + RBS * ka + kb;
          constexpr int k2 = ka + RAS * kb;
          constexpr bool upper = k2 >= RS / 2;
          constexpr bool edge = (k2 == 0) || (k2 == RS / 2);
          z[j] = cmul(z[j], gcur[kb]);
          if constexpr (WITH_MEM) {                  // spectre.py:548-549
            const float4 m = mcur[kb];
            float2 add;
            if (edge && k1 == 0) add = make_float2(m.x, m.z);
            else if (upper)      add = make_float2(m.x + m.w, m.z - m.y);
            else                 add = make_float2(m.x - m.w, m.y + m.z);
            z[j].x += add.x * inv_n; z[j].y += add.y * inv_n;
          }
        });
        fftB_stage1_group<RAS, RBS, true, ka, OFF, RF>(z);
        if constexpr (ka + 1 < RAS) {
          if constexpr (WITH_MEM) {   // no double buffering next to the 4-register memory_fft loads
            static_for<0, RBS>([&](auto kbc) {
              constexpr int k2n = ka + 1 + RAS * decltype(kbc)::value;
              gcur[decltype(kbc)::value] = fetch_gate(k2n, k2n >= RS / 2, k2n == 0 || k2n == RS / 2);
            });
          } else {
            static_for<0, RBS>([&](auto kbc) { gcur[decltype(kbc)::value] = gnxt[decltype(kbc)::value]; });
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the gate prefetch one group deep (register budget)
      });
      fftB_stage2<RAS, RBS, true, OFF, RF>(z);   // natural order: position OFF + n2
    });
  }

  // ---- E2: position t*RS + n2 -> image row n2, column (k1 = u + RS*t, p); thread u reads its row, slot k1 ----
  {
    if constexpr (XV) {
      constexpr int PS = RF + 4, RW = kPC * PS;
      exchange_planes_b128_w2<RF, RAF, RBF, false, RS, RF / RS, RS, RW>(z, img, p * PS + u,
          [](auto rc, auto tc) { return std::integral_constant<int, decltype(rc)::value + RS * decltype(tc)::value>{}; },   // row n2, column offset RS t
          [&](auto mc) { constexpr int m = decltype(mc)::value; return u * RW + p * PS + m; });
    } else {
      exchange_planes<RF, RAF, RBF, false>(z, img,
          [&](auto jc) { constexpr int j = decltype(jc)::value; constexpr int t = j / RS, n2 = j % RS;
                         return n2 * ROW2 + (u + RS * t) * kPC + p; },
          [&](auto mc) { constexpr int m = decltype(mc)::value; return u * ROW2 + m * kPC + p; });
    }
  }

  // ---- conj twiddle, I2 and store (spectre.py:553 keeps rows < min(N, n_fft)) -----------------------------
  {
    float2 wa[RAF], wb[RBF];
    load_twiddle_bases(wa, wb);
    static_for<1, RF>([&](auto jc) {
      constexpr int j = decltype(jc)::value;     // position j carries k1 = j = ja + RAF*jb
      constexpr int ja = j % RAF, jb = j / RAF;
      if constexpr (ja > 0) z[j] = cmulc(z[j], wa[ja]);
      if constexpr (jb > 0) z[j] = cmulc(z[j], wb[jb]);
    });
    fftA_stage1<RAF, RBF, true>(z);
  }
  {
    char* ob = reinterpret_cast<char*>(a.out) + ((size_t)b * a.out_sb + (size_t)ct * (2 * kPC)) * ES_OUT;
    const uint32_t ooff = (uint32_t)(((long long)u * out_sn + 2 * p) * ES_OUT);
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs_out;
    [[maybe_unused]] uint32_t ooff_c = ooff;
    if constexpr (GENERAL) {
      const int rows = a.N_in < N ? a.N_in : N;                // spectre.py:553 keeps rows < min(N, n_fft)
      rs_out = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)((long long)rows * out_sn * ES_OUT), kRsrcFlags);
      ooff_c = cvalid ? ooff : 0x80000000u;
    }
    static_for<0, RF>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      // the last butterfly stage runs group by group; each group's rows are stored as soon as they exist
      if constexpr ((j % RBF) == 0) fftA_stage2_group<RAF, RBF, true, j / RBF>(z);
      constexpr int n1 = (j / RBF) + RAF * (j % RBF);
      if constexpr (GENERAL) {
        const uint32_t off = ooff_c + (uint32_t)((long long)n1 * RS * out_sn * ES_OUT);
        if constexpr (OUT_BF16) {
          __builtin_amdgcn_raw_buffer_store_b32(f32x2_to_bf16x2_rne(z[j].x, z[j].y), rs_out, off, 0, 0);
        } else {
          rt_u32x2 t;
          t.x = __float_as_uint(z[j].x); t.y = __float_as_uint(z[j].y);
          __builtin_amdgcn_raw_buffer_store_b64(t, rs_out, off, 0, 0);
        }
      } else {
        char* ptr = ob + (size_t)n1 * RS * out_sn * ES_OUT + ooff;
        if constexpr (OUT_BF16) {
          *reinterpret_cast<uint32_t*>(ptr) = f32x2_to_bf16x2_rne(z[j].x, z[j].y);
        } else {
          *reinterpret_cast<float2*>(ptr) = z[j];
        }
      }
    });
  }
  }  // tile loop
}

// host-side launcher for one (RF, RS) (defined in regtile_*.hip so the heavy kernels compile in parallel)
template <int RF, int RS>
hipError_t launch_regtile(const RegtileArgs& a, bool in_bf16, bool out_bf16, int mode, hipStream_t stream);

#define SFFT_DEFINE_REGTILE_LAUNCHER(RF_, RS_)                                                               \
  template <>                                                                                                \
  hipError_t launch_regtile<RF_, RS_>(const RegtileArgs& a, bool in_bf16, bool out_bf16, int mode,           \
                                      hipStream_t stream) {                                                  \
    if (mode != 0 && !in_bf16 && out_bf16) return hipErrorInvalidValue;   /* f32 -> bf16: fast mode only */   \
    const dim3 grid(a.n_wg), block(regtile_threads<RF_, RS_>());                                             \
    const size_t lds = regtile_lds_total<RF_, RS_>();                                                        \
    const int key = (in_bf16 ? 16 : 0) | (out_bf16 ? 8 : 0) | mode;                                          \
    static std::atomic<bool> lds_opt_in[16][32];   /* [device][variant]: >64 KiB of dynamic LDS needs a one-time opt-in */ \
    auto go = [&](auto kern) -> hipError_t {                                                                 \
      int dev = 0;                                                                                           \
      (void)hipGetDevice(&dev);                                                                              \
      if (dev < 0 || dev >= 16 || !lds_opt_in[dev][key]) {                                                   \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e != hipSuccess) return e;                                                                       \
        if (dev >= 0 && dev < 16) lds_opt_in[dev][key] = true;                                               \
      }                                                                                                      \
      hipLaunchKernelGGL(kern, grid, block, lds, stream, a);                                                 \
      return hipGetLastError();                                                                              \
    };                                                                                                       \
    switch (key) {                                                                                           \
      case 0: return go(spectre_mix_regtile<RF_, RS_, false, false, 0>);                                     \
      case 1: return go(spectre_mix_regtile<RF_, RS_, false, false, 1>);                                     \
      case 2: return go(spectre_mix_regtile<RF_, RS_, false, false, 2>);                                     \
      case 3: return go(spectre_mix_regtile<RF_, RS_, false, false, 3>);                                     \
      case 4: return go(spectre_mix_regtile<RF_, RS_, false, false, 4>);                                     \
      case 8: return go(spectre_mix_regtile<RF_, RS_, false, true, 0>);                                      \
      case 16: return go(spectre_mix_regtile<RF_, RS_, true, false, 0>);                                     \
      case 17: return go(spectre_mix_regtile<RF_, RS_, true, false, 1>);   /* bf16 rows in, fp32 rows out   */ \
      case 18: return go(spectre_mix_regtile<RF_, RS_, true, false, 2>);   /* (activations under autocast): */ \
      case 19: return go(spectre_mix_regtile<RF_, RS_, true, false, 3>);   /* every mode, not only the fast */ \
      case 20: return go(spectre_mix_regtile<RF_, RS_, true, false, 4>);   /* one                           */ \
      case 24: return go(spectre_mix_regtile<RF_, RS_, true, true, 0>);                                      \
      case 25: return go(spectre_mix_regtile<RF_, RS_, true, true, 1>);                                      \
      case 26: return go(spectre_mix_regtile<RF_, RS_, true, true, 2>);                                      \
      case 27: return go(spectre_mix_regtile<RF_, RS_, true, true, 3>);                                      \
      case 28: return go(spectre_mix_regtile<RF_, RS_, true, true, 4>);                                      \
      default: return hipErrorInvalidValue;                                                                  \
    }                                                                                                        \
  }

}  // namespace sfft
