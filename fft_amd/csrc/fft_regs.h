// fft_regs.h — register-resident small FFTs for gfx950 wavefronts (one transform per lane).
//
// Building blocks of the spectral-mix kernels: radix-2/4/8 butterflies and two-factor (RA x RB)
// in-register transforms of length 16/32/64 whose every array index is a compile-time constant, so the
// whole working set lives in VGPRs (no scratch, §5.4 rule 20 of the CDNA guide).
//
// Two data-flow types (both compute the same DFT; they differ in where inputs/outputs sit):
//   type A : input index q at position q            -> output index k = ka + RA*kb at position RB*ka + kb
//   type B : input index k = ka + RA*kb at position RB*ka + kb  -> output index n at position n
// so  A (forward) -> pointwise filter -> B (inverse)  needs no reordering in between, and the second
// stage of A / first stage of B act on the same RB-element register groups (fused in the kernel).
//
// Sign convention: INV=false multiplies by exp(-2 pi i ...), INV=true by exp(+2 pi i ...), no scaling.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace sfft {

// cos / sin of 2*pi*m/64, m = 0..63, rounded once from double
constexpr double kCos64[64] = {
    1, 0.99518472667219693, 0.98078528040323043, 0.95694033573220882,
    0.92387953251128674, 0.88192126434835505, 0.83146961230254524, 0.77301045336273699,
    0.70710678118654757, 0.63439328416364549, 0.55557023301960229, 0.47139673682599781,
    0.38268343236508984, 0.29028467725446233, 0.19509032201612833, 0.09801714032956077,
    0, -0.098017140329560645, -0.19509032201612819, -0.29028467725446216,
    -0.38268343236508973, -0.4713967368259977, -0.55557023301960196, -0.63439328416364538,
    -0.70710678118654746, -0.77301045336273699, -0.83146961230254535, -0.88192126434835494,
    -0.92387953251128674, -0.95694033573220882, -0.98078528040323043, -0.99518472667219682,
    -1, -0.99518472667219693, -0.98078528040323043, -0.95694033573220894,
    -0.92387953251128685, -0.88192126434835505, -0.83146961230254546, -0.7730104533627371,
    -0.70710678118654768, -0.63439328416364593, -0.55557023301960218, -0.47139673682599786,
    -0.38268343236509034, -0.29028467725446244, -0.19509032201612866, -0.098017140329560451,
    0, 0.09801714032956009, 0.1950903220161283, 0.29028467725446205,
    0.38268343236509, 0.47139673682599759, 0.55557023301960184, 0.6343932841636456,
    0.70710678118654735, 0.77301045336273666, 0.83146961230254524, 0.88192126434835483,
    0.92387953251128652, 0.95694033573220882, 0.98078528040323032, 0.99518472667219693,
};
constexpr double kSin64[64] = {
    0, 0.098017140329560604, 0.19509032201612825, 0.29028467725446233,
    0.38268343236508978, 0.47139673682599764, 0.55557023301960218, 0.63439328416364549,
    0.70710678118654746, 0.77301045336273699, 0.83146961230254524, 0.88192126434835494,
    0.92387953251128674, 0.95694033573220894, 0.98078528040323043, 0.99518472667219682,
    1, 0.99518472667219693, 0.98078528040323043, 0.95694033573220894,
    0.92387953251128674, 0.88192126434835505, 0.83146961230254546, 0.7730104533627371,
    0.70710678118654757, 0.63439328416364549, 0.55557023301960218, 0.47139673682599786,
    0.38268343236508989, 0.29028467725446239, 0.19509032201612861, 0.098017140329560826,
    0, -0.09801714032956059, -0.19509032201612836, -0.29028467725446211,
    -0.38268343236508967, -0.47139673682599764, -0.55557023301960196, -0.63439328416364527,
    -0.70710678118654746, -0.77301045336273666, -0.83146961230254524, -0.88192126434835494,
    -0.92387953251128652, -0.95694033573220882, -0.98078528040323032, -0.99518472667219693,
    -1, -0.99518472667219693, -0.98078528040323043, -0.95694033573220894,
    -0.92387953251128663, -0.88192126434835505, -0.83146961230254546, -0.77301045336273688,
    -0.70710678118654768, -0.63439328416364593, -0.55557023301960218, -0.47139673682599792,
    -0.38268343236509039, -0.2902846772544625, -0.19509032201612872, -0.098017140329560506,
};

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * b
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
  return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}

// a * W_64^M (INV=false) or a * conj(W_64^M) (INV=true), W_64 = exp(-2 pi i / 64); M is a constant.
template <int M, bool INV>
__device__ __forceinline__ float2 twid64(float2 a) {
  constexpr int m = INV ? ((64 - (M % 64)) % 64) : (((M % 64) + 64) % 64);   // conj(W^M) = W^(64-M)
  if constexpr (m == 0) {
    return a;
  } else if constexpr (m == 16) {        // * (-i)
    return make_float2(a.y, -a.x);
  } else if constexpr (m == 32) {        // * (-1)
    return make_float2(-a.x, -a.y);
  } else if constexpr (m == 48) {        // * (+i)
    return make_float2(-a.y, a.x);
  } else {                               // (x + iy)(c - is)
    constexpr float c = (float)kCos64[m];
    constexpr float s = (float)kSin64[m];
    return make_float2(a.x * c + a.y * s, a.y * c - a.x * s);
  }
}

// ---- butterflies: natural-order in, natural-order out, in place on references -------------------
template <bool INV>
__device__ __forceinline__ void bfly2(float2& a0, float2& a1) {
  const float2 t = csub(a0, a1);
  a0 = cadd(a0, a1);
  a1 = t;
}

template <bool INV>
__device__ __forceinline__ void bfly4(float2& a0, float2& a1, float2& a2, float2& a3) {
  const float2 t0 = cadd(a0, a2), t1 = csub(a0, a2);
  const float2 t2 = cadd(a1, a3), t3 = twid64<16, INV>(csub(a1, a3));
  a0 = cadd(t0, t2);
  a1 = cadd(t1, t3);
  a2 = csub(t0, t2);
  a3 = csub(t1, t3);
}

template <bool INV>
__device__ __forceinline__ void bfly8(float2& a0, float2& a1, float2& a2, float2& a3,
                                      float2& a4, float2& a5, float2& a6, float2& a7) {
  // decimation in frequency: even outputs = DFT4(a_j + a_{j+4}), odd outputs = DFT4((a_j - a_{j+4}) W_8^j)
  float2 s0 = cadd(a0, a4), d0 = csub(a0, a4);
  float2 s1 = cadd(a1, a5), d1 = twid64<8, INV>(csub(a1, a5));
  float2 s2 = cadd(a2, a6), d2 = twid64<16, INV>(csub(a2, a6));
  float2 s3 = cadd(a3, a7), d3 = twid64<24, INV>(csub(a3, a7));
  bfly4<INV>(s0, s1, s2, s3);
  bfly4<INV>(d0, d1, d2, d3);
  a0 = s0; a2 = s1; a4 = s2; a6 = s3;
  a1 = d0; a3 = d1; a5 = d2; a7 = d3;
}

// radix-R butterfly over z[BASE + STRIDE*j], j = 0..R-1
template <int R, bool INV, int BASE, int STRIDE, int NTOT>
__device__ __forceinline__ void bfly(float2 (&z)[NTOT]) {
  static_assert(R == 2 || R == 4 || R == 8, "radix");
  if constexpr (R == 2) {
    bfly2<INV>(z[BASE], z[BASE + STRIDE]);
  } else if constexpr (R == 4) {
    bfly4<INV>(z[BASE], z[BASE + STRIDE], z[BASE + 2 * STRIDE], z[BASE + 3 * STRIDE]);
  } else {
    bfly8<INV>(z[BASE], z[BASE + STRIDE], z[BASE + 2 * STRIDE], z[BASE + 3 * STRIDE],
               z[BASE + 4 * STRIDE], z[BASE + 5 * STRIDE], z[BASE + 6 * STRIDE], z[BASE + 7 * STRIDE]);
  }
}

// ---- type A (on the sub-array z[OFF .. OFF + RA*RB) of an NTOT-element register array) --------------
// stage 1: for every q0, radix-RA over q1 (positions RB*q1 + q0), then * W_R^(q0*ka)
template <int RA, int RB, bool INV, int OFF = 0, int NTOT = RA * RB>
__device__ __forceinline__ void fftA_stage1(float2 (&z)[NTOT]) {
  constexpr int R = RA * RB, U = 64 / R;   // W_R = W_64^U
  static_for<0, RB>([&](auto q0c) {
    constexpr int q0 = decltype(q0c)::value;
    bfly<RA, INV, OFF + q0, RB, NTOT>(z);
    static_for<1, RA>([&](auto kac) {
      constexpr int ka = decltype(kac)::value;
      z[OFF + RB * ka + q0] = twid64<U * q0 * ka, INV>(z[OFF + RB * ka + q0]);
    });
  });
}
// stage 2 for one ka: radix-RB over q0 (positions RB*ka + q0) -> kb at RB*ka + kb
template <int RA, int RB, bool INV, int KA, int OFF = 0, int NTOT = RA * RB>
__device__ __forceinline__ void fftA_stage2_group(float2 (&z)[NTOT]) {
  bfly<RB, INV, OFF + RB * KA, 1, NTOT>(z);
}
template <int RA, int RB, bool INV, int OFF = 0, int NTOT = RA * RB>
__device__ __forceinline__ void fftA(float2 (&z)[NTOT]) {
  fftA_stage1<RA, RB, INV, OFF, NTOT>(z);
  static_for<0, RA>([&](auto kac) { fftA_stage2_group<RA, RB, INV, decltype(kac)::value, OFF, NTOT>(z); });
}

// ---- type B -------------------------------------------------------------------------------------
// stage 1 for one ka: radix-RB over kb (positions RB*ka + kb) -> n_lo, then * W_R^(ka*n_lo)
template <int RA, int RB, bool INV, int KA, int OFF = 0, int NTOT = RA * RB>
__device__ __forceinline__ void fftB_stage1_group(float2 (&z)[NTOT]) {
  constexpr int R = RA * RB, U = 64 / R;
  bfly<RB, INV, OFF + RB * KA, 1, NTOT>(z);
  if constexpr (KA > 0) {
    static_for<1, RB>([&](auto nc) {
      constexpr int nlo = decltype(nc)::value;
      z[OFF + RB * KA + nlo] = twid64<U * KA * nlo, INV>(z[OFF + RB * KA + nlo]);
    });
  }
}
// stage 2: for every n_lo, radix-RA over ka (positions RB*ka + n_lo) -> n_hi at RB*n_hi + n_lo
template <int RA, int RB, bool INV, int OFF = 0, int NTOT = RA * RB>
__device__ __forceinline__ void fftB_stage2(float2 (&z)[NTOT]) {
  static_for<0, RB>([&](auto nc) { bfly<RA, INV, OFF + decltype(nc)::value, RB, NTOT>(z); });
}

// factorisation used for each in-register length
template <int X> struct FftCfg;
template <> struct FftCfg<64> { static constexpr int RA = 8, RB = 8; };
template <> struct FftCfg<32> { static constexpr int RA = 4, RB = 8; };
template <> struct FftCfg<16> { static constexpr int RA = 4, RB = 4; };

}  // namespace sfft
