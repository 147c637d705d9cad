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

// ---- butterflies with TWIDDLED INPUTS in scaled (Linzer-Feig) form -------------------------------------------------------------------
// The twiddle between the two stages of an RA x RB transform is applied to the INPUTS of the second stage's butterflies, as
//   a * W_64^m = cos(t) * (-i)^q * ( (x + tan(t) y) + i (y - tan(t) x) ),   m = 16 q + r, |r| <= 8, t = 2 pi r / 64:
// the quadrant (-i)^q is a free relabelling, the rotation costs 2 fused multiply-adds instead of the 4 instructions of a complex
// multiplication, and the scale cos(t) (in [0.707, 1]: the tangent never exceeds 1) is not applied at all: every addition of the
// butterfly that follows becomes a +- rho * b with rho = the ratio of the two operands' pending scales, a compile-time constant, i.e. the
// same number of instructions as the plain additions.  The first input of every such butterfly has m = 0, so the result carries no
// scale.  The (1 +- i) / sqrt 2 inside a radix-8 butterfly is treated the same way (two additions, 1 / sqrt 2 pending).
// A 64-point transform: 768 additions + 160 other instructions instead of 768 + ~290 (profiles/r05_isa_census.txt: tools/isa_census.py).
template <int M, bool INV>
struct TwSplit {
  static constexpr int m = INV ? ((64 - (((M % 64) + 64) % 64)) % 64) : (((M % 64) + 64) % 64);   // conj(W^M) = W^(64 - M)
  static constexpr int q = ((m + 8) >> 4) & 3;                  // W^m = (-i)^q W^r
  static constexpr int r = m - 16 * ((m + 8) >> 4);             // -8 .. 7
  static constexpr double c = kCos64[(r + 64) % 64];
  static constexpr double t = kSin64[(r + 64) % 64] / kCos64[(r + 64) % 64];
};
// a compile-time scale factor as a type: cos of the reduced twiddle angle, optionally times 1 / sqrt 2
template <int M, bool INV, bool HALF = false>
struct TwScale { static constexpr double v = TwSplit<M, INV>::c * (HALF ? 0.70710678118654752440 : 1.0); };

template <int Q>
__device__ __forceinline__ float2 quadrant(float2 a) {          // (-i)^Q a
  if constexpr (Q == 0) return a;
  else if constexpr (Q == 1) return make_float2(a.y, -a.x);
  else if constexpr (Q == 2) return make_float2(-a.x, -a.y);
  else return make_float2(-a.y, a.x);
}
// u with a * W^M = TwScale<M>::v * u
template <int M, bool INV>
__device__ __forceinline__ float2 tw_unscaled(float2 a) {
  using T = TwSplit<M, INV>;
  const float2 b = quadrant<T::q>(a);
  if constexpr (T::r == 0) {
    return b;
  } else {
    constexpr float t = (float)T::t;
    return make_float2(__builtin_fmaf(t, b.y, b.x), __builtin_fmaf(-t, b.x, b.y));
  }
}
// a + rho b and a - rho b with rho = SB::v / SA::v (the result carries SA's scale)
template <class SA, class SB>
__device__ __forceinline__ float2 sadd(float2 a, float2 b) {
  constexpr double rho = SB::v / SA::v;
  if constexpr (rho == 1.0) return cadd(a, b);
  else { constexpr float r = (float)rho; return make_float2(__builtin_fmaf(r, b.x, a.x), __builtin_fmaf(r, b.y, a.y)); }
}
template <class SA, class SB>
__device__ __forceinline__ float2 ssub(float2 a, float2 b) {
  constexpr double rho = SB::v / SA::v;
  if constexpr (rho == 1.0) return csub(a, b);
  else { constexpr float r = (float)rho; return make_float2(__builtin_fmaf(-r, b.x, a.x), __builtin_fmaf(-r, b.y, a.y)); }
}
// radix-4 butterfly of inputs with pending scales S0..S3; the outputs carry S0
template <bool INV, class S0, class S1, class S2, class S3>
__device__ __forceinline__ void bfly4_s(float2& a0, float2& a1, float2& a2, float2& a3) {
  const float2 t0 = sadd<S0, S2>(a0, a2), t1 = ssub<S0, S2>(a0, a2);                       // scale S0
  const float2 t2 = sadd<S1, S3>(a1, a3), t3 = twid64<16, INV>(ssub<S1, S3>(a1, a3));      // scale S1
  a0 = sadd<S0, S1>(t0, t2);
  a1 = sadd<S0, S1>(t1, t3);
  a2 = ssub<S0, S1>(t0, t2);
  a3 = ssub<S0, S1>(t1, t3);
}
// radix-4 butterfly of a_j * W_64^(Mj), M0 = 0
template <bool INV, int M1, int M2, int M3>
__device__ __forceinline__ void bfly4_tw(float2& a0, float2& a1, float2& a2, float2& a3) {
  a1 = tw_unscaled<M1, INV>(a1); a2 = tw_unscaled<M2, INV>(a2); a3 = tw_unscaled<M3, INV>(a3);
  bfly4_s<INV, TwScale<0, INV>, TwScale<M1, INV>, TwScale<M2, INV>, TwScale<M3, INV>>(a0, a1, a2, a3);
}
// radix-8 butterfly of a_j * W_64^(Mj), M0 = 0 (decimation in frequency, outputs in natural order like bfly8)
template <bool INV, int M1, int M2, int M3, int M4, int M5, int M6, int M7>
__device__ __forceinline__ void bfly8_tw(float2& a0, float2& a1, float2& a2, float2& a3, float2& a4, float2& a5, float2& a6, float2& a7) {
  using C0 = TwScale<0, INV>;  using C1 = TwScale<M1, INV>; using C2 = TwScale<M2, INV>; using C3 = TwScale<M3, INV>;
  using C4 = TwScale<M4, INV>; using C5 = TwScale<M5, INV>; using C6 = TwScale<M6, INV>; using C7 = TwScale<M7, INV>;
  const float2 u1 = tw_unscaled<M1, INV>(a1), u2 = tw_unscaled<M2, INV>(a2), u3 = tw_unscaled<M3, INV>(a3), u4 = tw_unscaled<M4, INV>(a4),
               u5 = tw_unscaled<M5, INV>(a5), u6 = tw_unscaled<M6, INV>(a6), u7 = tw_unscaled<M7, INV>(a7);
  float2 s0 = sadd<C0, C4>(a0, u4), d0 = ssub<C0, C4>(a0, u4);          // scale C0
  float2 s1 = sadd<C1, C5>(u1, u5), e1 = ssub<C1, C5>(u1, u5);          // scale C1
  float2 s2 = sadd<C2, C6>(u2, u6), e2 = ssub<C2, C6>(u2, u6);          // scale C2
  float2 s3 = sadd<C3, C7>(u3, u7), e3 = ssub<C3, C7>(u3, u7);          // scale C3
  // (a_j - a_{j+4}) W_8^j: j = 1, 3 leave 1 / sqrt 2 pending
  float2 d1 = INV ? make_float2(e1.x - e1.y, e1.x + e1.y) : make_float2(e1.x + e1.y, e1.y - e1.x);            // * (1 -+ i)
  float2 d2 = twid64<16, INV>(e2);                                                                            // * (-+ i)
  float2 d3 = INV ? make_float2(-(e3.x + e3.y), e3.x - e3.y) : make_float2(e3.y - e3.x, -(e3.x + e3.y));      // * (-1 -+ i)
  bfly4_s<INV, C0, C1, C2, C3>(s0, s1, s2, s3);
  bfly4_s<INV, C0, TwScale<M1, INV, true>, C2, TwScale<M3, INV, true>>(d0, d1, d2, d3);
  a0 = s0; a2 = s1; a4 = s2; a6 = s3;
  a1 = d0; a3 = d1; a5 = d2; a7 = d3;
}
// radix-R butterfly over z[BASE + STRIDE*j] whose input j still needs the twiddle W_64^(U * KA * j)
template <int R, bool INV, int BASE, int STRIDE, int U, int KA, int NTOT>
__device__ __forceinline__ void bfly_tw(float2 (&z)[NTOT]) {
  static_assert(R == 4 || R == 8, "radix");
  constexpr int E = U * KA;
  if constexpr (R == 4) {
    bfly4_tw<INV, E, 2 * E, 3 * E>(z[BASE], z[BASE + STRIDE], z[BASE + 2 * STRIDE], z[BASE + 3 * STRIDE]);
  } else {
    bfly8_tw<INV, E, 2 * E, 3 * E, 4 * E, 5 * E, 6 * E, 7 * E>(z[BASE], z[BASE + STRIDE], z[BASE + 2 * STRIDE], z[BASE + 3 * STRIDE],
                                                                z[BASE + 4 * STRIDE], z[BASE + 5 * STRIDE], z[BASE + 6 * STRIDE], z[BASE + 7 * STRIDE]);
  }
}
// the same without twiddles (radix 2 has no constants at all)
template <int R, bool INV, int BASE, int STRIDE, int NTOT>
__device__ __forceinline__ void bfly_plain(float2 (&z)[NTOT]) {
  if constexpr (R == 2) bfly<2, INV, BASE, STRIDE, NTOT>(z);
  else bfly_tw<R, INV, BASE, STRIDE, 0, 0, NTOT>(z);
}

// ---- type A (on the sub-array z[OFF .. OFF + RA*RB) of an NTOT-element register array) --------------
// stage 1: for every q0, radix-RA over q1 (positions RB*q1 + q0) -> ka at RB*ka + q0.  The twiddle W_R^(q0*ka) is applied by stage 2,
// on the inputs of its butterflies (scaled form, above): between the two stages the values are NOT yet twiddled.
template <int RA, int RB, bool INV, int OFF = 0, int NTOT = RA * RB>
__device__ __forceinline__ void fftA_stage1(float2 (&z)[NTOT]) {
  static_for<0, RB>([&](auto q0c) { bfly_plain<RA, INV, OFF + decltype(q0c)::value, RB, NTOT>(z); });
}
// stage 2 for one ka: twiddle W_R^(q0*ka) + radix-RB over q0 (positions RB*ka + q0) -> kb at RB*ka + kb
template <int RA, int RB, bool INV, int KA, int OFF = 0, int NTOT = RA * RB>
__device__ __forceinline__ void fftA_stage2_group(float2 (&z)[NTOT]) {
  constexpr int R = RA * RB, U = 64 / R;   // W_R = W_64^U
  bfly_tw<RB, INV, OFF + RB * KA, 1, U, KA, NTOT>(z);
}
template <int RA, int RB, bool INV, int OFF = 0, int NTOT = RA * RB>
__device__ __forceinline__ void fftA(float2 (&z)[NTOT]) {
  fftA_stage1<RA, RB, INV, OFF, NTOT>(z);
  static_for<0, RA>([&](auto kac) { fftA_stage2_group<RA, RB, INV, decltype(kac)::value, OFF, NTOT>(z); });
}

// ---- type B -------------------------------------------------------------------------------------
// stage 1 for one ka: radix-RB over kb (positions RB*ka + kb) -> n_lo (the twiddle W_R^(ka*n_lo) is applied by stage 2)
template <int RA, int RB, bool INV, int KA, int OFF = 0, int NTOT = RA * RB>
__device__ __forceinline__ void fftB_stage1_group(float2 (&z)[NTOT]) {
  bfly_plain<RB, INV, OFF + RB * KA, 1, NTOT>(z);
}
// stage 2 for one n_lo: twiddle W_R^(ka*n_lo) + radix-RA over ka (positions RB*ka + n_lo) -> n_hi at RB*n_hi + n_lo
template <int RA, int RB, bool INV, int NLO, int OFF = 0, int NTOT = RA * RB>
__device__ __forceinline__ void fftB_stage2_group(float2 (&z)[NTOT]) {
  constexpr int R = RA * RB, U = 64 / R;
  bfly_tw<RA, INV, OFF + NLO, RB, U, NLO, NTOT>(z);
}
template <int RA, int RB, bool INV, int OFF = 0, int NTOT = RA * RB>
__device__ __forceinline__ void fftB_stage2(float2 (&z)[NTOT]) {
  static_for<0, RB>([&](auto nc) { fftB_stage2_group<RA, RB, INV, decltype(nc)::value, OFF, NTOT>(z); });
}

// factorisation used for each in-register length
template <int X> struct FftCfg;
template <> struct FftCfg<64> { static constexpr int RA = 8, RB = 8; };
template <> struct FftCfg<32> { static constexpr int RA = 4, RB = 8; };
template <> struct FftCfg<16> { static constexpr int RA = 4, RB = 4; };

}  // namespace sfft
