// fft_regs_mixed_scaled.h — EXPERIMENT (round 3, not part of the product): fft_amd/csrc/fft_regs_mixed.h with the compile-time twiddles in
// scaled (Linzer-Feig) form.  10 % fewer fp instructions at 60 x 50, +-1 % as a kernel, +5..30 % where it pushes a 128-VGPR kernel over its
// cap (profiles/r03_mixed_engine_ab.log).  tests/test_fft_engine_cpu.py checks it on the host next to the shipped engine.
//
// compile-time mixed-radix FFTs on register arrays (one transform per lane), gfx950.
//
// Generalises fft_regs.h's two-factor "type A" transform to any length R = RA * RB * ... built from the primitive
// butterflies 2, 3, 4, 5, 7, 8 (e.g. 60 = 4 x (3 x 5), 50 = 2 x (5 x 5)).  Every register index is a compile-time
// constant: the data never moves to undo a digit reversal; instead a transform takes a MAP (logical index ->
// physical register) for its input and publishes out_pos<R>(k), the logical slot where output k is left:
//
//   input  : logical index i  lives in  z[Map::at(i)]
//   output : DFT bin k        lives in  z[Map::at(out_pos<R>(k))]
//
// so a following transform (or an LDS exchange, or a store) just composes maps.  Twiddles inside a transform are
// literals from fft_tables_mixed.h.  Sign: INV=false multiplies by exp(-2 pi i ...), INV=true by exp(+2 pi i ...).
#pragma once
#include "../fft_amd/csrc/fft_regs.h"
#include "../fft_amd/csrc/fft_tables_mixed.h"

namespace sfft {

// ---- factorisation: R = RA * RB with RA primitive; RB == 1 marks a primitive length ---------------------------------
template <int R> struct Split { static constexpr int RA = R, RB = 1; static_assert(R == 2 || R == 3 || R == 4 || R == 5 || R == 7 || R == 8, "primitive"); };
template <> struct Split<6>  { static constexpr int RA = 2, RB = 3; };
template <> struct Split<10> { static constexpr int RA = 2, RB = 5; };
template <> struct Split<12> { static constexpr int RA = 4, RB = 3; };
template <> struct Split<14> { static constexpr int RA = 2, RB = 7; };
template <> struct Split<15> { static constexpr int RA = 3, RB = 5; };
template <> struct Split<16> { static constexpr int RA = 4, RB = 4; };
template <> struct Split<20> { static constexpr int RA = 4, RB = 5; };
template <> struct Split<24> { static constexpr int RA = 8, RB = 3; };
template <> struct Split<25> { static constexpr int RA = 5, RB = 5; };
template <> struct Split<30> { static constexpr int RA = 2, RB = 15; };
template <> struct Split<32> { static constexpr int RA = 4, RB = 8; };
template <> struct Split<40> { static constexpr int RA = 8, RB = 5; };
template <> struct Split<48> { static constexpr int RA = 8, RB = 6; };
template <> struct Split<50> { static constexpr int RA = 2, RB = 25; };
template <> struct Split<56> { static constexpr int RA = 8, RB = 7; };
template <> struct Split<60> { static constexpr int RA = 4, RB = 15; };
template <> struct Split<64> { static constexpr int RA = 8, RB = 8; };

// logical slot (relative to the transform's input map) where output bin k is left
template <int R> constexpr int out_pos(int k) {
  if constexpr (Split<R>::RB == 1) return k;
  else return Split<R>::RB * (k % Split<R>::RA) + out_pos<Split<R>::RB>(k / Split<R>::RA);
}

// ---- maps -----------------------------------------------------------------------------------------------------------
struct IdentityMap { static constexpr int at(int i) { return i; } };
template <class Parent, int BASE, int STRIDE> struct SubMap { static constexpr int at(int i) { return Parent::at(BASE + STRIDE * i); } };
// input of a transform that consumes the output of an R-point transform bin by bin: logical k -> out_pos<R>(k)
template <int R, class Parent = IdentityMap> struct OutPosMap { static constexpr int at(int k) { return Parent::at(out_pos<R>(k)); } };

// ---- pending scales ----------------------------------------------------------------------------------------------------
// The twiddles between the stages of a composite length are applied in the scaled (Linzer-Feig) form of fft_regs.h:
//   a * W = cos(t) * (-i)^q * ((x + tan(t) y) + i (y - tan(t) x)),   |t| <= pi/4,
// two fused multiply-adds; the factor cos(t) stays PENDING on the value — a compile-time constant the consuming butterfly folds into its
// additions (a + rho b with rho the ratio of two pending scales costs what a + b costs) and into the constants of the radix-3/5/7
// butterflies.  A transform's contract: input logical i is z[Map::at(i)] * Scale::at(i); every output carries Scale::at(0).  The
// outermost call passes UnitScale, and input 0 of every stage-2 transform has twiddle 1, so results are exact DFT bins again.
struct UnitScale { static constexpr double at(int) { return 1.0; } };
template <class Parent, int BASE, int STRIDE> struct SubScale { static constexpr double at(int i) { return Parent::at(BASE + STRIDE * i); } };
template <class Scale, int I> struct ScaleAt { static constexpr double v = Scale::at(I); };

// W_R^M (forward) / conj(W_R^M) (INV) = (-i)^q * c * (1 - i t): quarter turn q, cos c and tangent t of the remaining angle
template <int R> constexpr double tw_cos(int m) { if constexpr (64 % R == 0) return kCos64[(64 / R) * m]; else return TwTab<R>::c[m]; }
template <int R> constexpr double tw_sin(int m) { if constexpr (64 % R == 0) return kSin64[(64 / R) * m]; else return TwTab<R>::s[m]; }
struct TwParts { int q; bool quarter; double c, t; };
template <int R, bool INV>
constexpr TwParts tw_parts(int M) {
  const int m = INV ? ((R - (M % R)) % R) : (M % R);
  if ((4 * m) % R == 0) return TwParts{(4 * m) / R, true, 1.0, 0.0};       // exactly (-i)^(4m/R)
  const double c0 = tw_cos<R>(m), s0 = tw_sin<R>(m);                         // W^m = c0 - i s0
  const double ac = c0 < 0 ? -c0 : c0, as = s0 < 0 ? -s0 : s0;
  const int q = (c0 > 0 && as <= c0) ? 0 : ((s0 > 0 && ac < s0) ? 1 : ((c0 < 0 && as <= ac) ? 2 : 3));
  const double c = q == 0 ? c0 : (q == 1 ? s0 : (q == 2 ? -c0 : -s0));
  const double sn = q == 0 ? s0 : (q == 1 ? -c0 : (q == 2 ? -s0 : c0));
  return TwParts{q, false, c, sn / c};
}
template <int R, int M, bool INV>
struct TwGen {
  static constexpr TwParts p = tw_parts<R, INV>(M);
  static constexpr int q = p.q;
  static constexpr bool quarter = p.quarter;
  static constexpr double c = p.c, t = p.t;
};
// u with a * W_R^M = TwGen<R, M, INV>::c * u
template <int R, int M, bool INV>
__device__ __forceinline__ float2 tw_unscaled_gen(float2 a) {
  using T = TwGen<R, M, INV>;
  const float2 b = quadrant<T::q>(a);
  if constexpr (T::quarter) {
    return b;
  } else {
    constexpr float t = (float)T::t;
    return make_float2(__builtin_fmaf(t, b.y, b.x), __builtin_fmaf(-t, b.x, b.y));
  }
}
// stage-2 inputs of an R = RA x RB transform for one ka: the stage-1 sub-transform over q1 left Scale::at(q0) on (q0, ka), the twiddle
// W_R^(q0 ka) adds its cosine
template <class Parent, int R, int KA, bool INV>
struct Stage2Scale { static constexpr double at(int q0) { return Parent::at(q0) * tw_parts<R, INV>(q0 * KA).c; } };

// ---- primitive butterflies with pending input scales S0..; the outputs carry S0 ------------------------------------------
// (with every scale 1 these are the plain butterflies: sadd / ssub fall back to additions, the constants to the textbook ones)
// a + k b with a compile-time k
__device__ __forceinline__ float2 caxpy(float k, float2 b, float2 a) { return make_float2(__builtin_fmaf(k, b.x, a.x), __builtin_fmaf(k, b.y, a.y)); }

template <bool INV, class S0, class S1>
__device__ __forceinline__ void bfly2_s(float2& v0, float2& v1) {
  const float2 t = ssub<S0, S1>(v0, v1);
  v0 = sadd<S0, S1>(v0, v1);
  v1 = t;
}
template <bool INV, class S0, class S1, class S2>
__device__ __forceinline__ void bfly3_s(float2& v0, float2& v1, float2& v2) {
  constexpr double r1 = S1::v / S0::v;
  constexpr float s = (float)((INV ? -0.86602540378443865 : 0.86602540378443865) * r1);   // forward W3 = -1/2 - i sqrt(3)/2
  const float2 t = sadd<S1, S2>(v1, v2);               // scale S1
  const float2 d = ssub<S1, S2>(v1, v2);
  const float2 m = caxpy((float)(-0.5 * r1), t, v0);
  const float2 r = make_float2(s * d.y, -s * d.x);     // -i s d
  v0 = sadd<S0, S1>(v0, t);
  v1 = cadd(m, r);
  v2 = csub(m, r);
}
template <bool INV, class S0, class S1, class S2, class S3, class S4>
__device__ __forceinline__ void bfly5_s(float2& v0, float2& v1, float2& v2, float2& v3, float2& v4) {
  constexpr double r1 = S1::v / S0::v, r2 = S2::v / S0::v;
  constexpr double c1 = 0.30901699437494742, c2 = -0.80901699437494742;       // cos(2pi/5), cos(4pi/5)
  constexpr double s1 = INV ? -0.95105651629515357 : 0.95105651629515357;     // sin(2pi/5)
  constexpr double s2 = INV ? -0.58778525229247313 : 0.58778525229247313;     // sin(4pi/5)
  constexpr float c11 = (float)(c1 * r1), c22 = (float)(c2 * r2), c21 = (float)(c2 * r1), c12 = (float)(c1 * r2);
  constexpr float s11 = (float)(s1 * r1), s22 = (float)(s2 * r2), s21 = (float)(s2 * r1), s12 = (float)(s1 * r2);
  const float2 a1 = sadd<S1, S4>(v1, v4), b1 = ssub<S1, S4>(v1, v4);          // scale S1
  const float2 a2 = sadd<S2, S3>(v2, v3), b2 = ssub<S2, S3>(v2, v3);          // scale S2
  const float2 m1 = make_float2(v0.x + c11 * a1.x + c22 * a2.x, v0.y + c11 * a1.y + c22 * a2.y);
  const float2 m2 = make_float2(v0.x + c21 * a1.x + c12 * a2.x, v0.y + c21 * a1.y + c12 * a2.y);
  const float2 q1 = make_float2(s11 * b1.y + s22 * b2.y, -(s11 * b1.x + s22 * b2.x));   // -i (s1 b1 + s2 b2)
  const float2 q2 = make_float2(s21 * b1.y - s12 * b2.y, -(s21 * b1.x - s12 * b2.x));   // -i (s2 b1 - s1 b2)
  v0 = sadd<S0, S2>(sadd<S0, S1>(v0, a1), a2);
  v1 = cadd(m1, q1);
  v4 = csub(m1, q1);
  v2 = cadd(m2, q2);
  v3 = csub(m2, q2);
}
template <bool INV, class S0, class S1, class S2, class S3, class S4, class S5, class S6>
__device__ __forceinline__ void bfly7_s(float2& v0, float2& v1, float2& v2, float2& v3, float2& v4, float2& v5, float2& v6) {
  constexpr double r1 = S1::v / S0::v, r2 = S2::v / S0::v, r3 = S3::v / S0::v;
  constexpr double c1 = 0.62348980185873353, c2 = -0.22252093395631440, c3 = -0.90096886790241913;   // cos(2 pi j / 7)
  constexpr double sg = INV ? -1.0 : 1.0;
  constexpr double s1 = sg * 0.78183148246802981, s2 = sg * 0.97492791218182361, s3 = sg * 0.43388373911755812;
  const float2 a1 = sadd<S1, S6>(v1, v6), b1 = ssub<S1, S6>(v1, v6);          // scale S1
  const float2 a2 = sadd<S2, S5>(v2, v5), b2 = ssub<S2, S5>(v2, v5);          // scale S2
  const float2 a3 = sadd<S3, S4>(v3, v4), b3 = ssub<S3, S4>(v3, v4);          // scale S3
  // X_k = m_k - i q_k, X_{7-k} = m_k + i q_k with m_k = x0 + sum_j a_j cos(2 pi jk/7), q_k = sum_j b_j sin(2 pi jk/7)
  constexpr float c1a = (float)(c1 * r1), c2b = (float)(c2 * r2), c3c = (float)(c3 * r3);
  constexpr float c2a = (float)(c2 * r1), c3b = (float)(c3 * r2), c1c = (float)(c1 * r3);
  constexpr float c3a = (float)(c3 * r1), c1b = (float)(c1 * r2), c2c = (float)(c2 * r3);
  constexpr float s1a = (float)(s1 * r1), s2b = (float)(s2 * r2), s3c = (float)(s3 * r3);
  constexpr float s2a = (float)(s2 * r1), s3b = (float)(s3 * r2), s1c = (float)(s1 * r3);
  constexpr float s3a = (float)(s3 * r1), s1b = (float)(s1 * r2), s2c = (float)(s2 * r3);
  const float2 m1 = make_float2(v0.x + c1a * a1.x + c2b * a2.x + c3c * a3.x, v0.y + c1a * a1.y + c2b * a2.y + c3c * a3.y);
  const float2 m2 = make_float2(v0.x + c2a * a1.x + c3b * a2.x + c1c * a3.x, v0.y + c2a * a1.y + c3b * a2.y + c1c * a3.y);
  const float2 m3 = make_float2(v0.x + c3a * a1.x + c1b * a2.x + c2c * a3.x, v0.y + c3a * a1.y + c1b * a2.y + c2c * a3.y);
  const float2 q1 = make_float2(s1a * b1.x + s2b * b2.x + s3c * b3.x, s1a * b1.y + s2b * b2.y + s3c * b3.y);
  const float2 q2 = make_float2(s2a * b1.x - s3b * b2.x - s1c * b3.x, s2a * b1.y - s3b * b2.y - s1c * b3.y);
  const float2 q3 = make_float2(s3a * b1.x - s1b * b2.x + s2c * b3.x, s3a * b1.y - s1b * b2.y + s2c * b3.y);
  v0 = sadd<S0, S3>(sadd<S0, S2>(sadd<S0, S1>(v0, a1), a2), a3);
  v1 = make_float2(m1.x + q1.y, m1.y - q1.x);   v6 = make_float2(m1.x - q1.y, m1.y + q1.x);
  v2 = make_float2(m2.x + q2.y, m2.y - q2.x);   v5 = make_float2(m2.x - q2.y, m2.y + q2.x);
  v3 = make_float2(m3.x + q3.y, m3.y - q3.x);   v4 = make_float2(m3.x - q3.y, m3.y + q3.x);
}
template <class S> struct HalfScale { static constexpr double v = S::v * 0.70710678118654752440; };
// radix 8, decimation in frequency as bfly8; the (1 -+ i) / sqrt 2 of the odd half is two additions with 1 / sqrt 2 pending
template <bool INV, class C0, class C1, class C2, class C3, class C4, class C5, class C6, class C7>
__device__ __forceinline__ void bfly8_s(float2& a0, float2& a1, float2& a2, float2& a3, float2& a4, float2& a5, float2& a6, float2& a7) {
  float2 s0 = sadd<C0, C4>(a0, a4), d0 = ssub<C0, C4>(a0, a4);          // scale C0
  float2 s1 = sadd<C1, C5>(a1, a5), e1 = ssub<C1, C5>(a1, a5);          // scale C1
  float2 s2 = sadd<C2, C6>(a2, a6), e2 = ssub<C2, C6>(a2, a6);          // scale C2
  float2 s3 = sadd<C3, C7>(a3, a7), e3 = ssub<C3, C7>(a3, a7);          // scale C3
  float2 d1 = INV ? make_float2(e1.x - e1.y, e1.x + e1.y) : make_float2(e1.x + e1.y, e1.y - e1.x);            // * (1 -+ i)
  float2 d2 = twid64<16, INV>(e2);                                                                            // * (-+ i)
  float2 d3 = INV ? make_float2(-(e3.x + e3.y), e3.x - e3.y) : make_float2(e3.y - e3.x, -(e3.x + e3.y));      // * (-1 -+ i)
  bfly4_s<INV, C0, C1, C2, C3>(s0, s1, s2, s3);
  bfly4_s<INV, C0, HalfScale<C1>, C2, HalfScale<C3>>(d0, d1, d2, d3);
  a0 = s0; a2 = s1; a4 = s2; a6 = s3;
  a1 = d0; a3 = d1; a5 = d2; a7 = d3;
}

// ---- the transform ----------------------------------------------------------------------------------------------------
template <int R, bool INV, class Map, class Scale, int NTOT>
__device__ __forceinline__ void fft_cts(float2 (&z)[NTOT]) {
  constexpr int RA = Split<R>::RA, RB = Split<R>::RB;
  if constexpr (RB == 1) {
    using S0 = ScaleAt<Scale, 0>; using S1 = ScaleAt<Scale, 1>; using S2 = ScaleAt<Scale, (R > 2 ? 2 : 0)>; using S3 = ScaleAt<Scale, (R > 3 ? 3 : 0)>;
    using S4 = ScaleAt<Scale, (R > 4 ? 4 : 0)>; using S5 = ScaleAt<Scale, (R > 5 ? 5 : 0)>; using S6 = ScaleAt<Scale, (R > 6 ? 6 : 0)>; using S7 = ScaleAt<Scale, (R > 7 ? 7 : 0)>;
    if constexpr (R == 2) bfly2_s<INV, S0, S1>(z[Map::at(0)], z[Map::at(1)]);
    else if constexpr (R == 3) bfly3_s<INV, S0, S1, S2>(z[Map::at(0)], z[Map::at(1)], z[Map::at(2)]);
    else if constexpr (R == 4) bfly4_s<INV, S0, S1, S2, S3>(z[Map::at(0)], z[Map::at(1)], z[Map::at(2)], z[Map::at(3)]);
    else if constexpr (R == 5) bfly5_s<INV, S0, S1, S2, S3, S4>(z[Map::at(0)], z[Map::at(1)], z[Map::at(2)], z[Map::at(3)], z[Map::at(4)]);
    else if constexpr (R == 7) bfly7_s<INV, S0, S1, S2, S3, S4, S5, S6>(z[Map::at(0)], z[Map::at(1)], z[Map::at(2)], z[Map::at(3)], z[Map::at(4)], z[Map::at(5)], z[Map::at(6)]);
    else bfly8_s<INV, S0, S1, S2, S3, S4, S5, S6, S7>(z[Map::at(0)], z[Map::at(1)], z[Map::at(2)], z[Map::at(3)], z[Map::at(4)], z[Map::at(5)], z[Map::at(6)], z[Map::at(7)]);
  } else {
    // input q = RB*q1 + q0.  Stage 1: radix RA over q1 for every q0 (bin ka replaces q1 = ka), rotated by W_R^(q0 ka) (cosine pending)
    static_for<0, RB>([&](auto q0c) {
      constexpr int q0 = decltype(q0c)::value;
      fft_cts<RA, INV, SubMap<Map, q0, RB>, SubScale<Scale, q0, RB>, NTOT>(z);
      static_for<1, RA>([&](auto kac) {
        constexpr int ka = decltype(kac)::value;
        constexpr int pos = Map::at(RB * ka + q0);
        z[pos] = tw_unscaled_gen<R, q0 * ka, INV>(z[pos]);
      });
    });
    // Stage 2: length RB over q0 for every ka; bin k = ka + RA*kb ends at logical RB*ka + out_pos<RB>(kb)
    static_for<0, RA>([&](auto kac) {
      constexpr int ka = decltype(kac)::value;
      fft_cts<RB, INV, SubMap<Map, RB * ka, 1>, Stage2Scale<Scale, R, ka, INV>, NTOT>(z);
    });
  }
}

// the exact transform (no pending scales on the inputs, none on the outputs)
template <int R, bool INV, class Map, int NTOT>
__device__ __forceinline__ void fft_ct(float2 (&z)[NTOT]) { fft_cts<R, INV, Map, UnitScale, NTOT>(z); }

}  // namespace sfft
