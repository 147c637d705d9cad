// fft_regs_mixed.h — compile-time mixed-radix FFTs on register arrays (one transform per lane), gfx950.
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
#include "fft_regs.h"
#include "fft_tables_mixed.h"

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
template <> struct Split<48> { static constexpr int RA = 3, RB = 16; };   // coprime (8 x 6 is not): prime-factor form
template <> struct Split<50> { static constexpr int RA = 2, RB = 25; };
template <> struct Split<56> { static constexpr int RA = 8, RB = 7; };
template <> struct Split<60> { static constexpr int RA = 4, RB = 15; };
template <> struct Split<64> { static constexpr int RA = 8, RB = 8; };

// Coprime factors take the prime-factor (Good-Thomas) form: with the input index written as n = (RB n1 + RA n2) mod R,
//   W_R^(n k) = W_RA^(n1 k) W_RB^(n2 k) = W_RA^(n1 (k mod RA)) W_RB^(n2 (k mod RB)),
// a plain RA x RB two-dimensional transform WITHOUT twiddles between the stages — bin k is the pair (k mod RA, k mod RB).  Everything is
// a compile-time register index here, so the two index maps cost nothing; at 60 = 4 x 15 (15 = 3 x 5) this removes all 74 complex
// multiplications of the Cooley-Tukey form (23 % of the transform's instructions), at 50 = 2 x 25 the 24 of the outer stage (round 4).
constexpr int sfft_gcd(int a, int b) { return b == 0 ? a : sfft_gcd(b, a % b); }
template <int R> constexpr bool split_is_coprime() { return Split<R>::RB != 1 && sfft_gcd(Split<R>::RA, Split<R>::RB) == 1; }

// logical slot (relative to the transform's input map) where output bin k is left
template <int R> constexpr int out_pos(int k) {
  if constexpr (Split<R>::RB == 1) return k;
  else if constexpr (split_is_coprime<R>()) return (Split<R>::RB * (k % Split<R>::RA) + Split<R>::RA * out_pos<Split<R>::RB>(k % Split<R>::RB)) % R;
  else return Split<R>::RB * (k % Split<R>::RA) + out_pos<Split<R>::RB>(k / Split<R>::RA);
}
// logical input index that the transform's first stage consumes i-th (kernels request / unpack their rows in this order)
template <int R> constexpr int in_order(int i) {
  if constexpr (Split<R>::RB == 1) return i;
  else if constexpr (split_is_coprime<R>()) return (Split<R>::RB * (i % Split<R>::RA) + Split<R>::RA * (i / Split<R>::RA)) % R;
  else return (i / Split<R>::RA) + Split<R>::RB * (i % Split<R>::RA);
}

// ---- maps -----------------------------------------------------------------------------------------------------------
struct IdentityMap { static constexpr int at(int i) { return i; } };
template <class Parent, int BASE, int STRIDE> struct SubMap { static constexpr int at(int i) { return Parent::at(BASE + STRIDE * i); } };
template <class Parent, int BASE, int STRIDE, int R> struct ModSubMap { static constexpr int at(int i) { return Parent::at((BASE + STRIDE * i) % R); } };
// input of a transform that consumes the output of an R-point transform bin by bin: logical k -> out_pos<R>(k)
template <int R, class Parent = IdentityMap> struct OutPosMap { static constexpr int at(int k) { return Parent::at(out_pos<R>(k)); } };

// ---- primitive butterflies 3 and 5 with direction ---------------------------------------------------------------------
template <bool INV>
__device__ __forceinline__ void bfly3(float2& v0, float2& v1, float2& v2) {
  constexpr float s = INV ? -0.86602540378443865f : 0.86602540378443865f;   // forward W3 = -1/2 - i sqrt(3)/2
  const float2 t = cadd(v1, v2);
  const float2 d = csub(v1, v2);
  const float2 m = make_float2(v0.x - 0.5f * t.x, v0.y - 0.5f * t.y);
  const float2 r = make_float2(s * d.y, -s * d.x);                           // -i s d
  v0 = cadd(v0, t);
  v1 = cadd(m, r);
  v2 = csub(m, r);
}
template <bool INV>
__device__ __forceinline__ void bfly5(float2& v0, float2& v1, float2& v2, float2& v3, float2& v4) {
  constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;     // cos(2pi/5), cos(4pi/5)
  constexpr float s1 = INV ? -0.95105651629515357f : 0.95105651629515357f;   // sin(2pi/5)
  constexpr float s2 = INV ? -0.58778525229247313f : 0.58778525229247313f;   // sin(4pi/5)
  const float2 a1 = cadd(v1, v4), b1 = csub(v1, v4);
  const float2 a2 = cadd(v2, v3), b2 = csub(v2, v3);
  const float2 m1 = make_float2(v0.x + c1 * a1.x + c2 * a2.x, v0.y + c1 * a1.y + c2 * a2.y);
  const float2 m2 = make_float2(v0.x + c2 * a1.x + c1 * a2.x, v0.y + c2 * a1.y + c1 * a2.y);
  const float2 q1 = make_float2(s1 * b1.y + s2 * b2.y, -(s1 * b1.x + s2 * b2.x));   // -i (s1 b1 + s2 b2)
  const float2 q2 = make_float2(s2 * b1.y - s1 * b2.y, -(s2 * b1.x - s1 * b2.x));   // -i (s2 b1 - s1 b2)
  v0 = make_float2(v0.x + a1.x + a2.x, v0.y + a1.y + a2.y);
  v1 = cadd(m1, q1);
  v4 = csub(m1, q1);
  v2 = cadd(m2, q2);
  v3 = csub(m2, q2);
}

template <bool INV>
__device__ __forceinline__ void bfly7(float2& v0, float2& v1, float2& v2, float2& v3, float2& v4, float2& v5, float2& v6) {
  constexpr float c1 = 0.62348980185873353f, c2 = -0.22252093395631440f, c3 = -0.90096886790241913f;   // cos(2 pi j / 7)
  constexpr float sg = INV ? -1.f : 1.f;
  constexpr float s1 = sg * 0.78183148246802981f, s2 = sg * 0.97492791218182361f, s3 = sg * 0.43388373911755812f;
  const float2 a1 = cadd(v1, v6), b1 = csub(v1, v6);
  const float2 a2 = cadd(v2, v5), b2 = csub(v2, v5);
  const float2 a3 = cadd(v3, v4), b3 = csub(v3, v4);
  // X_k = m_k - i q_k, X_{7-k} = m_k + i q_k with m_k = x0 + sum_j a_j cos(2 pi jk/7), q_k = sum_j b_j sin(2 pi jk/7)
  const float2 m1 = make_float2(v0.x + c1 * a1.x + c2 * a2.x + c3 * a3.x, v0.y + c1 * a1.y + c2 * a2.y + c3 * a3.y);
  const float2 m2 = make_float2(v0.x + c2 * a1.x + c3 * a2.x + c1 * a3.x, v0.y + c2 * a1.y + c3 * a2.y + c1 * a3.y);
  const float2 m3 = make_float2(v0.x + c3 * a1.x + c1 * a2.x + c2 * a3.x, v0.y + c3 * a1.y + c1 * a2.y + c2 * a3.y);
  const float2 q1 = make_float2(s1 * b1.x + s2 * b2.x + s3 * b3.x, s1 * b1.y + s2 * b2.y + s3 * b3.y);
  const float2 q2 = make_float2(s2 * b1.x - s3 * b2.x - s1 * b3.x, s2 * b1.y - s3 * b2.y - s1 * b3.y);
  const float2 q3 = make_float2(s3 * b1.x - s1 * b2.x + s2 * b3.x, s3 * b1.y - s1 * b2.y + s2 * b3.y);
  v0 = make_float2(v0.x + a1.x + a2.x + a3.x, v0.y + a1.y + a2.y + a3.y);
  v1 = make_float2(m1.x + q1.y, m1.y - q1.x);   v6 = make_float2(m1.x - q1.y, m1.y + q1.x);
  v2 = make_float2(m2.x + q2.y, m2.y - q2.x);   v5 = make_float2(m2.x - q2.y, m2.y + q2.x);
  v3 = make_float2(m3.x + q3.y, m3.y - q3.x);   v4 = make_float2(m3.x - q3.y, m3.y + q3.x);
}

// a * W_R^M (forward) or a * conj(W_R^M) (INV), literal constants
template <int R, int M, bool INV>
__device__ __forceinline__ float2 twid_ct(float2 a) {
  constexpr int m = INV ? ((R - (M % R)) % R) : (M % R);
  if constexpr (m == 0) {
    return a;
  } else if constexpr (4 * m == R) {          // -i
    return make_float2(a.y, -a.x);
  } else if constexpr (2 * m == R) {          // -1
    return make_float2(-a.x, -a.y);
  } else if constexpr (4 * m == 3 * R) {      // +i
    return make_float2(-a.y, a.x);
  } else {                                    // (x + iy)(c - is)
    constexpr float c = (float)TwTab<R>::c[m];
    constexpr float s = (float)TwTab<R>::s[m];
    return make_float2(a.x * c + a.y * s, a.y * c - a.x * s);
  }
}
// lengths that already have W_64-based literals in fft_regs.h
template <int R, int M, bool INV>
__device__ __forceinline__ float2 twid_any(float2 a) {
  if constexpr (64 % R == 0) return twid64<(64 / R) * M, INV>(a);
  else return twid_ct<R, M, INV>(a);
}

// ---- the transform ----------------------------------------------------------------------------------------------------
template <int R, bool INV, class Map, int NTOT>
__device__ __forceinline__ void fft_ct(float2 (&z)[NTOT]) {
  constexpr int RA = Split<R>::RA, RB = Split<R>::RB;
  if constexpr (RB == 1) {
    if constexpr (R == 2) bfly2<INV>(z[Map::at(0)], z[Map::at(1)]);
    else if constexpr (R == 3) bfly3<INV>(z[Map::at(0)], z[Map::at(1)], z[Map::at(2)]);
    else if constexpr (R == 4) bfly4<INV>(z[Map::at(0)], z[Map::at(1)], z[Map::at(2)], z[Map::at(3)]);
    else if constexpr (R == 5) bfly5<INV>(z[Map::at(0)], z[Map::at(1)], z[Map::at(2)], z[Map::at(3)], z[Map::at(4)]);
    else if constexpr (R == 7) bfly7<INV>(z[Map::at(0)], z[Map::at(1)], z[Map::at(2)], z[Map::at(3)], z[Map::at(4)], z[Map::at(5)], z[Map::at(6)]);
    else bfly8<INV>(z[Map::at(0)], z[Map::at(1)], z[Map::at(2)], z[Map::at(3)], z[Map::at(4)], z[Map::at(5)], z[Map::at(6)], z[Map::at(7)]);
  } else if constexpr (split_is_coprime<R>()) {
    // input n = (RB n1 + RA n2) mod R.  Stage 1: radix RA over n1 for every n2 (bin ka replaces n1 = ka); stage 2: length RB over n2 for
    // every ka; bin k = (k mod RA, k mod RB) ends at logical (RB ka + RA out_pos<RB>(kb)) mod R.  No twiddles.
    static_for<0, RB>([&](auto n2c) { fft_ct<RA, INV, ModSubMap<Map, RA * decltype(n2c)::value, RB, R>, NTOT>(z); });
    static_for<0, RA>([&](auto kac) { fft_ct<RB, INV, ModSubMap<Map, RB * decltype(kac)::value, RA, R>, NTOT>(z); });
  } else {
    // input q = RB*q1 + q0.  Stage 1: radix RA over q1 for every q0 (bin ka replaces q1 = ka), times W_R^(q0 ka)
    static_for<0, RB>([&](auto q0c) {
      constexpr int q0 = decltype(q0c)::value;
      fft_ct<RA, INV, SubMap<Map, q0, RB>, NTOT>(z);
      static_for<1, RA>([&](auto kac) {
        constexpr int ka = decltype(kac)::value;
        constexpr int pos = Map::at(RB * ka + q0);
        z[pos] = twid_any<R, q0 * ka, INV>(z[pos]);
      });
    });
    // Stage 2: length RB over q0 for every ka; bin k = ka + RA*kb ends at logical RB*ka + out_pos<RB>(kb)
    static_for<0, RA>([&](auto kac) { fft_ct<RB, INV, SubMap<Map, RB * decltype(kac)::value, 1>, NTOT>(z); });
  }
}

}  // namespace sfft
