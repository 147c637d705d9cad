// Runs the compile-time in-register FFTs of fft_amd/csrc/fft_regs_mixed.h on the host (g++), one length per call.
#ifdef SFFT_ENGINE_HEADER
#include SFFT_ENGINE_HEADER          // tools/fft_regs_mixed_scaled.h: the round-3 scaled-twiddle experiment, same interface
#else
#include "fft_regs_mixed.h"
#endif
using namespace sfft;

template <int R, bool INV>
static void run(float* d) {
  float2 z[R];
  for (int i = 0; i < R; ++i) z[i] = make_float2(d[2 * i], d[2 * i + 1]);
  fft_ct<R, INV, IdentityMap, R>(z);
  static_for<0, R>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    d[2 * k] = z[out_pos<R>(k)].x;
    d[2 * k + 1] = z[out_pos<R>(k)].y;
  });
}
// forward transform, then inverse consuming the bins where they lie (OutPosMap): returns R * input in natural order
template <int R>
static void round_trip(float* d) {
  float2 z[R];
  for (int i = 0; i < R; ++i) z[i] = make_float2(d[2 * i], d[2 * i + 1]);
  fft_ct<R, false, IdentityMap, R>(z);
  fft_ct<R, true, OutPosMap<R>, R>(z);
  static_for<0, R>([&](auto nc) {
    constexpr int n = decltype(nc)::value;
    constexpr int pos = OutPosMap<R>::at(out_pos<R>(n));
    d[2 * n] = z[pos].x;
    d[2 * n + 1] = z[pos].y;
  });
}

#define CASE(R_) case R_: if (mode == 2) round_trip<R_>(d); else if (mode == 1) run<R_, true>(d); else run<R_, false>(d); return 0;
extern "C" int fft_engine_run(int R, int mode, float* d) {
  switch (R) {
    CASE(2) CASE(3) CASE(4) CASE(5) CASE(7) CASE(8) CASE(6) CASE(14) CASE(10) CASE(12) CASE(15) CASE(16) CASE(20) CASE(24) CASE(25)
    CASE(30) CASE(32) CASE(40) CASE(48) CASE(50) CASE(56) CASE(60) CASE(64)
    default: return -1;
  }
}

// The power-of-two two-factor transforms of fft_regs.h (type A: natural order in -> bin k = ka + RA kb at position RB ka + kb;
// type B: bins at those positions in -> natural order out), twiddles in the scaled form.  mode 0: forward A, bins returned in natural
// order; 1: inverse A; 2: forward A then inverse B (returns X * input, no reordering in between); 3: inverse through B alone
template <int X, bool INV>
static void run_a(float* d) {
  constexpr int RA = FftCfg<X>::RA, RB = FftCfg<X>::RB;
  float2 z[X];
  for (int i = 0; i < X; ++i) z[i] = make_float2(d[2 * i], d[2 * i + 1]);
  fftA<RA, RB, INV>(z);
  for (int k = 0; k < X; ++k) { const int ka = k % RA, kb = k / RA; d[2 * k] = z[RB * ka + kb].x; d[2 * k + 1] = z[RB * ka + kb].y; }
}
template <int X>
static void run_ab(float* d, bool only_b) {
  constexpr int RA = FftCfg<X>::RA, RB = FftCfg<X>::RB;
  float2 z[X];
  if (only_b) { for (int k = 0; k < X; ++k) { const int ka = k % RA, kb = k / RA; z[RB * ka + kb] = make_float2(d[2 * k], d[2 * k + 1]); } }
  else { for (int i = 0; i < X; ++i) z[i] = make_float2(d[2 * i], d[2 * i + 1]); fftA<RA, RB, false>(z); }
  static_for<0, RA>([&](auto kac) { fftB_stage1_group<RA, RB, true, decltype(kac)::value>(z); });
  fftB_stage2<RA, RB, true>(z);
  for (int i = 0; i < X; ++i) { d[2 * i] = z[i].x; d[2 * i + 1] = z[i].y; }
}
#define CASE2(X_) case X_: if (mode == 0) run_a<X_, false>(d); else if (mode == 1) run_a<X_, true>(d); else run_ab<X_>(d, mode == 3); return 0;
extern "C" int fft_regs_run(int X, int mode, float* d) {
  switch (X) { CASE2(16) CASE2(32) CASE2(64) default: return -1; }
}
