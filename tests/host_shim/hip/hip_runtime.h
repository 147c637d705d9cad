// Host shim so that the in-register FFT headers (fft_regs.h, fft_regs_mixed.h) compile with plain g++ for the CPU-tier
// test of the butterfly engine (tests/test_fft_engine_cpu.py).  Test infrastructure only.
#pragma once
#include <cstdint>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
