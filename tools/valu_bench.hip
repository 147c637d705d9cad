// valu_bench — wave64 fp32 VALU issue rates on gfx950: plain vs packed (v_pk_*_f32) ops, 1..4 waves per SIMD.
// Decides whether the FFT butterflies should be written on float2 packed ops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s) {
  f2 r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = f2{s * i + threadIdx.x, s - i};
  f2 c = f2{s, s * 0.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (OP == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[i].x) : "v"(r[i].x), "v"(c.x));
      if (OP == 1) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[i].x) : "v"(r[i].x), "v"(c.x), "v"(c.y));
      if (OP == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(c));
      if (OP == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(c), "v"(c));
      if (OP == 4) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(c));
      if (OP == 5) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r[i].x) : "v"(r[i].x), "v"(c.x));
      if (OP == 6) asm volatile("v_sub_f32 %0, %1, %2\n\tv_add_f32 %3, %4, %5" : "=v"(r[i].x), "=v"(r[i].y) : "v"(r[i].x), "v"(c.x), "v"(r[i].y), "v"(c.y));
      if (OP == 7) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r[i]) : "v"(r[i]), "v"(c));
    }
  }
  float acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += r[i].x + r[i].y;
  if (acc == 1234.5f) out[threadIdx.x] = acc;
}

template <int OP> void run(const char* name, int ops_per_stmt, int lanes_per_op) {
  float* out; CK(hipMalloc(&out, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;
  for (int wps : {1, 2, 4, 8}) {          // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
    const int grid = 256 * wps;
    k<OP><<<grid, 256>>>(out, 10, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<OP><<<grid, 256>>>(out, iters, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)iters * 16 * ops_per_stmt * wps;   // wave-instructions issued on one SIMD
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-34s waves/SIMD=%d : %7.3f ms  %.2f cycles@2.4GHz per wave-instr  %.1f Tlane-op/s\n", name, wps, ms,
           cyc / instr_per_simd, instr_per_simd * 1024 * 64.0 * lanes_per_op / (ms * 1e-3) / 1e12);
  }
}
int main() {
  run<0>("v_add_f32", 1, 1); run<5>("v_mul_f32", 1, 1); run<1>("v_fma_f32", 1, 1);
  run<2>("v_pk_add_f32", 1, 2); run<4>("v_pk_mul_f32", 1, 2); run<3>("v_pk_fma_f32", 1, 2);
  run<7>("v_pk_add_f32 neg (a-b)", 1, 2); run<6>("v_sub_f32+v_add_f32 pair", 2, 1);
  return 0;
}
