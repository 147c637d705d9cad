// pk_rate_bench — issue rate of packed fp32 VALU instructions on gfx950: v_fma_f32 / v_add_f32 against v_pk_fma_f32 / v_pk_add_f32 /
// v_pk_mul_f32 (two fp32 results per lane per instruction), 1, 2 and 4 waves per SIMD, independent accumulators.  Prints cycles per
// instruction per SIMD (s_memtime x clock ratio is avoided: wall time x an assumed clock is printed next to instructions / ns).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pk_rate_bench.hip -o tools/pk_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kAcc = 16, kIter = 4096;

template <int KIND>
__global__ void __launch_bounds__(1024) rate(float* out, float k) {
  f2 a[kAcc];
  for (int i = 0; i < kAcc; ++i) a[i] = f2{(float)threadIdx.x * 1e-3f + i, (float)i};
  f2 b = f2{k, -k}, c = f2{1e-6f, 2e-6f};
  for (int it = 0; it < kIter; ++it) {
#pragma unroll
    for (int i = 0; i < kAcc; ++i) {
      if constexpr (KIND == 0) {         // 2 x v_fma_f32
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(b.y), "v"(c.y));
      } else if constexpr (KIND == 1) {  // 1 x v_pk_fma_f32
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      } else if constexpr (KIND == 2) {  // 2 x v_add_f32
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(b.x));
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].y) : "v"(b.y));
      } else if constexpr (KIND == 3) {  // 1 x v_pk_add_f32
        asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      } else if constexpr (KIND == 4) {  // 1 x v_pk_mul_f32
        asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
      } else if constexpr (KIND == 5) {  // v_pk_add_f32 with swapped halves and a negated half: a + (-i) b style
        asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "+v"(a[i]) : "v"(b));
      } else {                           // v_pk_fma_f32 with an SGPR-pair constant
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "s"(c));
      }
    }
  }
  f2 s = a[0];
  for (int i = 1; i < kAcc; ++i) s += a[i];
  if (s.x == 123.456f) out[threadIdx.x] = s.x + s.y;
}

int main() {
  float* out; CK(hipMalloc(&out, 4096));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"2 x v_fma_f32", "1 x v_pk_fma_f32", "2 x v_add_f32", "1 x v_pk_add_f32", "1 x v_pk_mul_f32", "1 x v_pk_add_f32 op_sel/neg", "1 x v_pk_fma_f32 (SGPR pair)"};
  auto run = [&](int kind, int threads) {
    auto launch = [&] {
      switch (kind) {
        case 0: hipLaunchKernelGGL(rate<0>, dim3(cus), dim3(threads), 0, 0, out, 0.5f); break;
        case 1: hipLaunchKernelGGL(rate<1>, dim3(cus), dim3(threads), 0, 0, out, 0.5f); break;
        case 2: hipLaunchKernelGGL(rate<2>, dim3(cus), dim3(threads), 0, 0, out, 0.5f); break;
        case 3: hipLaunchKernelGGL(rate<3>, dim3(cus), dim3(threads), 0, 0, out, 0.5f); break;
        case 4: hipLaunchKernelGGL(rate<4>, dim3(cus), dim3(threads), 0, 0, out, 0.5f); break;
        case 5: hipLaunchKernelGGL(rate<5>, dim3(cus), dim3(threads), 0, 0, out, 0.5f); break;
        default: hipLaunchKernelGGL(rate<6>, dim3(cus), dim3(threads), 0, 0, out, 0.5f); break;
      }
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    const double pairs = (double)kAcc * kIter * (threads / 64) / 4.0;   // complex-sized results per SIMD (waves spread over 4 SIMDs)
    printf("  %-30s %4d threads/CU: %.3f ms  -> %.2f ns per (2-float result) per SIMD = %.2f cycles at 2.4 GHz\n", names[kind], threads, ms, ms * 1e6 / pairs, ms * 1e6 / pairs * 2.4);
  };
  for (int threads : {256, 512, 1024})
    for (int kind = 0; kind < 7; ++kind) run(kind, threads);
  return 0;
}
