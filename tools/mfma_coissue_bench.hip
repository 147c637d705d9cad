// mfma_coissue_bench — VERDICT r03 item 8: the fp32 MFMA pipe is idle in the spectral mix; can it take butterfly work BESIDE the VALU?
// Three instruction streams per wave, two waves per SIMD (512-thread workgroups, one per CU, like the 4096 kernel), no memory traffic:
//   valu       radix-4 butterflies on 32 complex registers (16 independent v_add/v_sub per butterfly: the FFT's instruction mix)
//   mfma       v_mfma_f32_4x4x1_16B_f32 chains on 4 independent accumulator quads: D[i][lane] += A[i] * B[lane] per 4-lane block —
//              the one MFMA shape that applies a small matrix to LANE-PRIVATE data (a lane's own register is the B operand)
//   both       the two interleaved, R VALU instructions per MFMA
// Reported: time per iteration of each stream alone and together, i.e. how much of the MFMA stream hides behind the VALU stream.
// What a butterfly would cost there: a radix-4 complex butterfly is a real 8 x 8 matrix per lane = 64 FMAs = 16 of these MFMAs
// (4 outputs x 1 input each), against 16 VALU additions.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_coissue_bench.hip -o tools/mfma_coissue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int R>   // MODE 1 valu, 2 mfma, 3 both; R = VALU butterflies (16 instructions each) per group of 4 MFMAs
__global__ void __launch_bounds__(512, 2) k(float* out, int iters, float seed) {
  float re[32], im[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) { re[i] = seed * (threadIdx.x + i); im[i] = seed * (threadIdx.x - i); }
  f4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f4{seed, seed, seed, seed};
  float a = seed * threadIdx.x, b = seed + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if constexpr (MODE & 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int q = (4 * (g * R + r)) & 31;     // butterfly on registers q, q+1, q+2, q+3 (radix-4, no twiddles: 16 additions)
          const float t0r = re[q] + re[q + 2], t0i = im[q] + im[q + 2], t1r = re[q] - re[q + 2], t1i = im[q] - im[q + 2];
          const float t2r = re[q + 1] + re[q + 3], t2i = im[q + 1] + im[q + 3], t3r = re[q + 1] - re[q + 3], t3i = im[q + 1] - im[q + 3];
          re[q] = t0r + t2r; im[q] = t0i + t2i; re[q + 2] = t0r - t2r; im[q + 2] = t0i - t2i;
          re[q + 1] = t1r + t3i; im[q + 1] = t1i - t3r; re[q + 3] = t1r - t3i; im[q + 3] = t1i + t3r;
        }
      }
      if constexpr (MODE & 2) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += re[i] + im[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 1.2345e-30f) out[threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, iters = 2000;
  float* out; CK(hipMalloc(&out, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto kern) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(cus), dim3(512), 0, 0, out, iters, 1e-6f);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(cus), dim3(512), 0, 0, out, iters, 1e-6f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
  };
  printf("%d CUs, 512-thread workgroups (two waves per SIMD), %d iterations x 8 groups; per group: R radix-4 butterflies (16 VALU additions each) and / or 4 v_mfma_f32_4x4x1_16B_f32\n", cus, iters);
  const float m2 = time(k<2, 1>);
  printf("MFMA alone (4 per group)                      %.3f ms  = %.1f ns per MFMA and wave pair\n", m2, m2 * 1e6 / (iters * 8 * 4));
#define ROW(R) { const float v = time(k<1, R>), bth = time(k<3, R>); \
    printf("R = %d: VALU alone %.3f ms, with MFMA %.3f ms -> %.0f %% of the MFMA stream hidden (VALU instr per MFMA: %d)\n", R, v, bth, 100.0 * (1.0 - (bth - v) / m2), 4 * R); }
  ROW(1) ROW(2) ROW(4) ROW(8)
  printf("A radix-4 butterfly as a lane-private real 8 x 8 matrix = 16 such MFMAs; the same butterfly on the VALU = 16 additions.\n");
  return 0;
}
