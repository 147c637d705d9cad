// io_width_bench — does the width of the per-lane access (8 B vs 16 B) change how fast one CU moves a
// 64-byte-segment tile?  Non-persistent grid (one 256 KiB tile per workgroup, XCD-contiguous), with and
// without a compute delay, at full and partial chip occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ int xcd_tile(int t, int n) { const int q = n / 8, rem = n % 8, x = t % 8, i = t / 8; return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + i; }

// VG: 0 = natural VGPR count (132 -> 3 waves/SIMD), 1 = clobber v200 (-> 2 waves/SIMD, like the FFT kernel)
// BAR: __syncthreads() between delay iterations (the FFT kernel's waves are barrier-locked)
template <int W, int VG = 0, int BAR = 0>
__global__ void __launch_bounds__(512) tile_copy(const float* __restrict__ in, float* __restrict__ out, int N, int D, int tpr, int n_tiles,
                                                 int delay, float fa, float fb) {
  extern __shared__ char smem[];
  if (VG) asm volatile("v_mov_b32 v200, 0" ::: "v200");
  if (delay < 0) smem[threadIdx.x] = 0;
  constexpr int LPR = 64 / W, EPT = 4096 * LPR / 512;       // lanes per row, rows per thread
  const int t = xcd_tile(blockIdx.x, n_tiles);
  const int b = t / tpr, ct = t % tpr;
  const int p = threadIdx.x % LPR, r = threadIdx.x / LPR, RC = 512 / LPR;
  const char* si = reinterpret_cast<const char*>(in + (size_t)b * N * D + ct * 16);
  char* so = reinterpret_cast<char*>(out + (size_t)b * N * D + ct * 16);
  const uint32_t voff = (uint32_t)(r * D * 4 + p * W);
  float v[EPT][W / 4];
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const char* ptr = si + (size_t)(q * RC) * D * 4 + voff;
    if (W == 8) { float2 x = *reinterpret_cast<const float2*>(ptr); v[q][0] = x.x; v[q][1] = x.y; }
    else { float4 x = *reinterpret_cast<const float4*>(ptr); v[q][0] = x.x; v[q][1] = x.y; v[q][W / 4 - 2] = x.z; v[q][W / 4 - 1] = x.w; }
  }
  for (int it = 0; it < delay; ++it) {
    if (BAR && (it % 10) == 0) __syncthreads();
#pragma unroll
    for (int q = 0; q < EPT; ++q)
#pragma unroll
      for (int k = 0; k < W / 4; ++k) v[q][k] = fmaf(v[q][k], fa, fb);
  }
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    char* ptr = so + (size_t)(q * RC) * D * 4 + voff;
    if (W == 8) *reinterpret_cast<float2*>(ptr) = make_float2(v[q][0], v[q][1]);
    else *reinterpret_cast<float4*>(ptr) = make_float4(v[q][0], v[q][1], v[q][W / 4 - 2], v[q][W / 4 - 1]);
  }
}

template <int W, int VG = 0, int BAR = 0> void run(const float* in, float* out, int B, int N, int D, int grid_tiles, int delay, int lds = 0) {
  CK(hipFuncSetAttribute((const void*)tile_copy<W, VG, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int tpr = D / 16, n_tiles = B * tpr;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  tile_copy<W, VG, BAR><<<grid_tiles, 512, lds>>>(in, out, N, D, tpr, n_tiles, delay, 1.f, 0.f); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) tile_copy<W, VG, BAR><<<grid_tiles, 512, lds>>>(in, out, N, D, tpr, n_tiles, delay, 1.f, 0.f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  printf("W=%2d VG=%d BAR=%d lds=%3dKB tiles=%5d delay=%3d : %7.3f ms  %7.1f GB/s  %6.2f us per tile per CU\n", W, VG, BAR, lds / 1024, grid_tiles, delay, ms,
         2.0 * grid_tiles * N * 64 / ms / 1e6, ms * 1e3 / ((grid_tiles + 255) / 256));
}
int main() {
  const int B = 256, N = 4096, D = 768; const size_t n = (size_t)B * N * D;
  float *in, *out; CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMemset(in, 0x3c, n * 4));
  for (int delay : {0, 60, 100}) {
    run<8, 0, 0>(in, out, B, N, D, 12288, delay);             // 3 slots/SIMD, no LDS
    run<8, 1, 0>(in, out, B, N, D, 12288, delay);             // 2 slots/SIMD
    run<8, 0, 1>(in, out, B, N, D, 12288, delay);             // 3 slots, barriers
    run<8, 0, 1>(in, out, B, N, D, 12288, delay, 70 * 1024);  // 3 slots, barriers, LDS allows 2 WGs
    run<8, 0, 1>(in, out, B, N, D, 12288, delay, 133 * 1024); // 3 slots, barriers, LDS allows 1 WG
    run<8, 1, 1>(in, out, B, N, D, 12288, delay, 133 * 1024); // the FFT kernel's situation
  }
  return 0;
}
