// iolab2 — which flavour of global load / store moves the 4096-row x 64-B tile fastest when every CU runs the FFT kernel's
// rhythm (load tile, ~18 us of barrier-locked compute, store tile; one 512-thread workgroup per CU)?
// Loads: plain / nt / sc1.  Stores: plain / nt / sc1 / sc0 sc1.  8 or 16 bytes per lane.  Inline asm so the cache-policy bits are
// exactly what is written here.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/iolab2.hip -o tools/iolab2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int xcd_tile(int t, int n) { const int q = n / 8, rem = n % 8, x = t % 8, i = t / 8; return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + i; }

template <int LF> __device__ __forceinline__ void ld16(f32x4& v, const char* p) {
  if (LF == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p));
  if (LF == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(v) : "v"(p));
  if (LF == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v) : "v"(p));
}
template <int LF> __device__ __forceinline__ void ld8(f32x2& v, const char* p) {
  if (LF == 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(v) : "v"(p));
  if (LF == 1) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=&v"(v) : "v"(p));
  if (LF == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(v) : "v"(p));
}
template <int SF> __device__ __forceinline__ void st16(char* p, f32x4 v) {
  if (SF == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if (SF == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
  if (SF == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if (SF == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
template <int SF> __device__ __forceinline__ void st8(char* p, f32x2 v) {
  if (SF == 0) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if (SF == 1) asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
  if (SF == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if (SF == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

template <int W, int LF, int SF>
__global__ void __launch_bounds__(512) tile_io(const float* __restrict__ in, float* __restrict__ out, int N, int D, int tpr, int n_tiles, int delay, float fa, float fb) {
  extern __shared__ char smem[];
  asm volatile("v_mov_b32 v200, 0" ::: "v200");
  if (delay < 0) smem[threadIdx.x] = 0;
  constexpr int LPR = 64 / W, EPT = 4096 * LPR / 512;
  const int t = xcd_tile(blockIdx.x, n_tiles);
  const int b = t / tpr, ct = t % tpr;
  const int p = threadIdx.x % LPR, r = threadIdx.x / LPR, RC = 512 / LPR;
  const char* si = reinterpret_cast<const char*>(in + (size_t)b * N * D + ct * 16) + (size_t)r * D * 4 + p * W;
  char* so = reinterpret_cast<char*>(out + (size_t)b * N * D + ct * 16) + (size_t)r * D * 4 + p * W;
  if constexpr (W == 16) {
    f32x4 v[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) ld16<LF>(v[q], si + (size_t)(q * RC) * D * 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < EPT; ++q) asm volatile("" : "+v"(v[q]));
    for (int it = 0; it < delay; ++it) {
      if ((it % 10) == 0) __syncthreads();
#pragma unroll
      for (int q = 0; q < EPT; ++q) v[q] = v[q] * fa + fb;
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) st16<SF>(so + (size_t)(q * RC) * D * 4, v[q]);
  } else {
    f32x2 v[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) ld8<LF>(v[q], si + (size_t)(q * RC) * D * 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < EPT; ++q) asm volatile("" : "+v"(v[q]));
    for (int it = 0; it < delay; ++it) {
      if ((it % 10) == 0) __syncthreads();
#pragma unroll
      for (int q = 0; q < EPT; ++q) v[q] = v[q] * fa + fb;
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) st8<SF>(so + (size_t)(q * RC) * D * 4, v[q]);
  }
}
static hipEvent_t e0, e1;
template <int W, int LF, int SF> void run(const float* in, float* out, int delay) {
  const int B = 256, N = 4096, D = 768, tpr = D / 16, n_tiles = B * tpr, grid = n_tiles;
  CK(hipFuncSetAttribute((const void*)tile_io<W, LF, SF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto f = [&] { tile_io<W, LF, SF><<<grid, 512, 133 * 1024>>>(in, out, N, D, tpr, n_tiles, delay, 1.f, 0.f); };
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const char* lf[] = {"plain", "nt", "sc1"}; const char* sf[] = {"plain", "nt", "sc1", "sc0sc1"};
  printf("W=%2d load=%-5s store=%-6s delay=%3d : %7.3f ms  %7.1f GB/s  %6.2f us/tile/CU\n", W, lf[LF], sf[SF], delay, ms, 2.0 * grid * N * 64 / ms / 1e6, ms * 1e3 / 48);
  fflush(stdout);
}
template <int W> void sweep(const float* in, float* out, int delay) {
  run<W, 0, 0>(in, out, delay); run<W, 0, 1>(in, out, delay); run<W, 0, 2>(in, out, delay); run<W, 0, 3>(in, out, delay);
  run<W, 1, 0>(in, out, delay); run<W, 1, 1>(in, out, delay); run<W, 1, 2>(in, out, delay);
  run<W, 2, 0>(in, out, delay); run<W, 2, 2>(in, out, delay);
}
int main() {
  const int B = 256, N = 4096, D = 768; const size_t n = (size_t)B * N * D;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float *in, *out; CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMemset(in, 0x3c, n * 4)); CK(hipMemset(out, 0, n * 4));
  for (int delay : {86, 0}) { sweep<16>(in, out, delay); sweep<8>(in, out, delay); }
  return 0;
}
