// copy_ceiling — which PURE COPY of 3.2 GB -> 3.2 GB is the fastest one on this box?  MI355X_MICROARCH.md quotes
// 6.29 TB/s for "a float4 copy"; spectre_probe_copy's persistent dense copy reaches 5.15-5.8.  Every form here moves the same bytes:
//   grid   grid-stride float4 copy, W waves per CU resident, U independent 16-byte loads per lane per iteration, then U stores
//   flat   non-persistent: one workgroup per 256 x U float4 chunk (the classic "one element per thread" copy)
//   bulk   a workgroup loads a whole 256-KiB chunk into registers, THEN stores it (reads and writes of a CU never interleave)
//   xcd    grid-stride where every XCD owns one contiguous eighth of the buffer
//   nt     the same with non-temporal loads and / or stores
//   memcpy hipMemcpyAsync device-to-device
// Sizes: the headline tensor (256 x 4096 x 768 fp32 = 3 GiB) and smaller ones (the 256-MiB Infinity Cache is visible below 1 GiB).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/copy_ceiling.hip -o tools/copy_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ f4 ld(const f4* p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT> __device__ __forceinline__ void st(f4* p, f4 v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// grid-stride: iteration i of the whole grid covers U consecutive "grid rows" of gridDim * blockDim float4 each
template <int U, bool NTL, bool NTS, int THREADS>
__global__ void __launch_bounds__(THREADS) copy_grid(const f4* __restrict__ src, f4* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * THREADS;
  size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int q = 0; q < U; ++q) v[q] = ld<NTL>(src + i + q * stride);
#pragma unroll
    for (int q = 0; q < U; ++q) st<NTS>(dst + i + q * stride, v[q]);
  }
  for (; i < n4; i += stride) st<NTS>(dst + i, ld<NTL>(src + i));
}

// a workgroup owns contiguous chunks of THREADS * U float4 (U x 16 bytes per lane, lanes contiguous), chunk c -> workgroup c % grid
template <int U, bool NTL, bool NTS, int THREADS>
__global__ void __launch_bounds__(THREADS) copy_chunk(const f4* __restrict__ src, f4* __restrict__ dst, size_t n4) {
  const size_t chunk = (size_t)THREADS * U, nch = n4 / chunk;
  for (size_t c = blockIdx.x; c < nch; c += gridDim.x) {
    const size_t b = c * chunk + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int q = 0; q < U; ++q) v[q] = ld<NTL>(src + b + q * THREADS);
#pragma unroll
    for (int q = 0; q < U; ++q) st<NTS>(dst + b + q * THREADS, v[q]);
  }
}

// every XCD (workgroup b runs on XCD b % 8) owns one contiguous eighth of the buffer and streams it with its own workgroups
template <int U, int THREADS>
__global__ void __launch_bounds__(THREADS) copy_xcd(const f4* __restrict__ src, f4* __restrict__ dst, size_t n4) {
  const int xcd = blockIdx.x % 8, wg = blockIdx.x / 8, nwg = gridDim.x / 8;
  const size_t part = n4 / 8, chunk = (size_t)THREADS * U, nch = part / chunk;
  const f4* s = src + xcd * part; f4* d = dst + xcd * part;
  for (size_t c = wg; c < nch; c += nwg) {
    const size_t b = c * chunk + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int q = 0; q < U; ++q) v[q] = s[b + q * THREADS];
#pragma unroll
    for (int q = 0; q < U; ++q) d[b + q * THREADS] = v[q];
  }
}

__global__ void __launch_bounds__(512) store_k(f4* dst, size_t n4) {
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
__global__ void __launch_bounds__(512) load_k(const f4* src, size_t n4, float* sink) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) *sink = acc.x;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 12;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs\n", prop.name, cus);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* sink; CK(hipMalloc(&sink, 4));
  const size_t full = (size_t)256 * 4096 * 768 * 4;
  f4 *a, *b; CK(hipMalloc(&a, full)); CK(hipMalloc(&b, full));
  hipLaunchKernelGGL(store_k, dim3(2048), dim3(512), 0, 0, a, full / 16);
  hipLaunchKernelGGL(store_k, dim3(2048), dim3(512), 0, 0, b, full / 16);
  auto time = [&](auto f) {
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
  // power ramp: 40 ms of back-to-back copies before anything is timed
  for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((copy_grid<4, false, false, 256>), dim3(cus * 8), dim3(256), 0, 0, a, b, full / 16);
  CK(hipDeviceSynchronize());
  struct Row { std::string name; float ms; size_t bytes; };
  std::vector<Row> rows;
  auto report = [&](const char* name, size_t bytes, float ms) {
    printf("  %-58s %8.4f ms  %7.1f GB/s (r+w)\n", name, ms, 2.0 * bytes / ms / 1e6); fflush(stdout);
    rows.push_back({name, ms, bytes});
  };
  for (size_t bytes : {full, (size_t)1 << 30, (size_t)256 << 20}) {
    const size_t n4 = bytes / 16;
    printf("== %zu MiB -> %zu MiB\n", bytes >> 20, bytes >> 20);
    char nm[128];
#define GRID(U, NTL, NTS, T, WPC) do { const int g = cus * (WPC) * 64 / (T); \
      snprintf(nm, sizeof nm, "grid-stride  U=%d T=%d waves/CU=%d%s%s", U, T, WPC, NTL ? " nt-load" : "", NTS ? " nt-store" : ""); \
      report(nm, bytes, time([&] { hipLaunchKernelGGL((copy_grid<U, NTL, NTS, T>), dim3(g), dim3(T), 0, 0, a, b, n4); })); } while (0)
    GRID(1, false, false, 256, 8); GRID(1, false, false, 256, 16); GRID(1, false, false, 256, 32);
    GRID(2, false, false, 256, 16); GRID(4, false, false, 256, 8); GRID(4, false, false, 256, 16); GRID(4, false, false, 256, 32);
    GRID(8, false, false, 256, 8); GRID(8, false, false, 256, 16); GRID(16, false, false, 256, 8);
    GRID(4, false, false, 512, 16); GRID(4, false, false, 1024, 16); GRID(4, false, false, 64, 16);
    GRID(4, true, false, 256, 16); GRID(4, false, true, 256, 16); GRID(4, true, true, 256, 16); GRID(1, true, true, 256, 32);
#define FLAT(U, NTL, NTS, T) do { const size_t g = n4 / ((size_t)(T) * (U)); \
      snprintf(nm, sizeof nm, "flat (1 WG per chunk) U=%d T=%d%s%s", U, T, NTL ? " nt-load" : "", NTS ? " nt-store" : ""); \
      report(nm, bytes, time([&] { hipLaunchKernelGGL((copy_chunk<U, NTL, NTS, T>), dim3((unsigned)g), dim3(T), 0, 0, a, b, n4); })); } while (0)
    FLAT(1, false, false, 256); FLAT(2, false, false, 256); FLAT(4, false, false, 256); FLAT(8, false, false, 256); FLAT(1, false, false, 1024);
    FLAT(4, false, false, 1024); FLAT(1, true, true, 256); FLAT(4, true, true, 256);
#define CHUNK(U, NTL, NTS, T, WGPC) do { \
      snprintf(nm, sizeof nm, "persistent chunks U=%d T=%d WG/CU=%d%s%s", U, T, WGPC, NTL ? " nt-load" : "", NTS ? " nt-store" : ""); \
      report(nm, bytes, time([&] { hipLaunchKernelGGL((copy_chunk<U, NTL, NTS, T>), dim3(cus * (WGPC)), dim3(T), 0, 0, a, b, n4); })); } while (0)
    CHUNK(16, false, false, 512, 1); CHUNK(16, false, false, 512, 2); CHUNK(16, false, false, 512, 4);
    CHUNK(32, false, false, 512, 1); CHUNK(32, false, false, 512, 2);          // "bulk": 256 KiB per workgroup in registers, then stores
    CHUNK(32, false, false, 256, 2); CHUNK(32, false, false, 256, 4);
    CHUNK(8, false, false, 256, 8); CHUNK(4, false, false, 256, 8); CHUNK(16, true, true, 512, 2);
#define XCD(U, T, WGPC) do { snprintf(nm, sizeof nm, "xcd-contiguous eighths U=%d T=%d WG/CU=%d", U, T, WGPC); \
      report(nm, bytes, time([&] { hipLaunchKernelGGL((copy_xcd<U, T>), dim3(cus * (WGPC)), dim3(T), 0, 0, a, b, n4); })); } while (0)
    XCD(4, 256, 4); XCD(4, 256, 8); XCD(16, 512, 2);
    report("hipMemcpyAsync device-to-device", bytes, time([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }));
    report("hipMemcpyDtoDAsync", bytes, time([&] { CK(hipMemcpyDtoDAsync((hipDeviceptr_t)b, (hipDeviceptr_t)a, bytes, 0)); }));
    // the two directions separately, and back to back as two launches (what a chip-wide read phase + write phase would cost)
    const float ldm = time([&] { hipLaunchKernelGGL(load_k, dim3(2048), dim3(512), 0, 0, a, n4, sink); });
    const float stm = time([&] { hipLaunchKernelGGL(store_k, dim3(2048), dim3(512), 0, 0, b, n4); });
    printf("  load only %.4f ms (%.1f GB/s)   store only %.4f ms (%.1f GB/s)   sum %.4f ms = %.1f GB/s (r+w)\n", ldm, bytes / ldm / 1e6, stm, bytes / stm / 1e6,
           ldm + stm, 2.0 * bytes / (ldm + stm) / 1e6);
    // reverse direction of the best grid form (placement classes: DESIGN.md section 5 item 7)
    report("grid-stride U=4 T=256 waves/CU=16, b -> a", bytes, time([&] { hipLaunchKernelGGL((copy_grid<4, false, false, 256>), dim3(cus * 4), dim3(256), 0, 0, b, a, n4); }));
    auto best = std::min_element(rows.begin(), rows.end(), [&](const Row& x, const Row& y) { return (x.bytes == bytes ? x.ms : 1e9f) < (y.bytes == bytes ? y.ms : 1e9f); });
    printf("  BEST at %zu MiB: %s  %.4f ms  %.1f GB/s\n", bytes >> 20, best->name.c_str(), best->ms, 2.0 * bytes / best->ms / 1e6);
  }
  return 0;
}
