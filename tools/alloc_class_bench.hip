// alloc_class_bench — does HOW device memory is allocated decide which of the two placement classes it lands in (DESIGN.md section 5,
// round 3, item 7)?  K buffers of 3 GiB each through hipMalloc, hipMallocAsync (stream-ordered pool), hipExtMallocWithFlags and the virtual
// memory API (hipMemCreate + hipMemMap at the recommended / minimum granularity); a dense store-only and a load-only pass over each.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/alloc_class_bench.hip -o tools/alloc_class_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) store_k(f4* dst, size_t n4) {
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
__global__ void __launch_bounds__(512) load_k(const f4* src, size_t n4, float* sink) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) *sink = acc.x;
}

int main() {
  const size_t bytes = (size_t)3 << 30, n4 = bytes / 16;
  const int K = 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* sink; CK(hipMalloc(&sink, 4));
  auto time = [&](auto f) { for (int i = 0; i < 4; ++i) f(); CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 10; };
  auto probe = [&](const char* how, void* p) {
    const float st = time([&] { hipLaunchKernelGGL(store_k, dim3(2048), dim3(512), 0, 0, (f4*)p, n4); });
    const float ld = time([&] { hipLaunchKernelGGL(load_k, dim3(2048), dim3(512), 0, 0, (const f4*)p, n4, sink); });
    printf("  %-44s %p  store %.4f ms (%.2f TB/s)  load %.4f ms (%.2f TB/s)\n", how, p, st, bytes / st / 1e9, ld, bytes / ld / 1e9);
    fflush(stdout);
  };
  { void* w; CK(hipMalloc(&w, bytes)); for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(store_k, dim3(2048), dim3(512), 0, 0, (f4*)w, n4); CK(hipDeviceSynchronize()); CK(hipFree(w)); }  // power ramp
  printf("hipMalloc:\n");
  std::vector<void*> keep;
  for (int i = 0; i < K; ++i) { void* p; CK(hipMalloc(&p, bytes)); keep.push_back(p); probe("hipMalloc", p); }
  printf("hipMallocAsync (default pool):\n");
  for (int i = 0; i < 4; ++i) { void* p; if (hipMallocAsync(&p, bytes, 0) != hipSuccess) { printf("  not available\n"); break; } CK(hipStreamSynchronize(0)); keep.push_back(p); probe("hipMallocAsync", p); }
  printf("hipExtMallocWithFlags:\n");
  for (unsigned fl : {(unsigned)hipDeviceMallocDefault, (unsigned)hipDeviceMallocUncached, (unsigned)hipDeviceMallocContiguous}) {
    for (int i = 0; i < 2; ++i) { void* p; if (hipExtMallocWithFlags(&p, bytes, fl) != hipSuccess) { printf("  flags %#x: not available\n", fl); (void)hipGetLastError(); break; }
      keep.push_back(p); char nm[64]; snprintf(nm, 64, "hipExtMallocWithFlags(%#x)", fl); probe(nm, p); }
  }
  printf("virtual memory API:\n");
  {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    if (hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum) != hipSuccess ||
        hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { printf("  not available\n"); }
    else {
      printf("  granularity: minimum %zu, recommended %zu\n", gmin, grec);
      for (int mode = 0; mode < 3; ++mode) {            // 0: one 3-GiB handle, 1: 2-MiB (minimum-granularity) handles, 2: 64-MiB handles
        const size_t chunk = mode == 0 ? bytes : mode == 1 ? (gmin > (2u << 20) ? gmin : (2u << 20)) : ((size_t)64 << 20);
        for (int i = 0; i < 3; ++i) {
          hipDeviceptr_t va;
          if (hipMemAddressReserve(&va, bytes, 0, 0, 0) != hipSuccess) { printf("  reserve failed\n"); break; }
          bool ok = true;
          for (size_t off = 0; off < bytes && ok; off += chunk) {
            hipMemGenericAllocationHandle_t h;
            ok = hipMemCreate(&h, chunk, &prop, 0) == hipSuccess && hipMemMap((char*)va + off, chunk, 0, h, 0) == hipSuccess;
          }
          hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
          ok = ok && hipMemSetAccess(va, bytes, &acc, 1) == hipSuccess;
          if (!ok) { printf("  hipMemCreate / hipMemMap failed: %s\n", hipGetErrorString(hipGetLastError())); break; }
          char nm[64]; snprintf(nm, 64, "hipMemCreate, %zu-MiB handles", chunk >> 20); probe(nm, (void*)va);
        }
      }
    }
  }
  return 0;
}
