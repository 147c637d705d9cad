// vram_map — where do the "allocation classes" of MI355X device memory come from?  (LABNOTES round 3 item 7: a dense store-only pass over a
// 3-GiB buffer runs at 6.9 TB/s on some allocations and at 5.7 TB/s on others; the class belongs to the allocation.)  Allocate 3-GiB
// buffers one after the other until most of the 288 GB is taken, time a flat store-only and a flat load-only pass over each, and print
// them in allocation order: if the class follows the physical address, the order shows it (runs of one class, period, boundaries).
// usage: vram_map [n_buffers = 80] [GiB per buffer = 3]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/vram_map.hip -o tools/vram_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) st(f4* __restrict__ dst) { dst[(size_t)blockIdx.x * 256 + threadIdx.x] = f4{1.f, 2.f, 3.f, 4.f}; }
__global__ void __launch_bounds__(256) ld(const f4* __restrict__ src, f4* __restrict__ sink) {
  const f4 v = src[(size_t)blockIdx.x * 256 + threadIdx.x];
  if (v.x == 1.2345e-30f) sink[0] = v;
}
// the class-sensitive form (profiles/r04_pattern_lab_tile_maps.log: 5.3 vs 6.5 TB/s by box): persistent workgroups, each walks 128-byte
// row segments of 4096 rows at a 3072-byte row stride through its own 12.6-MB batch element (the product's static tile map), stores only
// ROT: workgroup w starts with column tile (ROT * w) mod tpw of its batch element instead of with column 0 (all 256 batch elements are
// 12 MiB = 3 x 4 MiB apart: without the rotation every workgroup is at the same address modulo 4 MiB whenever they are in step)
// PAD: the batch elements are (4096 + PAD) rows apart instead of 4096 (what if the 256 concurrently written regions were NOT a multiple of
// 4 MiB apart?)
template <int ROT, int PAD = 0>
__global__ void __launch_bounds__(512) st_tiles(char* __restrict__ dst, int n_tiles, int tpw) {
  const int tid = threadIdx.x, wg = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
  const long long row_bytes = 3072, lane_off = (long long)(tid / 8) * row_bytes + (tid % 8) * 16, step = 64 * row_bytes;
  for (int it0 = 0; it0 < tpw; ++it0) {
    const int it = ROT ? (it0 + ROT * wg) % tpw : it0;
    const int t = wg * tpw + it;
    if (t >= n_tiles) break;
    const long long base = (long long)(t / 24) * (4096 + PAD) * row_bytes + (long long)(t % 24) * 128;
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int q = 0; q < 8; ++q) *reinterpret_cast<f4*>(dst + base + lane_off + (long long)(c * 8 + q) * step) = f4{1.f, 2.f, 3.f, 4.f};
  }
}
// a buffer assembled from physical chunks of `chunk` bytes through the virtual-memory API, mapped in a SHUFFLED order (every chunk is a
// separate allocation of the driver: physically scattered by construction)
static char* vmm_buffer(size_t bytes, size_t chunk, unsigned seed) {
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  if (chunk % gran) { fprintf(stderr, "chunk %zu is not a multiple of the granularity %zu\n", chunk, gran); exit(1); }
  void* va = nullptr; CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
  const size_t n = bytes / chunk;
  std::vector<hipMemGenericAllocationHandle_t> h(n);
  for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&h[i], chunk, &prop, 0));
  std::vector<size_t> order(n); for (size_t i = 0; i < n; ++i) order[i] = i;
  for (size_t i = n - 1; i > 0; --i) { seed = seed * 1664525u + 1013904223u; std::swap(order[i], order[(seed >> 8) % (i + 1)]); }
  for (size_t i = 0; i < n; ++i) CK(hipMemMap((char*)va + i * chunk, chunk, 0, h[order[i]], 0));
  hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, bytes, &acc, 1));
  return (char*)va;
}
int main(int argc, char** argv) {
  const int want = argc > 1 ? atoi(argv[1]) : 80;
  const size_t bytes = (size_t)(argc > 2 ? atoi(argv[2]) : 3) << 30;
  std::vector<char*> bufs;
  std::vector<const char*> kind;
  if (argc > 3 && !strcmp(argv[3], "far")) {       // buffers whose 32-MiB chunks come alternately from two pools allocated ~D GiB apart in time
    // (= far apart in the driver's physical space, if it hands memory out in order): does spanning two regions make a buffer fast?
    const size_t chunk = (size_t)32 << 20, nchunk = bytes / chunk;
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    auto pool = [&](size_t n) { std::vector<hipMemGenericAllocationHandle_t> h(n); for (auto& x : h) CK(hipMemCreate(&x, chunk, &prop, 0)); return h; };
    auto assemble = [&](const std::vector<hipMemGenericAllocationHandle_t>& a, const std::vector<hipMemGenericAllocationHandle_t>& b, size_t& ia, size_t& ib, bool mix) {
      void* va = nullptr; CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
      for (size_t i = 0; i < nchunk; ++i) { const bool fromb = mix && (i & 1); CK(hipMemMap((char*)va + i * chunk, chunk, 0, fromb ? b[ib++] : a[ia++], 0)); }
      CK(hipMemSetAccess(va, bytes, &acc, 1)); return (char*)va; };
    for (int gap : {8, 32, 64}) {
      auto A = pool(nchunk * 3);                   // enough chunks for: one pure buffer + half of two mixed ones
      char* dummy = nullptr; CK(hipMalloc(&dummy, (size_t)gap << 30));
      auto Bp = pool(nchunk * 2);
      size_t ia = 0, ib = 0;
      static char names[16][64]; static int nn = 0;
      bufs.push_back(assemble(A, Bp, ia, ib, false)); snprintf(names[nn], 64, "pure, pool A (gap %d GiB)", gap); kind.push_back(names[nn++]);
      bufs.push_back(assemble(A, Bp, ia, ib, true)); snprintf(names[nn], 64, "A/B alternating, %d GiB apart", gap); kind.push_back(names[nn++]);
      bufs.push_back(assemble(A, Bp, ia, ib, true)); snprintf(names[nn], 64, "A/B alternating, %d GiB apart", gap); kind.push_back(names[nn++]);
      { size_t z = 0; std::vector<hipMemGenericAllocationHandle_t> rest(Bp.begin() + ib, Bp.end()); if (rest.size() >= nchunk) { bufs.push_back(assemble(rest, rest, z, z, false)); snprintf(names[nn], 64, "pure, pool B (gap %d GiB)", gap); kind.push_back(names[nn++]); } }
    }
  } else
  if (argc > 3 && !strcmp(argv[3], "flags")) {     // hipMalloc against hipExtMallocWithFlags: uncached / fine-grained / contiguous
    for (int rep = 0; rep < 5; ++rep) {
      char* p = nullptr; CK(hipMalloc(&p, bytes)); bufs.push_back(p); kind.push_back("hipMalloc");
      if (hipExtMallocWithFlags((void**)&p, bytes, hipDeviceMallocUncached) == hipSuccess) { bufs.push_back(p); kind.push_back("uncached"); } else (void)hipGetLastError();
      if (hipExtMallocWithFlags((void**)&p, bytes, hipDeviceMallocFinegrained) == hipSuccess) { bufs.push_back(p); kind.push_back("fine-grained"); } else (void)hipGetLastError();
      if (hipExtMallocWithFlags((void**)&p, bytes, hipDeviceMallocContiguous) == hipSuccess) { bufs.push_back(p); kind.push_back("contiguous"); } else (void)hipGetLastError();
    }
  } else
  if (argc > 3) {                                  // "vmm": a few hipMalloc buffers, then buffers assembled from 2-MiB / 32-MiB / 256-MiB / 1-GiB chunks
    for (int i = 0; i < 6; ++i) { char* p = nullptr; CK(hipMalloc(&p, bytes)); bufs.push_back(p); kind.push_back("hipMalloc"); }
    for (int rep = 0; rep < 3; ++rep) {
      bufs.push_back(vmm_buffer(bytes, (size_t)2 << 20, 1 + rep)); kind.push_back("vmm 2 MiB chunks, shuffled");
      bufs.push_back(vmm_buffer(bytes, (size_t)32 << 20, 11 + rep)); kind.push_back("vmm 32 MiB chunks, shuffled");
      bufs.push_back(vmm_buffer(bytes, (size_t)256 << 20, 21 + rep)); kind.push_back("vmm 256 MiB chunks, shuffled");
      bufs.push_back(vmm_buffer(bytes, (size_t)1 << 30, 31 + rep)); kind.push_back("vmm 1 GiB chunks, shuffled");
    }
  } else
  for (int i = 0; i < want; ++i) { char* p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; } bufs.push_back(p); kind.push_back("hipMalloc"); }
  printf("%zu buffers of %.0f GiB\n", bufs.size(), bytes / 1073741824.0);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f4* sink; CK(hipMalloc(&sink, 64));
  const dim3 grid((unsigned)(bytes / 4096));
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(st, grid, dim3(256), 0, 0, (f4*)bufs[0]);
  for (int pass = 0; pass < 2; ++pass)
    for (size_t i = 0; i < bufs.size(); ++i) {
      float ms[9];
      const int n_tiles = (int)(bytes / (4096 * 3072)) * 24, tpw = (n_tiles + 255) / 256;
      for (int w = 0; w < 9; ++w) {
        auto go = [&] { if (w == 0) hipLaunchKernelGGL(st, grid, dim3(256), 0, 0, (f4*)bufs[i]); else if (w == 1) hipLaunchKernelGGL(ld, grid, dim3(256), 0, 0, (const f4*)bufs[i], sink);
                        else if (w == 2) hipLaunchKernelGGL(st_tiles<0>, dim3(256), dim3(512), 0, 0, bufs[i], n_tiles, tpw);
                        else if (w == 3) hipLaunchKernelGGL(st_tiles<1>, dim3(256), dim3(512), 0, 0, bufs[i], n_tiles, tpw);
                        else if (w == 4) hipLaunchKernelGGL(st_tiles<5>, dim3(256), dim3(512), 0, 0, bufs[i], n_tiles, tpw);
                        else if (w == 5) hipLaunchKernelGGL((st_tiles<0, 1>), dim3(240), dim3(512), 0, 0, bufs[i], 240 * 24, 24);
                        else if (w == 6) hipLaunchKernelGGL((st_tiles<0, 16>), dim3(240), dim3(512), 0, 0, bufs[i], 240 * 24, 24);
                        else if (w == 7) hipLaunchKernelGGL((st_tiles<0, 64>), dim3(240), dim3(512), 0, 0, bufs[i], 240 * 24, 24);
                        else hipLaunchKernelGGL((st_tiles<0, 171>), dim3(240), dim3(512), 0, 0, bufs[i], 240 * 24, 24); };
        for (int r = 0; r < 3; ++r) go();
        CK(hipEventRecord(e0)); for (int r = 0; r < 8; ++r) go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[w], e0, e1)); ms[w] /= 8;
      }
      printf("[%s] ", kind[i]);
      printf("pass %d buffer %3zu  va %p  store %.4f ms %7.1f GB/s   load %.4f ms %7.1f GB/s   tile-shaped store %.4f ms %7.1f GB/s   rot1 %.4f ms  rot5 %.4f ms   240 elements padded by 1 / 16 / 64 / 171 rows: %.4f %.4f %.4f %.4f ms (x 256/240: %.4f %.4f %.4f %.4f)\n", pass, i, (void*)bufs[i], ms[0], bytes / ms[0] / 1e6, ms[1], bytes / ms[1] / 1e6,
             ms[2], (double)n_tiles * 4096 * 128 / ms[2] / 1e6, ms[3], ms[4], ms[5], ms[6], ms[7], ms[8], ms[5] * 256 / 240, ms[6] * 256 / 240, ms[7] * 256 / 240, ms[8] * 256 / 240);
    }
  return 0;
}
