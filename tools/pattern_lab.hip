// pattern_lab — what does the memory system want from a kernel that moves the spectral mix's bytes in the spectral mix's TILES?
// copy_ceiling (r04) found that a flat one-workgroup-per-4-KiB copy reaches 6.3 TB/s (6.57 with nt) where every persistent copy stays at
// 5.5-5.8: at any time the flat copy's traffic sits in ONE compact, moving window of the address space.  This lab keeps the product's
// tile shape (S bytes of 4096 consecutive rows at the 3072-byte row stride, S = 64 / 128) and varies WHO moves WHICH tile WHEN:
//   map 0  static, far apart: gang g walks tiles [g * tpw * GANG, ...) — the product kernel's order (every gang in its own region)
//   map 1  static, compact:   iteration it of the whole grid covers tiles [it * n_wg, (it + 1) * n_wg) — one moving window
//   map 2  dynamic:           a gang takes the next GANG tiles from an atomic counter
//   map 3  flat:              one workgroup per tile, non-persistent (the dispatcher is the counter)
//   rot 1  workgroup w starts its row walk at chunk (w / GANG) % chunks-per-tile and wraps (neighbours in different rows at the same time)
// usage: pattern_lab [reps] [D] [short]     (D = channels per row, default 768: the row stride in floats; short = only the product's maps)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pattern_lab.hip -o tools/pattern_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_contiguous(int wg, int n) {
  const int nx = 8;
  const int q = n / nx, rem = n % nx;
  const int xcd = wg % nx, idx = wg / nx;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
}

struct LabArgs {
  const char* src; char* dst;
  long long row_bytes;
  int seg, tile_rows, mode;     // mode 0 copy, 1 load, 2 store
  int n_tiles, cols, tpw, gang, map, rot, n_wg;
  unsigned* counter;
};

template <int U, int THREADS, bool NT>
__global__ void __launch_bounds__(THREADS) lab_kernel(const LabArgs a) {
  const int tid = threadIdx.x;
  const int lps = a.seg / 16;                                            // lanes per segment
  const long long lane_off = (long long)(tid / lps) * a.row_bytes + (tid % lps) * 16;
  const int rows_per_inst = THREADS / lps;
  const long long step = (long long)rows_per_inst * a.row_bytes;         // bytes between instructions
  const int chunks = a.tile_rows / (rows_per_inst * U);
  const int wg = a.map == 3 ? blockIdx.x : xcd_contiguous(blockIdx.x, gridDim.x);
  const int member = wg % a.gang, g = wg / a.gang;
  __shared__ unsigned next_s;
  f4 v[U];
#pragma unroll
  for (int q = 0; q < U; ++q) v[q] = f4{1.f, 2.f, 3.f, 4.f};
  for (int it = 0;; ++it) {
    int t;
    if (a.map == 0) { if (it >= a.tpw) break; t = g * a.tpw * a.gang + member + a.gang * it; }
    else if (a.map == 1) { t = it * a.n_wg + wg; }
    else if (a.map == 4) { const int b = blockIdx.x; t = it * a.n_wg + (a.gang == 2 ? ((b / 16) * 8 + b % 8) * 2 + (b / 8) % 2 : b); }   // compact, neighbours on different XCDs
    else if (a.map == 2) {
      if (tid == 0) next_s = atomicAdd(a.counter, 1u);                   // (per workgroup: gang members take consecutive tickets almost always)
      __syncthreads();
      t = (int)next_s;
      __syncthreads();
    } else { if (it > 0) break; t = a.gang == 2 ? (wg / 16) * 16 + (wg % 8) * 2 + ((wg / 8) % 2) : wg; }   // pairs (b, b + 8) share an XCD
    if (t >= a.n_tiles) break;
    const long long base = (long long)(t / a.cols) * a.tile_rows * a.row_bytes + (long long)(t % a.cols) * a.seg;
    const int c0 = a.rot ? (g % chunks) : 0;
    for (int cc = 0; cc < chunks; ++cc) {
      int c = cc + c0; if (c >= chunks) c -= chunks;
      const long long off = base + lane_off + (long long)c * U * step;
      if (a.mode != 2) {
#pragma unroll
        for (int q = 0; q < U; ++q) { const f4* p = reinterpret_cast<const f4*>(a.src + off + q * step); v[q] = NT ? __builtin_nontemporal_load(p) : *p; }
      }
      if (a.mode != 1) {
#pragma unroll
        for (int q = 0; q < U; ++q) { f4* p = reinterpret_cast<f4*>(a.dst + off + q * step); if (NT) __builtin_nontemporal_store(v[q], p); else *p = v[q]; }
      } else {
#pragma unroll
        for (int q = 0; q < U; ++q) if (v[q].x == 1.2345e-30f) *reinterpret_cast<f4*>(a.dst + off + q * step) = v[q];
      }
    }
  }
}

__global__ void __launch_bounds__(256) flat_copy(const f4* __restrict__ src, f4* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  dst[i] = src[i];
}
__global__ void __launch_bounds__(256) flat_copy_nt(const f4* __restrict__ src, f4* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const long long B = 256, N = 4096, D = argc > 2 ? atoll(argv[2]) : 768, row_bytes = D * 4, rows = B * N;
  const bool brief = argc > 3;
  const size_t bytes = (size_t)rows * row_bytes;
  char *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  unsigned* counter; CK(hipMalloc(&counter, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto f) {
    for (int i = 0; i < 4; ++i) f();
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
  for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(flat_copy, dim3((unsigned)(bytes / 4096)), dim3(256), 0, 0, (const f4*)a, (f4*)b);
  CK(hipDeviceSynchronize());
  printf("flat float4 copy (1 WG per 4 KiB)          %.4f ms  %.1f GB/s\n", time([&] { hipLaunchKernelGGL(flat_copy, dim3((unsigned)(bytes / 4096)), dim3(256), 0, 0, (const f4*)a, (f4*)b); }), 0.0);
  { const float ms = time([&] { hipLaunchKernelGGL(flat_copy, dim3((unsigned)(bytes / 4096)), dim3(256), 0, 0, (const f4*)a, (f4*)b); }); printf("flat float4 copy                           %.4f ms  %.1f GB/s\n", ms, 2.0 * bytes / ms / 1e6); }
  { const float ms = time([&] { hipLaunchKernelGGL(flat_copy_nt, dim3((unsigned)(bytes / 4096)), dim3(256), 0, 0, (const f4*)a, (f4*)b); }); printf("flat float4 copy, nt                       %.4f ms  %.1f GB/s\n", ms, 2.0 * bytes / ms / 1e6); }

  auto run = [&](int seg, int map, int rot, int per_cu, int mode, bool nt, int threads) {
    LabArgs x{};
    x.src = a; x.dst = b; x.row_bytes = row_bytes; x.seg = seg; x.tile_rows = 4096; x.mode = mode;
    x.gang = seg < 128 ? 128 / seg : 1; x.cols = (int)(row_bytes / seg); x.n_tiles = (int)(rows / 4096) * x.cols;
    x.map = map; x.rot = rot; x.counter = counter;
    int slots = cus * per_cu / x.gang * x.gang;
    x.tpw = (x.n_tiles + slots - 1) / slots;
    x.n_wg = map == 3 ? x.n_tiles : x.gang * ((x.n_tiles + x.gang * x.tpw - 1) / (x.gang * x.tpw));
    auto go = [&] {
      if (map == 2) CK(hipMemsetAsync(counter, 0, 4, 0));
      if (threads == 512) { if (nt) hipLaunchKernelGGL((lab_kernel<8, 512, true>), dim3(x.n_wg), dim3(512), 0, 0, x); else hipLaunchKernelGGL((lab_kernel<8, 512, false>), dim3(x.n_wg), dim3(512), 0, 0, x); }
      else { if (nt) hipLaunchKernelGGL((lab_kernel<8, 256, true>), dim3(x.n_wg), dim3(256), 0, 0, x); else hipLaunchKernelGGL((lab_kernel<8, 256, false>), dim3(x.n_wg), dim3(256), 0, 0, x); }
    };
    const float ms = time(go);
    const double nb = (mode == 0 ? 2.0 : 1.0) * bytes;
    static const char* mapn[5] = {"static far", "static compact", "dynamic", "flat", "compact xcd-il"};
    static const char* moden[3] = {"copy", "load", "store"};
    printf("seg %3d  %-14s rot %d  WG/CU %d  T=%d %s %-5s  %.4f ms  %7.1f GB/s\n", seg, mapn[map], rot, per_cu, threads, nt ? "nt" : "  ", moden[mode], ms, nb / ms / 1e6);
    fflush(stdout);
  };
  if (brief) {     // the row-stride question: does the column walk pile up on a few L2 / memory channels?
    printf("D = %lld (row stride %lld B = %.2f lines)\n", D, row_bytes, row_bytes / 128.0);
    for (int seg : {128, 64}) for (int mode : {0, 1, 2}) run(seg, 0, 0, 1, mode, false, 512);
    return 0;
  }
  for (int seg : {128, 64}) {
    for (int mode : {0}) {
      for (int map : {0, 2, 4})
        for (int rot : {0})
          for (int per_cu : {1, 2, 4}) {
            if (map == 3 && per_cu != 1) continue;
            run(seg, map, rot, per_cu, mode, false, 512);
          }
      run(seg, 4, 0, 1, mode, true, 512);
    }
  }
  for (int mode : {1, 2}) for (int map : {0, 2, 4}) run(128, map, 0, 1, mode, false, 512);
  return 0;
}
