// iolab3 — how wide must a row segment be before the strided tile pattern of the 4096 kernel moves at the speed of a dense copy?
// One persistent 512-thread workgroup per CU moves 256-KiB tiles of a (B*N rows) x 3072-byte matrix.  A tile is S bytes of
// (256 KiB / S) consecutive rows; tiles next to each other in the launch order are next to each other in the row (the kernel's
// order).  S = 64 is the fp32 kernel (16 channels), S = 32 would be bf16, "dense" = 256 KiB contiguous.
// Part G: the same with groups of 2 / 4 / 8 neighbouring workgroups released together tile by tile (a counter per group in L2), to
// see whether requests for the two halves of a 128-byte line that arrive together are served as one.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/iolab3.hip -o tools/iolab3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int xcd_tile(int t, int n) { const int q = n / 8, rem = n % 8, x = t % 8, i = t / 8; return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + i; }

constexpr int kTile = 256 * 1024, kRow = 3072, kPer = kTile / 512 / 16;   // 32 x 16 bytes per thread and tile

// MODE 0 load, 1 store, 2 load then store (same tile of another buffer)
template <int MODE>
__global__ void __launch_bounds__(512) seg_io(const char* __restrict__ in, char* __restrict__ out, int S, int n_tiles, int tpw, int gs, unsigned* cnt, int split, int fs) {
  extern __shared__ char smem[];
  if (S < 0) smem[threadIdx.x] = 0;
  const int wg = xcd_tile(blockIdx.x, gridDim.x);
  const int tid = threadIdx.x;
  size_t lane_off, step;
  int cols;
  if (S == 0) { cols = 1; lane_off = (size_t)tid * 16; step = 512 * 16; }
  else { const int lps = S / 16, rpi = 512 / lps; cols = kRow / S; lane_off = (size_t)(tid / lps) * kRow + (tid % lps) * 16; step = (size_t)rpi * kRow; }
  const size_t rows_per_tile = S == 0 ? 0 : kTile / S;
  f32x4 v[kPer];
#pragma unroll
  for (int q = 0; q < kPer; ++q) v[q] = f32x4{1.f, 2.f, 3.f, 4.f};
  for (int it = 0; it < tpw; ++it) {
    const int t = wg + it * gridDim.x;
    if (t >= n_tiles) break;
    if (gs > 1) {
      if (tid == 0) {
        unsigned* c = cnt + (wg / gs) * 32;
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // relaxed: a release would write the L2 back
        const unsigned want = (unsigned)(it + 1) * gs;
        for (int spin = 0; spin < (1 << 18); ++spin) {
          if (__hip_atomic_fetch_add(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;   // an RMW is served by the L2
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
    }
    size_t base = S == 0 ? (size_t)t * kTile : ((size_t)(t / cols) * rows_per_tile) * kRow + (size_t)(t % cols) * S;
    // Part H (S = 64 only): the two 64-byte halves of a 128-byte line requested by the SAME workgroup at a chosen distance in time.
    //   split 1: 128 B x 2048 rows, instruction 2i = first halves of 128 rows, 2i+1 = their second halves (tens of ns apart)
    //   split 2: the same tile, all first halves, then all second halves (half a tile apart)
    //   split 3: 64 B x 4096 rows; the tile of iteration 2m+1 is the right-hand neighbour of iteration 2m's (a tile apart)
    size_t qoff[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) qoff[q] = q * step;
    if (split >= 4) {   // halves 2^(split-3) instructions apart: q = (block, half, index inside block)
      base = ((size_t)(t / 24) * 2048) * kRow + (size_t)(t % 24) * 128;
      const int gap = 1 << (split - 3);
#pragma unroll
      for (int q = 0; q < kPer; ++q) { const int blk = q / (2 * gap), half = (q / gap) & 1, rb = blk * gap + q % gap; qoff[q] = (size_t)rb * 128 * kRow + half * 64; }
    } else if (split == 1 || split == 2) {
      base = ((size_t)(t / 24) * 2048) * kRow + (size_t)(t % 24) * 128;
#pragma unroll
      for (int q = 0; q < kPer; ++q) { const int half = split == 1 ? (q & 1) : (q >> 4), rb = split == 1 ? (q >> 1) : (q & 15); qoff[q] = (size_t)rb * 128 * kRow + half * 64; }
    } else if (split == 3) {
      const int pi = wg + (it >> 1) * gridDim.x;
      base = ((size_t)(pi / 24) * 4096) * kRow + (size_t)(2 * (pi % 24) + (it & 1)) * 64;
    }
    if (MODE != 1) {
      const char* p = in + base + lane_off;
#pragma unroll
      for (int q = 0; q < kPer; ++q) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v[q]) : "v"(p + qoff[q]));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < kPer; ++q) asm volatile("" : "+v"(v[q]));
    }
    if (MODE != 0) {
      char* p = out + base + lane_off;
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        // Part I: the members of a gang re-align every fs store instructions.  Arrival = atomic without return (no wait), poll =
        // scalar load with glc (answered by the L2, counted by lgkmcnt: the wave's stores stay in flight).
        if (fs < 0 && q % (-fs) == 0) {   // Part J: wave w of workgroup 2m re-aligns with wave w of workgroup 2m+1 only (they own the two halves of
          // the same lines); scalar atomic + scalar poll: no VGPR, no vmcnt, no workgroup barrier
          unsigned* c = cnt + ((wg >> 1) * 8 + __builtin_amdgcn_readfirstlane(tid >> 6)) * 16;
          const unsigned one = 1, want = ((unsigned)it * (kPer / -fs) + q / -fs + 1) * 2;
          asm volatile("s_atomic_add %0, %1, 0x0" :: "s"(one), "s"(c) : "memory");
          for (int spin = 0; spin < (1 << 12); ++spin) {
            unsigned now;
            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(now) : "s"(c) : "memory");
            if (now >= want) break;
          }
        }
        if (fs > 0 && gs > 1 && q % fs == 0) {
          if (tid == 0) {
            unsigned* c = cnt + (wg / gs) * 32 + 16;
            __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = ((unsigned)it * (kPer / fs) + q / fs + 1) * gs;
            for (int spin = 0; spin < (1 << 14); ++spin) {
              unsigned now;
              asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(now) : "s"(c) : "memory");
              if (now >= want) break;
            }
          }
          __syncthreads();
        }
        asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p + qoff[q]), "v"(v[q]) : "memory");
      }
      if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
}

static hipEvent_t e0, e1;
static unsigned* g_cnt;
template <int MODE> void run(const char* in, char* out, int S, int gs, int split = 0, int fs = 0, int grid = 256) {
  const size_t total = (size_t)256 * 4096 * kRow;
  const int n_tiles = (int)(total / kTile), tpw = (n_tiles + grid - 1) / grid;
  CK(hipFuncSetAttribute((const void*)seg_io<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto f = [&] { CK(hipMemsetAsync(g_cnt, 0, 65536 * 4)); seg_io<MODE><<<grid, 512, 133 * 1024>>>(in, out, S, n_tiles, tpw, gs, g_cnt, split, fs); };
  f(); CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0;
  for (int i = 0; i < 5; ++i) {
    CK(hipMemsetAsync(g_cnt, 0, 65536 * 4));
    CK(hipEventRecord(e0)); seg_io<MODE><<<grid, 512, 133 * 1024>>>(in, out, S, n_tiles, tpw, gs, g_cnt, split, fs); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
  }
  const char* m[] = {"load ", "store", "copy "};
  const double bytes = (MODE == 2 ? 2.0 : 1.0) * n_tiles * (double)kTile;
  printf("%s segment=%5d B  group=%d split=%d fs=%d : mean %7.3f ms  best %7.3f ms  %7.1f GB/s (mean)  %6.2f us/tile/CU\n", m[MODE], S == 0 ? kTile : S, gs, split, fs, sum / 5, best,
         bytes / (sum / 5) / 1e6, sum / 5 * 1e3 / tpw);
  fflush(stdout);
}
int main() {
  const size_t n = (size_t)256 * 4096 * kRow;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  char *in, *out; CK(hipMalloc(&in, n)); CK(hipMalloc(&out, n)); CK(hipMemset(in, 0x3c, n)); CK(hipMemset(out, 0, n));
  CK(hipMalloc(&g_cnt, 65536 * 4));
  for (int S : {16, 32, 64, 128, 256, 512, 1024, 0}) { run<0>(in, out, S, 1); run<1>(in, out, S, 1); run<2>(in, out, S, 1); }
  printf("-- the halves of a line requested by one workgroup, at a distance in time --\n");
  for (int sp : {1, 4, 5, 6, 2, 3}) { run<0>(in, out, 64, 1, sp); run<1>(in, out, 64, 1, sp); run<2>(in, out, 64, 1, sp); }
  printf("-- groups of neighbouring workgroups released together --\n");
  for (int gs : {2, 4, 8}) { run<0>(in, out, 64, gs); run<1>(in, out, 64, gs); run<2>(in, out, 64, gs); }
  for (int gs : {2, 4}) { run<2>(in, out, 128, gs); }
  printf("-- wave pairs re-aligned every |fs| store instructions (scalar atomics) --\n");
  for (int fs : {-16, -8, -4, -2, -1}) { run<1>(in, out, 64, 1, 0, fs); run<2>(in, out, 64, 1, 0, fs); }
  printf("-- gangs re-aligned every fs store instructions --\n");
  for (int gs : {2, 4, 8}) for (int fs : {8, 4, 2, 1}) { run<1>(in, out, 64, gs, 0, fs); run<2>(in, out, 64, gs, 0, fs); }
  return 0;
}
