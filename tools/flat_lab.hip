// flat_lab — WHICH property of the flat float4 copy (6.26 TB/s on 3 GiB -> 3 GiB, one 256-thread workgroup per 4 KiB) do the persistent
// and tile-shaped copies (4.9-5.8 TB/s) lose?  Starting from the flat copy, one property is changed at a time:
//   perm R     consecutive workgroups still sweep R bytes sequentially, but the R-byte regions are visited in bit-reversed order
//              (R = 3 GiB: the flat copy; small R: the traffic of the ~2000 resident workgroups is scattered over the whole buffer)
//   rounds K   a workgroup moves K consecutive 4-KiB chunks one after the other (load, store, load, store ...): persistence without depth
//   depth U    a workgroup loads U chunks, then stores them (more bytes in flight per wave)
//   tile S     NON-persistent but tile-shaped: a workgroup moves S bytes of 4096 / S * ... rows = 4 KiB of one S-byte column (rows at the
//              3072-byte row stride); order A: consecutive workgroups = consecutive row blocks of ONE column (then the next column);
//              order B: consecutive workgroups = the same row block of consecutive columns (a row-major sweep: contiguous in memory)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/flat_lab.hip -o tools/flat_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bitrev(unsigned x, int bits) { return __brev(x) >> (32 - bits); }

// chunk index (4 KiB units) of workgroup b: low `lo` bits stay (sequential inside a region), the upper `hi` bits are bit-reversed
__global__ void __launch_bounds__(256) k_perm(const f4* __restrict__ src, f4* __restrict__ dst, int lo, int hi, unsigned nchunks) {
  const unsigned b = blockIdx.x;
  unsigned up = b >> lo;
  // nchunks is not a power of two (3 GiB = 3 * 2^18 chunks): reverse inside the largest power of two, leave the remainder in place
  const unsigned pow2 = 1u << (lo + hi);
  unsigned c = b;
  if (b < pow2) c = (bitrev(up, hi) << lo) | (b & ((1u << lo) - 1));
  if (c >= nchunks) return;
  const size_t i = (size_t)c * 256 + threadIdx.x;
  dst[i] = src[i];
}

template <int K>
__global__ void __launch_bounds__(256) k_rounds(const f4* __restrict__ src, f4* __restrict__ dst) {
  const size_t base = (size_t)blockIdx.x * 256 * K + threadIdx.x;
#pragma unroll 1
  for (int k = 0; k < K; ++k) dst[base + (size_t)k * 256] = src[base + (size_t)k * 256];
}

template <int U>
__global__ void __launch_bounds__(256) k_depth(const f4* __restrict__ src, f4* __restrict__ dst) {
  const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int k = 0; k < U; ++k) v[k] = src[base + (size_t)k * 256];
#pragma unroll
  for (int k = 0; k < U; ++k) dst[base + (size_t)k * 256] = v[k];
}

// tile-shaped, non-persistent: workgroup = S bytes x (4096 / S) rows... 256 lanes x 16 B = 4 KiB = S bytes of RPW = 4096 / S rows
__global__ void __launch_bounds__(256) k_tile(const char* __restrict__ src, char* __restrict__ dst, int S, int order, long long row_bytes, int rows_per_tile, unsigned n_wg) {
  const int lps = S / 16, rpw = 256 / lps;             // lanes per segment, rows per workgroup
  const int cols = (int)(row_bytes / S);               // S-byte columns per row
  const int rb_per_tile = rows_per_tile / rpw;         // row blocks per 4096-row tile
  const unsigned b = blockIdx.x;
  long long batch, col, rb;
  if (order == 0) {                                    // A: row blocks of one column first
    rb = b % rb_per_tile; col = (b / rb_per_tile) % cols; batch = b / ((long long)rb_per_tile * cols);
  } else {                                             // B: the same row block of all columns first (row-major sweep)
    col = b % cols; rb = (b / cols) % rb_per_tile; batch = b / ((long long)rb_per_tile * cols);
  }
  const long long off = (batch * rows_per_tile + rb * rpw + threadIdx.x / lps) * row_bytes + col * S + (threadIdx.x % lps) * 16;
  *reinterpret_cast<f4*>(dst + off) = *reinterpret_cast<const f4*>(src + off);
}

// persistent tile walker WITHOUT depth: a workgroup owns one S-byte column of 4096 rows and moves it 4 KiB at a time (load, store, ...)
__global__ void __launch_bounds__(256) k_tile_walk(const char* __restrict__ src, char* __restrict__ dst, int S, long long row_bytes, int rows_per_tile, int rounds_in_flight) {
  const int lps = S / 16, rpw = 256 / lps;
  const int cols = (int)(row_bytes / S);
  const unsigned b = blockIdx.x;
  const long long batch = b / cols, col = b % cols;
  const long long base = (batch * rows_per_tile + threadIdx.x / lps) * row_bytes + col * S + (threadIdx.x % lps) * 16;
  const long long step = (long long)rpw * row_bytes;
  const int n = rows_per_tile / rpw;
  if (rounds_in_flight == 1) {
#pragma unroll 1
    for (int k = 0; k < n; ++k) *reinterpret_cast<f4*>(dst + base + k * step) = *reinterpret_cast<const f4*>(src + base + k * step);
  } else {
#pragma unroll 1
    for (int k = 0; k < n; k += 8) {
      f4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f4*>(src + base + (k + q) * step);
#pragma unroll
      for (int q = 0; q < 8; ++q) *reinterpret_cast<f4*>(dst + base + (k + q) * step) = v[q];
    }
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const long long B = 256, N = 4096, D = 768, row_bytes = D * 4;
  const size_t bytes = (size_t)B * N * row_bytes;
  const unsigned nchunks = (unsigned)(bytes / 4096);
  char *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto f) {
    for (int i = 0; i < 4; ++i) f();
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
  auto rep = [&](const char* name, float ms) { printf("%-72s %.4f ms  %7.1f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6); fflush(stdout); };
  for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k_perm, dim3(nchunks), dim3(256), 0, 0, (const f4*)a, (f4*)b, 20, 0, nchunks);
  CK(hipDeviceSynchronize());
  char nm[160];
  // total chunk-index bits covered by the permutation: 2^19 chunks = 2 GiB (the remaining GiB stays sequential)
  for (int lo : {19, 13, 11, 9, 7, 5, 4, 3, 2, 1, 0}) {
    snprintf(nm, sizeof nm, "flat, regions of %6d KiB visited in bit-reversed order", 4 << lo);
    rep(nm, time([&] { hipLaunchKernelGGL(k_perm, dim3(nchunks), dim3(256), 0, 0, (const f4*)a, (f4*)b, lo, 19 - lo, nchunks); }));
  }
  rep("rounds K=2  (WG moves 2 consecutive chunks, one after the other)", time([&] { hipLaunchKernelGGL(k_rounds<2>, dim3(nchunks / 2), dim3(256), 0, 0, (const f4*)a, (f4*)b); }));
  rep("rounds K=4", time([&] { hipLaunchKernelGGL(k_rounds<4>, dim3(nchunks / 4), dim3(256), 0, 0, (const f4*)a, (f4*)b); }));
  rep("rounds K=16", time([&] { hipLaunchKernelGGL(k_rounds<16>, dim3(nchunks / 16), dim3(256), 0, 0, (const f4*)a, (f4*)b); }));
  rep("rounds K=64", time([&] { hipLaunchKernelGGL(k_rounds<64>, dim3(nchunks / 64), dim3(256), 0, 0, (const f4*)a, (f4*)b); }));
  rep("rounds K=256 (3072 workgroups of 1 MiB each)", time([&] { hipLaunchKernelGGL(k_rounds<256>, dim3(nchunks / 256), dim3(256), 0, 0, (const f4*)a, (f4*)b); }));
  rep("depth U=2  (WG loads 2 chunks, then stores them)", time([&] { hipLaunchKernelGGL(k_depth<2>, dim3(nchunks / 2), dim3(256), 0, 0, (const f4*)a, (f4*)b); }));
  rep("depth U=4", time([&] { hipLaunchKernelGGL(k_depth<4>, dim3(nchunks / 4), dim3(256), 0, 0, (const f4*)a, (f4*)b); }));
  rep("depth U=8", time([&] { hipLaunchKernelGGL(k_depth<8>, dim3(nchunks / 8), dim3(256), 0, 0, (const f4*)a, (f4*)b); }));
  rep("depth U=16", time([&] { hipLaunchKernelGGL(k_depth<16>, dim3(nchunks / 16), dim3(256), 0, 0, (const f4*)a, (f4*)b); }));
  for (int S : {1024, 512, 256, 128, 64}) {   // (S must divide 4096 bytes per workgroup AND the 3072-byte row)
    if (row_bytes % S) continue;
    for (int order : {0, 1}) {
      snprintf(nm, sizeof nm, "tile-shaped WG: %4d B x %3d rows, order %s", S, 4096 / S, order ? "B (row block of all columns first = row-major sweep)" : "A (row blocks of one column first)");
      rep(nm, time([&] { hipLaunchKernelGGL(k_tile, dim3(nchunks), dim3(256), 0, 0, a, b, S, order, row_bytes, 4096, nchunks); }));
    }
  }
  for (int S : {512, 128, 64}) {
    const unsigned n_wg = (unsigned)(B * (row_bytes / S));
    for (int rif : {1, 8}) {
      snprintf(nm, sizeof nm, "column walker: one WG per %3d-B column of 4096 rows, %d x 4 KiB in flight (%u WGs)", S, rif, n_wg);
      rep(nm, time([&] { hipLaunchKernelGGL(k_tile_walk, dim3(n_wg), dim3(256), 0, 0, a, b, S, row_bytes, 4096, rif); }));
    }
  }
  return 0;
}
