// segcopy_bench — HBM microbenchmark that decides the tile shape of the spectral-mix kernel.
//
// The spectral-mix kernel owns, per workgroup, ALL n_fft rows of a tile of T adjacent channels of
// a (B, N, D) tensor whose D axis is contiguous.  Each row of the tile is therefore one SEG-byte
// segment (SEG = T * sizeof(elem)) and consecutive rows are D*sizeof(elem) bytes apart.  On-chip
// capacity (512 KiB VGPR + 160 KiB LDS per CU) caps SEG*N at ~256 KiB, i.e. SEG = 64 B at N = 4096.
// This tool measures what HBM bandwidth that access pattern reaches on MI355X as a function of
//   SEG            : 16 / 32 / 64 / 128 / 256 bytes per row per workgroup
//   compute delay  : a dependent-FMA phase between the loads and the stores (mimics the FFT phase)
//   tile order     : which (batch, channel-tile) a workgroup id maps to (XCD-aware or not)
// against a plain contiguous float4 copy.  Every variant copies in -> out exactly, which is verified.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/segcopy_bench.hip -o tools/segcopy_bench
// Run  :  tools/segcopy_bench [B N D]          (defaults 256 4096 768, fp32)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// ---------------------------------------------------------------------------------------------
// contiguous baseline
__global__ void __launch_bounds__(256) copy_contig(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) out[i] = in[i];
}

// ---------------------------------------------------------------------------------------------
// Tile copy.  A workgroup copies one tile = N rows x SEG bytes.  Lane layout inside a wave:
//   p = lane % LPR  (LPR = SEG/8 lanes per row, each moving one 8-byte word)
//   r = lane / LPR  + (64/LPR) * wave   (row class)
// thread (p, r) moves rows r, r + RC, r + 2*RC, ...  (RC = row classes per workgroup), EPT rows each.
// All EPT loads are issued first (register resident tile, like the FFT kernel), then an optional
// dependent-FMA delay, then all EPT stores.
template <int SEG, int EPT>
__global__ void __launch_bounds__(512) copy_tile(const float* __restrict__ in, float* __restrict__ out,
                                                 int N, int D, int tiles_per_row, int order,
                                                 int delay_iters, float fa, float fb, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // only used to force 1 WG per CU
  constexpr int LPR = SEG / 8;
  const int RC = blockDim.x / LPR;          // row classes in the workgroup
  int t = blockIdx.x;
  if (order == 1) {
    // XCD-aware: workgroup b runs on XCD b%8 (observed).  Give each XCD a contiguous run of tiles so
    // that neighbouring channel tiles (which share 128-B lines / DRAM pages) meet in the same L2.
    const int nx = 8;
    int q = n_tiles / nx, rem = n_tiles % nx;
    int xcd = t % nx, idx = t / nx;
    t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  int b = t / tiles_per_row, ct = t % tiles_per_row;
  if (order == 2 && tiles_per_row % 8 == 0) {
    // XCD-striped: XCD x owns channel tiles [x*cpx, (x+1)*cpx) of EVERY batch element, so all 8 XCDs stream the
    // same rows (same DRAM pages) at the same time instead of rows 2^27-aligned apart.
    const int cpx = tiles_per_row / 8;
    const int xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
    ct = xcd * cpx + idx % cpx;
    b = idx / cpx;
  }
  const int p = threadIdx.x % LPR;
  const int r = threadIdx.x / LPR;
  // uniform (scalar) tile base + one 32-bit per-lane byte offset -> saddr-form global loads
  const char* sin_ = reinterpret_cast<const char*>(in + (size_t)b * N * D + (size_t)ct * (SEG / 4));
  char* sout_ = reinterpret_cast<char*>(out + (size_t)b * N * D + (size_t)ct * (SEG / 4));
  const uint32_t voff = (uint32_t)(r * D + p * 2) * 4u;
  if (delay_iters < 0) smem[threadIdx.x] = 0;  // keep smem referenced
  for (int row0 = 0; row0 < N; row0 += RC * EPT) {   // one pass when the tile fits the registers
    float2 v[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int row = row0 + r + q * RC;
      v[q] = *reinterpret_cast<const float2*>(sin_ + (size_t)(row - r) * D * 4 + voff);
    }
    for (int it = 0; it < delay_iters; ++it) {
#pragma unroll
      for (int q = 0; q < EPT; ++q) { v[q].x = fmaf(v[q].x, fa, fb); v[q].y = fmaf(v[q].y, fa, fb); }
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const int row = row0 + r + q * RC;
      *reinterpret_cast<float2*>(sout_ + (size_t)(row - r) * D * 4 + voff) = v[q];
    }
  }
}

struct Result { double ms; double gbs; };

template <int SEG, int EPT>
Result run_tile(const float* in, float* out, int B, int N, int D, int order, int delay, int lds_bytes, int iters) {
  int threads = (N / EPT) * (SEG / 8);
  if (threads > 512) threads = 512;            // larger segments: several register passes per tile
  const int tiles_per_row = D * 4 / SEG;
  const int n_tiles = B * tiles_per_row;
  if (threads < 64 || (N % ((threads / (SEG / 8)) * EPT)) != 0 || (D * 4) % SEG) return {0, 0};
  CK(hipFuncSetAttribute((const void*)copy_tile<SEG, EPT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w)
    copy_tile<SEG, EPT><<<n_tiles, threads, lds_bytes>>>(in, out, N, D, tiles_per_row, order, delay, 1.0f, 0.0f, n_tiles);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    copy_tile<SEG, EPT><<<n_tiles, threads, lds_bytes>>>(in, out, N, D, tiles_per_row, order, delay, 1.0f, 0.0f, n_tiles);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
  double bytes = 2.0 * B * (double)N * D * 4;
  return {ms, bytes / ms / 1e6};
}

int main(int argc, char** argv) {
  int B = 256, N = 4096, D = 768;
  if (argc >= 4) { B = atoi(argv[1]); N = atoi(argv[2]); D = atoi(argv[3]); }
  const size_t n = (size_t)B * N * D;
  float *in, *out;
  CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
  {
    std::vector<float> h(1 << 22);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    for (size_t off = 0; off < n; off += h.size())
      CK(hipMemcpy(in + off, h.data(), std::min(h.size(), n - off) * 4, hipMemcpyHostToDevice));
  }
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s CUs=%d clock=%d MHz  shape B=%d N=%d D=%d (%.2f GB per tensor)\n", prop.name,
         prop.multiProcessorCount, prop.clockRate / 1000, B, N, D, n * 4 / 1e9);
  const int iters = 10;
  // contiguous baseline
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) copy_contig<<<2048, 256>>>((const float4*)in, (float4*)out, n / 4);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) copy_contig<<<2048, 256>>>((const float4*)in, (float4*)out, n / 4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    printf("contig float4 grid-stride        : %8.3f ms  %8.1f GB/s\n", ms, 2.0 * n * 4 / ms / 1e6);
  }
  // tile variants:  (SEG, order, delay, lds)
  struct Cfg { int seg, order, delay, lds; };
  std::vector<Cfg> cfgs;
  for (int seg : {32, 64, 128})
    for (int order : {1, 2})
      for (int delay : {0, 60})
        for (int lds : {100 * 1024})
          cfgs.push_back({seg, order, delay, lds});
  printf("# SEG order delay lds_KB : ms GB/s   (EPT = rows per thread; order 1 = XCD-contiguous tiles)\n");
  for (auto c : cfgs) {
    Result r{0, 0}; int ept = 0;
    // pick EPT so the workgroup has 512 threads when possible (the FFT kernel's geometry)
    switch (c.seg) {
      case 16:  ept = 16; r = (N == 4096) ? run_tile<16, 16>(in, out, B, N, D, c.order, c.delay, c.lds, iters) : Result{0,0}; break;
      case 32:  ept = 32; r = run_tile<32, 32>(in, out, B, N, D, c.order, c.delay, c.lds, iters); break;
      case 64:  ept = 64; r = run_tile<64, 64>(in, out, B, N, D, c.order, c.delay, c.lds, iters); break;
      case 128: ept = 64; r = run_tile<128, 64>(in, out, B, N, D, c.order, c.delay, c.lds, iters); break;
      case 256: ept = 64; r = run_tile<256, 64>(in, out, B, N, D, c.order, c.delay, c.lds, iters); break;
    }
    if (r.ms > 0)
      printf("SEG=%3d EPT=%2d order=%d delay=%3d lds=%3dKB : %8.3f ms  %8.1f GB/s\n", c.seg, ept, c.order, c.delay,
             c.lds / 1024, r.ms, r.gbs);
    fflush(stdout);
  }
  // verify the last variant's output
  {
    std::vector<float> a(1 << 20), b2(1 << 20);
    CK(hipMemcpy(a.data(), in + (n / 2), a.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b2.data(), out + (n / 2), b2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < a.size(); ++i) bad += (a[i] != b2[i]);
    printf("# verify mismatches: %zu\n", bad);
  }
  return 0;
}
