// tilepipe_bench — how much HBM traffic can ONE compute unit keep in flight, and how well does a
// persistent workgroup overlap a compute phase with its tile loads/stores?
//
// Follow-up to segcopy_bench: 64-byte row segments + XCD-contiguous tile order reach the copy
// ceiling, but a load-all / compute / store-all workgroup serialises its compute phase.  This tool
// measures, for the spectral-mix geometry (tile = N rows x 64 B, 512 threads, 64 x 8 B per thread):
//   mode 0  persistent copy, per tile {load all, wait, delay, store all}   (stores overlap next loads)
//   mode 1  persistent copy, register tile split in H parts, software pipelined: part h+1's loads are
//           in flight while part h is "computed" (delay) and stored                  (H = 2, 4)
//   mode 2  load-only   (per-CU read rate)        mode 3  store-only (per-CU write rate)
// each at grid = 1, 8, 64, 256, 512 workgroups so the per-CU caps show up on an idle chip.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/tilepipe_bench.hip -o tools/tilepipe_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int SEG = 64, LPR = 8, THREADS = 512, RC = THREADS / LPR;   // 64 row classes

__device__ __forceinline__ int xcd_tile(int t, int n_tiles) {
  const int nx = 8;
  int q = n_tiles / nx, rem = n_tiles % nx;
  int xcd = t % nx, idx = t / nx;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
}

// H = number of register parts (1 = whole tile at once). EPT = 64 rows per thread in total.
template <int H, int MODE>
__global__ void __launch_bounds__(THREADS) tile_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int D,
                                                       int tiles_per_row, int n_tiles, int tiles_per_wg,
                                                       int delay_iters, float fa, float fb, int nt, int stagger_iters, int rot, int* qctr) {
  constexpr int EPT = 64, PE = EPT / H;
  const int p = threadIdx.x % LPR, r = threadIdx.x / LPR;
  const int Deff = (rot & 4) ? SEG / 4 : D;     // dense tiles: row stride = segment
  const uint32_t voff = (uint32_t)(r * Deff + p * 2) * 4u;
  float2 v[H][PE];
  float acc = 0.f;
  // de-synchronise the persistent workgroups: pairs of neighbouring tiles (which share 128-B lines)
  // keep the same phase, 16 phases per XCD.
  if (stagger_iters > 0) {
    const int phase = ((blockIdx.x / 8) / 2) % 16;
    float d = fa;
    for (int i = 0; i < phase * stagger_iters; ++i) d = fmaf(d, fa, fb);
    acc += d * 1e-30f;
  }
  // optional rotation of the row-block order so that workgroups in the same phase hit different rows
  const int qrot = (rot & 1) ? ((blockIdx.x / 16) * 5) % PE : 0;
  // virtual persistent order: workgroup w handles tiles w, w + grid, ...  (then XCD-contiguous remap)
  auto tile_ptrs = [&](int it, const char*& si, char*& so) {
    int t = blockIdx.x + it * gridDim.x;
    t = xcd_tile(t % n_tiles, n_tiles);
    int b = t / tiles_per_row, ct = t % tiles_per_row;
    if (rot & 2) { b = (blockIdx.x * 37 + it * 11) % (n_tiles / tiles_per_row); ct = (blockIdx.x * 5 + it) % tiles_per_row; }  // scattered tiles
    if (rot & 4) {   // dense tile: 256 KiB contiguous
      si = reinterpret_cast<const char*>(in + (size_t)t * N * (SEG / 4));
      so = reinterpret_cast<char*>(out + (size_t)t * N * (SEG / 4));
      return;
    }
    si = reinterpret_cast<const char*>(in + (size_t)b * N * D + (size_t)ct * (SEG / 4));
    so = reinterpret_cast<char*>(out + (size_t)b * N * D + (size_t)ct * (SEG / 4));
  };
  auto load_part = [&](const char* si, int h) {
#pragma unroll
    for (int q = 0; q < PE; ++q) {
      const f32x2* ptr = reinterpret_cast<const f32x2*>(si + (size_t)((h * PE + ((q + qrot) % PE)) * RC) * Deff * 4 + voff);
      f32x2 t = nt ? __builtin_nontemporal_load(ptr) : *ptr;
      v[h][q] = make_float2(t.x, t.y);
    }
  };
  auto store_part = [&](char* so, int h) {
#pragma unroll
    for (int q = 0; q < PE; ++q) {
      f32x2* ptr = reinterpret_cast<f32x2*>(so + (size_t)((h * PE + ((q + qrot) % PE)) * RC) * Deff * 4 + voff);
      f32x2 t; t.x = v[h][q].x; t.y = v[h][q].y;
      if (nt) __builtin_nontemporal_store(t, ptr); else *ptr = t;
    }
  };
  auto compute_part = [&](int h) {
    for (int it = 0; it < delay_iters; ++it) {
#pragma unroll
      for (int q = 0; q < PE; ++q) { v[h][q].x = fmaf(v[h][q].x, fa, fb); v[h][q].y = fmaf(v[h][q].y, fa, fb); }
    }
  };
  const char* si; char* so;
  if (MODE == 0) {
    for (int it = 0; it < tiles_per_wg; ++it) {
      tile_ptrs(it, si, so);
#pragma unroll
      for (int h = 0; h < H; ++h) load_part(si, h);
#pragma unroll
      for (int h = 0; h < H; ++h) compute_part(h);
#pragma unroll
      for (int h = 0; h < H; ++h) store_part(so, h);
    }
  } else if (MODE == 1) {
    // software pipeline over parts: prologue loads part 0 of tile 0; steady state: issue loads of the
    // NEXT part, then compute + store the current one.
    tile_ptrs(0, si, so);
    load_part(si, 0);
    for (int it = 0; it < tiles_per_wg; ++it) {
      const char* si_n = si; char* so_n = so;
      if (it + 1 < tiles_per_wg) tile_ptrs(it + 1, si_n, so_n);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        if (h + 1 < H) load_part(si, h + 1);
        else if (it + 1 < tiles_per_wg) load_part(si_n, 0);
        compute_part(h);
        store_part(so, h);
      }
      si = si_n; so = so_n;
    }
  } else if (MODE == 4) {
    // persistent + dynamic: each XCD owns a contiguous run of tiles and a counter; workgroups pull.
    __shared__ int s_tile;
    const int nx = 8, xcd = blockIdx.x % nx;
    const int q = n_tiles / nx;             // n_tiles % 8 == 0 here
    for (;;) {
      if (threadIdx.x == 0) s_tile = atomicAdd(&qctr[xcd * 32], 1);
      __syncthreads();
      const int idx = s_tile;
      __syncthreads();
      if (idx >= q) break;
      const int t = xcd * q + idx;
      const int b = t / tiles_per_row, ct = t % tiles_per_row;
      si = reinterpret_cast<const char*>(in + (size_t)b * N * D + (size_t)ct * (SEG / 4));
      so = reinterpret_cast<char*>(out + (size_t)b * N * D + (size_t)ct * (SEG / 4));
#pragma unroll
      for (int h = 0; h < H; ++h) load_part(si, h);
#pragma unroll
      for (int h = 0; h < H; ++h) compute_part(h);
#pragma unroll
      for (int h = 0; h < H; ++h) store_part(so, h);
    }
  } else if (MODE == 2) {
    for (int it = 0; it < tiles_per_wg; ++it) {
      tile_ptrs(it, si, so);
#pragma unroll
      for (int h = 0; h < H; ++h) load_part(si, h);
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int q = 0; q < PE; ++q) acc += v[h][q].x * v[h][q].y;
    }
  } else {
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int q = 0; q < PE; ++q) v[h][q] = make_float2(fa * (q + h), fb + threadIdx.x);
    for (int it = 0; it < tiles_per_wg; ++it) {
      tile_ptrs(it, si, so);
#pragma unroll
      for (int h = 0; h < H; ++h) store_part(so, h);
    }
  }
  if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <int H, int MODE>
void run(const char* label, const float* in, float* out, int B, int N, int D, int grid, int tiles_per_wg, int delay, int nt, int stagger = 0, int rot = 0) {
  const int tiles_per_row = D * 4 / SEG, n_tiles = B * tiles_per_row;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  static int* qctr = nullptr;
  if (!qctr) CK(hipMalloc(&qctr, 8 * 32 * 4));
  auto launch = [&] { if (MODE == 4) CK(hipMemsetAsync(qctr, 0, 8 * 32 * 4)); tile_kernel<H, MODE><<<grid, THREADS>>>(in, out, N, D, tiles_per_row, n_tiles, tiles_per_wg, delay, 1.0f, 0.0f, nt, stagger, rot, qctr); };
  launch(); CK(hipDeviceSynchronize());
  const int iters = 5;
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
  const double tile_bytes = (double)N * SEG * ((MODE >= 2) ? 1 : 2);
  const double bytes = tile_bytes * (MODE == 4 ? (double)n_tiles : (double)grid * tiles_per_wg);
  printf("%-6s H=%d grid=%4d tiles/wg=%3d delay=%3d nt=%d stag=%4d rot=%d : %8.3f ms  %8.1f GB/s total  %7.2f GB/s per WG  %7.2f us/tile\n",
         label, H, grid, tiles_per_wg, delay, nt, stagger, rot, ms, bytes / ms / 1e6, bytes / ms / 1e6 / grid, ms * 1e3 / tiles_per_wg);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int B = 256, N = 4096, D = 768;
  const size_t n = (size_t)B * N * D;
  float *in, *out;
  CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
  CK(hipMemset(in, 0x3c, n * 4)); CK(hipMemset(out, 0, n * 4));
  const int n_tiles = B * (D * 4 / SEG);   // 12288
  const int part = argc > 1 ? atoi(argv[1]) : 0;
  if (part == 0) {
    // (a) per-CU caps on an idle / partly loaded chip
    for (int grid : {1, 8, 64, 256}) {
      const int tpw = 48;
      run<1, 2>("load", in, out, B, N, D, grid, tpw, 0, 0);
      run<1, 3>("store", in, out, B, N, D, grid, tpw, 0, 0);
      run<1, 0>("copy", in, out, B, N, D, grid, tpw, 0, 0);
      run<2, 1>("pipe", in, out, B, N, D, grid, tpw, 0, 0);
    }
    for (int nt : {0, 1})
      for (int delay : {0, 30, 60, 90}) {
        run<1, 0>("copy", in, out, B, N, D, 256, n_tiles / 256, delay, nt);
        run<2, 1>("pipe", in, out, B, N, D, 256, n_tiles / 256, delay, nt);
        run<4, 1>("pipe", in, out, B, N, D, 256, n_tiles / 256, delay, nt);
      }
  } else if (part == 3) {
    // (d) what slows a CU down when a few neighbours are active?  static persistent copy, 48 tiles each
    for (int grid : {8, 16, 32, 64, 128, 256})
      for (int rot : {0, 2, 4}) {
        run<1, 0>(rot == 0 ? "adjacent" : rot == 2 ? "scattered" : "dense", in, out, B, N, D, grid, 24, 0, 0, 0, rot);
      }
    for (int grid : {8, 64, 256}) {
      run<1, 2>("ld-adj", in, out, B, N, D, grid, 24, 0, 0, 0, 0);
      run<1, 2>("ld-dense", in, out, B, N, D, grid, 24, 0, 0, 0, 4);
      run<1, 3>("st-adj", in, out, B, N, D, grid, 24, 0, 0, 0, 0);
      run<1, 3>("st-dense", in, out, B, N, D, grid, 24, 0, 0, 0, 4);
    }
  } else if (part == 2) {
    // (c) from one tile per workgroup to fully persistent (static), and the dynamic per-XCD queue
    for (int delay : {0, 60})
      for (int tpw : {1, 2, 4, 12, 48})
        run<1, 0>("copy", in, out, B, N, D, n_tiles / tpw, tpw, delay, 0);
    for (int delay : {0, 30, 60, 90})
      for (int grid : {256, 512})
        run<1, 4>("queue", in, out, B, N, D, grid, 0, delay, 0);
  } else {
    // (b) persistent + de-synchronised start (+ row rotation)
    // delay 60 ~ 12.8 us per tile; stagger unit s => phase k starts k*s*(~4 cycles) later
    for (int rot : {0, 1})
      for (int stag : {0, 500, 2000, 8000})
        for (int delay : {0, 60}) {
          run<1, 0>("copy", in, out, B, N, D, 256, n_tiles / 256, delay, 0, stag, rot);
          run<2, 1>("pipe", in, out, B, N, D, 256, n_tiles / 256, delay, 0, stag, rot);
        }
    run<1, 2>("load", in, out, B, N, D, 256, 48, 0, 0, 2000, 1);
    run<1, 3>("store", in, out, B, N, D, 256, 48, 0, 0, 2000, 1);
  }
  return 0;
}
