// iolab — HBM I/O experiments behind the round-2 redesign of the 4096-point register-tile kernel.
//   A  contiguous float4 copy ceiling (grid size x loads in flight per thread x nontemporal)
//   B  the kernel's tile pattern (4096 rows x 64 B at a 3072-B stride, one 512-thread workgroup per CU, 2 wave slots per
//      SIMD, barrier-locked compute delay), 8-byte vs 16-byte accesses per lane, copy / load-only / store-only,
//      with and without a per-workgroup jitter of the compute delay (is the chip in lock-step?)
//   C  persistent register-neutral pipeline: store register group g of tile t, then load group g of tile t+1
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/iolab.hip -o tools/iolab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int xcd_tile(int t, int n) { const int q = n / 8, rem = n % 8, x = t % 8, i = t / 8; return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + i; }

// ---------------- A: contiguous copy -----------------
template <int U, int NT>
__global__ void __launch_bounds__(256) copy4(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(in + i + k * stride) : in[i + k * stride];
#pragma unroll
    for (int k = 0; k < U; ++k) { if (NT) __builtin_nontemporal_store(v[k], out + i + k * stride); else out[i + k * stride] = v[k]; }
  }
  for (; i < n4; i += stride) out[i] = in[i];
}
// block-contiguous variant: each block copies a contiguous chunk (U*4 KiB per step)
template <int U, int NT>
__global__ void __launch_bounds__(256) copy4_chunk(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n4, size_t per_block) {
  const size_t b0 = (size_t)blockIdx.x * per_block, b1 = b0 + per_block < n4 ? b0 + per_block : n4;
  for (size_t i = b0 + threadIdx.x; i + (U - 1) * 256 < b1; i += U * 256) {
    f32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(in + i + k * 256) : in[i + k * 256];
#pragma unroll
    for (int k = 0; k < U; ++k) { if (NT) __builtin_nontemporal_store(v[k], out + i + k * 256); else out[i + k * 256] = v[k]; }
  }
}

// ---------------- B: tile pattern -----------------
// MODE 0 copy, 1 load only, 2 store only.  W = bytes per lane.  One tile per workgroup (non-persistent, XCD-contiguous).
template <int W, int MODE, int NT>
__global__ void __launch_bounds__(512) tile_io(const float* __restrict__ in, float* __restrict__ out, int N, int D, int tpr, int n_tiles,
                                               int delay, int jitter, float fa, float fb, unsigned* sem = nullptr, int sem_k = 0) {
  extern __shared__ char smem[];
  asm volatile("v_mov_b32 v200, 0" ::: "v200");                // 2 wave slots per SIMD, like the FFT kernel
  if (delay < 0) smem[threadIdx.x] = 0;
  if (sem_k > 0) {     // load-admission semaphore per XCD: [0] tickets, [32] waves whose loads have landed
    if (threadIdx.x == 0) {
      unsigned* sm = sem + (blockIdx.x % 8) * 64;
      const unsigned ticket = __hip_atomic_fetch_add(&sm[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while ((int)(ticket * 8u - __hip_atomic_load(&sm[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= sem_k * 8) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
  }
  constexpr int LPR = 64 / W, EPT = 4096 * LPR / 512, NW = W / 4;   // lanes per row, rows per thread, dwords per lane
  const int t = xcd_tile(blockIdx.x, n_tiles);
  const int b = t / tpr, ct = t % tpr;
  const int p = threadIdx.x % LPR, r = threadIdx.x / LPR, RC = 512 / LPR;
  const char* si = reinterpret_cast<const char*>(in + (size_t)b * N * D + ct * 16);
  char* so = reinterpret_cast<char*>(out + (size_t)b * N * D + ct * 16);
  const uint32_t voff = (uint32_t)(r * D * 4 + p * W);
  float v[EPT][NW];
  if (MODE != 2) {
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      const char* ptr = si + (size_t)(q * RC) * D * 4 + voff;
      if (W == 8) { f32x2 x = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(ptr)) : *reinterpret_cast<const f32x2*>(ptr); v[q][0] = x.x; v[q][1] = x.y; }
      else { f32x4 x = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ptr)) : *reinterpret_cast<const f32x4*>(ptr); v[q][0] = x.x; v[q][1] = x.y; v[q][NW - 2] = x.z; v[q][NW - 1] = x.w; }
    }
  } else {
#pragma unroll
    for (int q = 0; q < EPT; ++q)
#pragma unroll
      for (int k = 0; k < NW; ++k) v[q][k] = fa * (q + k) + threadIdx.x;
  }
  if (sem_k > 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&sem[(blockIdx.x % 8) * 64 + 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  int d = delay;
  if (jitter) { uint32_t hsh = (uint32_t)blockIdx.x * 2654435761u; hsh ^= hsh >> 15; d = (int)((long long)delay * (512 + (hsh & 1023)) / 1024); }
  for (int it = 0; it < d; ++it) {
    if ((it % 10) == 0) __syncthreads();
#pragma unroll
    for (int q = 0; q < EPT; ++q)
#pragma unroll
      for (int k = 0; k < NW; ++k) v[q][k] = fmaf(v[q][k], fa, fb);
  }
  if (MODE != 1) {
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
      char* ptr = so + (size_t)(q * RC) * D * 4 + voff;
      if (W == 8) { f32x2 x; x.x = v[q][0]; x.y = v[q][1]; if (NT) __builtin_nontemporal_store(x, reinterpret_cast<f32x2*>(ptr)); else *reinterpret_cast<f32x2*>(ptr) = x; }
      else { f32x4 x; x.x = v[q][0]; x.y = v[q][1]; x.z = v[q][NW - 2]; x.w = v[q][NW - 1]; if (NT) __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(ptr)); else *reinterpret_cast<f32x4*>(ptr) = x; }
    }
  } else {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < EPT; ++q)
#pragma unroll
      for (int k = 0; k < NW; ++k) acc += v[q][k];
    if (acc == 12345.678f) out[threadIdx.x] = acc;
  }
}

// ---------------- C: persistent, register-neutral pipeline -----------------
// one workgroup per CU walks through tpw tiles (pairs of workgroups on adjacent tiles); per tile: wait for the loads, delay,
// then per group of G row-blocks: store the group of tile t, load the same group of tile t+1 into the freed registers.
template <int W, int G>
__global__ void __launch_bounds__(512) tile_pipe(const float* __restrict__ in, float* __restrict__ out, int N, int D, int tpr, int n_tiles,
                                                 int tpw, int n_wg, int delay, float fa, float fb) {
  extern __shared__ char smem[];
  asm volatile("v_mov_b32 v200, 0" ::: "v200");
  if (delay < 0) smem[threadIdx.x] = 0;
  constexpr int LPR = 64 / W, EPT = 4096 * LPR / 512, NW = W / 4;
  const int p = threadIdx.x % LPR, r = threadIdx.x / LPR, RC = 512 / LPR;
  const uint32_t voff = (uint32_t)(r * D * 4 + p * W);
  const int wg_lin = xcd_tile(blockIdx.x, n_wg);
  const int pair_base = (wg_lin >> 1) * tpw * 2 + (wg_lin & 1);
  float v[EPT][NW];
  auto tile_base = [&](int it, const char*& si, char*& so) {
    int tile = pair_base + 2 * it; if (tile >= n_tiles) tile = n_tiles - 1;
    const int b = tile / tpr, ct = tile % tpr;
    si = reinterpret_cast<const char*>(in + (size_t)b * N * D + ct * 16);
    so = reinterpret_cast<char*>(out + (size_t)b * N * D + ct * 16);
  };
  auto ld = [&](const char* si, int q) {
    const char* ptr = si + (size_t)(q * RC) * D * 4 + voff;
    if (W == 8) { f32x2 x = *reinterpret_cast<const f32x2*>(ptr); v[q][0] = x.x; v[q][1] = x.y; }
    else { f32x4 x = *reinterpret_cast<const f32x4*>(ptr); v[q][0] = x.x; v[q][1] = x.y; v[q][NW - 2] = x.z; v[q][NW - 1] = x.w; }
  };
  auto st = [&](char* so, int q) {
    char* ptr = so + (size_t)(q * RC) * D * 4 + voff;
    if (W == 8) { f32x2 x; x.x = v[q][0]; x.y = v[q][1]; *reinterpret_cast<f32x2*>(ptr) = x; }
    else { f32x4 x; x.x = v[q][0]; x.y = v[q][1]; x.z = v[q][NW - 2]; x.w = v[q][NW - 1]; *reinterpret_cast<f32x4*>(ptr) = x; }
  };
  const char* si; char* so;
  tile_base(0, si, so);
#pragma unroll
  for (int q = 0; q < EPT; ++q) ld(si, q);
  for (int it = 0; it < tpw; ++it) {
    for (int k = 0; k < delay; ++k) {
      if ((k % 10) == 0) __syncthreads();
#pragma unroll
      for (int q = 0; q < EPT; ++q)
#pragma unroll
        for (int j = 0; j < NW; ++j) v[q][j] = fmaf(v[q][j], fa, fb);
    }
    const char* sn; char* son;
    tile_base(it + 1, sn, son);
    const bool more = it + 1 < tpw;
#pragma unroll
    for (int g = 0; g < EPT / G; ++g) {
#pragma unroll
      for (int q = g * G; q < (g + 1) * G; ++q) st(so, q);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
#pragma unroll
        for (int q = g * G; q < (g + 1) * G; ++q) ld(sn, q);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    si = sn; so = son;
  }
}


// ---------------- D: persistent, half of the next tile lands in LDS (LDS-DMA, wave-private slots) during the "I2" part of the compute --------
// per tile: read half A from this lane's LDS slots, load half B straight into registers, delay_main (barrier-locked: F1..E2),
// issue the LDS-DMA of the next tile's half A (the exchange image is free again), delay_i2 with the stores issued group by group.
// SPLIT = number of 16-byte row blocks (of 32 per thread) that go through LDS.
template <int SPLIT>
__global__ void __launch_bounds__(512) tile_dma(const float* __restrict__ in, float* __restrict__ out, int N, int D, int tpr, int n_tiles,
                                                int tpw, int n_wg, int delay_main, int delay_i2, float fa, float fb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  asm volatile("v_mov_b32 v200, 0" ::: "v200");
  constexpr int LPR = 4, EPT = 32, RC = 128;
  const int p = threadIdx.x % LPR, r = threadIdx.x / LPR, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t voff = (uint32_t)(r * D * 4 + p * 16);
  const int wg_lin = xcd_tile(blockIdx.x, n_wg);
  const int pair_base = (wg_lin >> 1) * tpw * 2 + (wg_lin & 1);
  char* slot = smem + __builtin_amdgcn_readfirstlane(wave) * (SPLIT * 1024);
  f32x4 v[EPT];
  auto tile_base = [&](int it, const char*& si, char*& so) {
    int tile = pair_base + 2 * it; if (tile >= n_tiles) tile = n_tiles - 1;
    const int b = tile / tpr, ct = tile % tpr;
    si = reinterpret_cast<const char*>(in + (size_t)b * N * D + ct * 16);
    so = reinterpret_cast<char*>(out + (size_t)b * N * D + ct * 16);
  };
  auto dma = [&](const char* si) {
#pragma unroll
    for (int q = 0; q < SPLIT; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(si + (size_t)(q * RC) * D * 4 + voff),
                                       (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
  };
  const char* si; char* so;
  tile_base(0, si, so);
  dma(si);
  for (int it = 0; it < tpw; ++it) {
    // the LDS-DMA of this tile was issued before the previous tile's 32 stores: in-order completion => vmcnt(32) covers it
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
#pragma unroll
    for (int q = 0; q < SPLIT; ++q) v[q] = *reinterpret_cast<const f32x4*>(slot + q * 1024 + lane * 16);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = SPLIT; q < EPT; ++q) v[q] = *reinterpret_cast<const f32x4*>(si + (size_t)(q * RC) * D * 4 + voff);
    for (int k = 0; k < delay_main; ++k) {
      if ((k % 10) == 0) __syncthreads();
#pragma unroll
      for (int q = 0; q < EPT; ++q) v[q] = v[q] * fa + fb;
    }
    const char* sn; char* son;
    tile_base(it + 1, sn, son);
    if (it + 1 < tpw) dma(sn);
    for (int g = 0; g < 8; ++g) {
      for (int k = 0; k < delay_i2 / 8; ++k) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) v[q] = v[q] * fa + fb;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(so + (size_t)((g * 4 + q) * RC) * D * 4 + voff) = v[g * 4 + q];
    }
    si = sn; so = son;
  }
}

static hipEvent_t e0, e1;
template <class F> float timeit(F f, int iters = 5) {
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}

template <int W, int MODE, int NT> void runB(const float* in, float* out, int delay, int jitter, int grid = 12288) {
  const int B = 256, N = 4096, D = 768, tpr = D / 16, n_tiles = B * tpr;
  CK(hipFuncSetAttribute((const void*)tile_io<W, MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const float ms = timeit([&] { tile_io<W, MODE, NT><<<grid, 512, 133 * 1024>>>(in, out, N, D, tpr, n_tiles, delay, jitter, 1.f, 0.f); });
  const double bytes = (MODE == 0 ? 2.0 : 1.0) * grid * N * 64;
  printf("B W=%2d %-5s nt=%d delay=%3d jitter=%d grid=%5d : %7.3f ms  %7.1f GB/s  %6.2f us/tile/CU\n", W, MODE == 0 ? "copy" : MODE == 1 ? "load" : "store", NT, delay, jitter, grid,
         ms, bytes / ms / 1e6, ms * 1e3 / ((grid + 255) / 256));
  fflush(stdout);
}
template <int W> void runS(const float* in, float* out, int delay, int k, unsigned* sem) {
  const int B = 256, N = 4096, D = 768, tpr = D / 16, n_tiles = B * tpr, grid = n_tiles;
  CK(hipFuncSetAttribute((const void*)tile_io<W, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const float ms = timeit([&] { tile_io<W, 0, 0><<<grid, 512, 133 * 1024>>>(in, out, N, D, tpr, n_tiles, delay, 0, 1.f, 0.f, sem, k); });
  printf("S W=%2d copy delay=%3d sem K=%2d : %7.3f ms  %7.1f GB/s  %6.2f us/tile/CU\n", W, delay, k, ms, 2.0 * grid * N * 64 / ms / 1e6, ms * 1e3 / 48);
  fflush(stdout);
}
template <int SPLIT> void runD(const float* in, float* out, int dmain, int di2, int tpw) {
  const int B = 256, N = 4096, D = 768, tpr = D / 16, n_tiles = B * tpr;
  const int n_wg = 2 * ((n_tiles + 2 * tpw - 1) / (2 * tpw));
  CK(hipFuncSetAttribute((const void*)tile_dma<SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const float ms = timeit([&] { tile_dma<SPLIT><<<n_wg, 512, 150 * 1024>>>(in, out, N, D, tpr, n_tiles, tpw, n_wg, dmain, di2, 1.f, 0.f); });
  printf("D split=%2d/32 delay=%3d+%3d tpw=%3d n_wg=%5d : %7.3f ms  %7.1f GB/s  %6.2f us/tile/CU\n", SPLIT, dmain, di2, tpw, n_wg, ms, 2.0 * n_tiles * N * 64 / ms / 1e6, ms * 1e3 * 256 / n_tiles);
  fflush(stdout);
}
template <int W, int G> void runC(const float* in, float* out, int delay, int tpw) {
  const int B = 256, N = 4096, D = 768, tpr = D / 16, n_tiles = B * tpr;
  const int n_wg = 2 * ((n_tiles + 2 * tpw - 1) / (2 * tpw));
  CK(hipFuncSetAttribute((const void*)tile_pipe<W, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const float ms = timeit([&] { tile_pipe<W, G><<<n_wg, 512, 133 * 1024>>>(in, out, N, D, tpr, n_tiles, tpw, n_wg, delay, 1.f, 0.f); });
  printf("C W=%2d G=%2d delay=%3d tpw=%3d n_wg=%5d : %7.3f ms  %7.1f GB/s  %6.2f us/tile/CU\n", W, G, delay, tpw, n_wg, ms, 2.0 * n_tiles * N * 64 / ms / 1e6, ms * 1e3 * 256 / n_tiles);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int B = 256, N = 4096, D = 768; const size_t n = (size_t)B * N * D;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float *in, *out; CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMemset(in, 0x3c, n * 4)); CK(hipMemset(out, 0, n * 4));
  const char* what = argc > 1 ? argv[1] : "ABC";
  auto has = [&](char c) { for (const char* s = what; *s; ++s) if (*s == c) return true; return false; };
  if (has('A')) {
    const size_t n4 = n / 4;
    auto rep = [&](const char* nm, int u, int nt, int grid, float ms) { printf("A %-6s U=%d nt=%d grid=%6d : %7.3f ms  %7.1f GB/s\n", nm, u, nt, grid, ms, 2.0 * n * 4 / ms / 1e6); fflush(stdout); };
    for (int bpc : {4, 8, 16, 32}) {
      const int g = 256 * bpc;
      rep("stride", 1, 0, g, timeit([&] { copy4<1, 0><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4); }));
      rep("stride", 4, 0, g, timeit([&] { copy4<4, 0><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4); }));
      rep("stride", 8, 0, g, timeit([&] { copy4<8, 0><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4); }));
      rep("stride", 4, 1, g, timeit([&] { copy4<4, 1><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4); }));
    }
    for (int g : {65536, 262144}) {
      const size_t pb = (n4 + g - 1) / g;
      rep("chunk", 4, 0, g, timeit([&] { copy4_chunk<4, 0><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4, pb); }));
      rep("chunk", 4, 1, g, timeit([&] { copy4_chunk<4, 1><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4, pb); }));
    }
    {
      const float ms = timeit([&] { CK(hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, 0)); });
      rep("hipMemcpy", 0, 0, 0, ms);
    }
  }
  if (has('E')) {   // copy ceiling, second sweep: few blocks per CU, 1-2 loads in flight per thread, larger blocks
    const size_t n4 = n / 4;
    auto rep = [&](const char* nm, int u, int nt, int grid, int bs, float ms) { printf("E %-6s U=%d nt=%d grid=%6d block=%4d : %7.3f ms  %7.1f GB/s\n", nm, u, nt, grid, bs, ms, 2.0 * n * 4 / ms / 1e6); fflush(stdout); };
    for (int g : {256, 512, 768, 1024, 1280, 1536, 2048, 3072}) {
      rep("stride", 1, 0, g, 256, timeit([&] { copy4<1, 0><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4); }));
      rep("stride", 1, 1, g, 256, timeit([&] { copy4<1, 1><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4); }));
      rep("stride", 2, 0, g, 256, timeit([&] { copy4<2, 0><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4); }));
      rep("stride", 2, 1, g, 256, timeit([&] { copy4<2, 1><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4); }));
    }
    for (int g : {16384, 32768, 65536, 131072}) {
      const size_t pb = (n4 + g - 1) / g;
      rep("chunk", 1, 0, g, 256, timeit([&] { copy4_chunk<1, 0><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4, pb); }));
      rep("chunk", 1, 1, g, 256, timeit([&] { copy4_chunk<1, 1><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4, pb); }));
      rep("chunk", 2, 1, g, 256, timeit([&] { copy4_chunk<2, 1><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4, pb); }));
      rep("chunk", 4, 1, g, 256, timeit([&] { copy4_chunk<4, 1><<<g, 256>>>((const f32x4*)in, (f32x4*)out, n4, pb); }));
    }
  }
  if (has('B')) {
    for (int delay : {0, 86}) {
      runB<8, 0, 0>(in, out, delay, 0); runB<16, 0, 0>(in, out, delay, 0);
      runB<8, 0, 1>(in, out, delay, 0); runB<16, 0, 1>(in, out, delay, 0);
    }
    runB<8, 0, 0>(in, out, 86, 1); runB<16, 0, 0>(in, out, 86, 1);
    runB<8, 1, 0>(in, out, 0, 0); runB<16, 1, 0>(in, out, 0, 0);
    runB<8, 2, 0>(in, out, 0, 0); runB<16, 2, 0>(in, out, 0, 0);
    runB<8, 1, 0>(in, out, 86, 0); runB<16, 1, 0>(in, out, 86, 0);
    runB<8, 2, 0>(in, out, 86, 0); runB<16, 2, 0>(in, out, 86, 0);
    // partial occupancy: per-CU burst rates when only some CUs stream (one generation of workgroups)
    for (int grid : {8, 32, 64, 128, 256}) { runB<8, 1, 0>(in, out, 0, 0, grid); runB<16, 1, 0>(in, out, 0, 0, grid); runB<8, 2, 0>(in, out, 0, 0, grid); runB<16, 2, 0>(in, out, 0, 0, grid); }
  }
  if (has('S')) {
    unsigned* sem; CK(hipMalloc(&sem, 8 * 64 * 4)); CK(hipMemset(sem, 0, 8 * 64 * 4));
    for (int delay : {86, 70, 50}) {
      runB<8, 0, 0>(in, out, delay, 0); runB<16, 0, 0>(in, out, delay, 0);
      for (int k : {6, 8, 10, 12, 14, 16, 20}) { runS<8>(in, out, delay, k, sem); runS<16>(in, out, delay, k, sem); }
    }
  }
  if (has('D')) {
    runB<16, 0, 0>(in, out, 86, 0); runB<16, 0, 0>(in, out, 0, 0);
    for (int tpw : {48, 6}) {
      runD<16>(in, out, 62, 24, tpw); runD<20>(in, out, 62, 24, tpw); runD<8>(in, out, 62, 24, tpw); runD<0>(in, out, 62, 24, tpw);
      runD<16>(in, out, 86, 0, tpw); runD<16>(in, out, 0, 0, tpw); runD<0>(in, out, 0, 0, tpw);
    }
  }
  if (has('C')) {
    for (int delay : {0, 86}) {
      runC<8, 8>(in, out, delay, 48); runC<16, 4>(in, out, delay, 48); runC<8, 64>(in, out, delay, 48);
      runC<8, 8>(in, out, delay, 6); runC<16, 4>(in, out, delay, 6);
    }
  }
  return 0;
}
