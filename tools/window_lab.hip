// window_lab — round 5: WHERE does a tile-shaped persistent kernel lose 17-25 % to the flat copy?
// Copies of the headline tensor in the spectral mix's tile shape (S bytes of 4096 consecutive rows at the 3072-byte row stride,
// S = 128 / 64) with the tile -> workgroup map as a TEMPLATE parameter, so that every form has its own kernel name and rocprofv3's
// counter rows can be told apart (tools/pmc_probe.py runs this binary under --pmc passes):
//   wl_flat                 one 256-thread workgroup per 4 KiB, dispatch order = address order (the guide's 6.3 TB/s form)
//   wl_tile<FAR, S>         static, far apart: gang g walks its own region (the product's order)
//   wl_tile<COMPACT, S>     static, compact: iteration `it` of the whole grid covers tiles [it * n_wg, (it + 1) * n_wg)
//   wl_tile<SKEW, S>        compact + workgroup w starts its row walk at chunk (7 w) mod chunks (neighbours never in the same rows)
//   wl_tile<DYN, S>         dynamic, one ticket per WORKGROUP from one counter (r04: +9.5 / +12.7 % on a slow-class box)
//   wl_tile<DYNG, S>        dynamic, one ticket per GANG (the 128 / S workgroups that share a line): the leader draws, the others read
//                           the ticket from a mailbox — the halves of a line stay with workgroups that walk in step
//   wl_tile<DYNX, S>        as DYNG with one counter per XCD (XCD x owns the x-th eighth of the tensor)
//   wl_pairx<XCH>           (round 6) the TWO-CU form of the 4096-row tile in copy form: a pair of workgroups (neighbours on one
//                           XCD) owns a 128-byte column of 4096 rows = the two 64-byte tiles of the product; member m requests WHOLE 128-byte
//                           lines of rows [2048 m, 2048 m + 2048) and, as the product would have to, hands the half that belongs to its partner's
//                           tile over through an L2-resident scratch ring (XCH >= 1: input side; XCH = 2: the output side as well, so that the
//                           stores are whole lines too).  Traffic emulation, UPPER BOUND: no flags, no waiting for the partner (a real hand-off
//                           needs both), sc1 loads so that the scratch is read from the L2 and not from a stale L1 line; one ticket per pair
// usage: window_lab [reps] [only]     (only = substring of a variant name)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/window_lab.hip -o tools/window_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

enum { FAR = 0, COMPACT = 1, SKEW = 2, DYN = 3, DYNG = 4, DYNX = 5 };

__device__ __forceinline__ int xcd_contiguous(int wg, int n) {
  const int nx = 8;
  const int q = n / nx, rem = n % nx;
  const int xcd = wg % nx, idx = wg / nx;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
}

struct LabArgs {
  const char* src; char* dst;
  long long row_bytes;
  int tile_rows, mode;          // mode 0 copy, 1 load, 2 store
  int n_tiles, cols, tpw, n_wg;
  unsigned* counter;            // [0..7] counters (16 words apart), mailbox from word 256: [gang][slot 0..15]
};

template <int MAP, int SEG, int U = 8, int GX = 0>
__global__ void __launch_bounds__(512) wl_tile(const LabArgs a) {
  constexpr int THREADS = 512, GANG = GX ? GX : 128 / SEG, LPS = SEG / 16, RPI = THREADS / LPS;
  const int tid = threadIdx.x;
  const long long lane_off = (long long)(tid / LPS) * a.row_bytes + (tid % LPS) * 16;
  const long long step = (long long)RPI * a.row_bytes;
  const int chunks = a.tile_rows / (RPI * U);
  const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
  const int member = wg % GANG, g = wg / GANG;
  const int xcd = blockIdx.x % 8;
  __shared__ unsigned next_s;
  unsigned* cnt = a.counter + (MAP == DYNX ? 16 * xcd : 0);
  unsigned* mbox = a.counter + 256 + 16 * g;
  const int per_x = a.n_tiles / GANG / 8;      // gang tickets per XCD (DYNX)
  f4 v[U];
#pragma unroll
  for (int q = 0; q < U; ++q) v[q] = f4{1.f, 2.f, 3.f, 4.f};
  for (int it = 0;; ++it) {
    int t;
    if (MAP == FAR) { if (it >= a.tpw) break; t = g * a.tpw * GANG + member + GANG * it; }
    else if (MAP == COMPACT || MAP == SKEW) { t = it * a.n_wg + wg; }
    else if (MAP == DYN) {
      if (tid == 0) next_s = atomicAdd(cnt, 1u);
      __syncthreads(); t = (int)next_s; __syncthreads();
    } else {                                   // DYNG / DYNX: one ticket per gang
      if (tid == 0) {
        unsigned tk;
        if (member == 0) {
          tk = atomicAdd(cnt, 1u);
          if (GANG > 1) __hip_atomic_store(mbox + (it & 15), ((unsigned)(it + 1) << 16) | tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          tk = 0xffffu;                         // (bounded: a gang whose leader never publishes stops instead of hanging the box)
          for (int sp = 0; sp < (1 << 20); ++sp) {
            const unsigned w = __hip_atomic_load(mbox + (it & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w >> 16) == (unsigned)(it + 1)) { tk = w & 0xffffu; break; }
            __builtin_amdgcn_s_sleep(2);
          }
        }
        next_s = tk;
      }
      __syncthreads(); t = (int)next_s; __syncthreads();
      if (MAP == DYNX) { if (t >= per_x) break; t += xcd * per_x; }
      t = t * GANG + member;
    }
    if (t >= a.n_tiles) break;
    const long long base = (long long)(t / a.cols) * a.tile_rows * a.row_bytes + (long long)(t % a.cols) * SEG;
    const int c0 = MAP == SKEW ? ((7 * g) % chunks) : 0;
    for (int cc = 0; cc < chunks; ++cc) {
      int c = cc + c0; if (c >= chunks) c -= chunks;
      const long long off = base + lane_off + (long long)c * U * step;
      if (a.mode != 2) {
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = *reinterpret_cast<const f4*>(a.src + off + q * step);
      }
      if (a.mode != 1) {
#pragma unroll
        for (int q = 0; q < U; ++q) *reinterpret_cast<f4*>(a.dst + off + q * step) = v[q];
      } else {
#pragma unroll
        for (int q = 0; q < U; ++q) if (v[q].x == 1.2345e-30f) *reinterpret_cast<f4*>(a.dst + off + q * step) = v[q];
      }
    }
  }
}

// XCH: 0 = whole lines of half the rows, no exchange (what the memory side sees: 128-byte tiles of 2048 rows); 1 = + the input-side
// exchange (32 KiB of every 64-KiB chunk out to the scratch, the partner's 32 KiB in); 2 = + the output-side exchange
template <int XCH>
__global__ void __launch_bounds__(512) wl_pairx(const LabArgs a, char* scratch) {
  constexpr int U = 8, THREADS = 512, RPI = THREADS / 8;       // 8 lanes per 128-byte line, 64 rows per instruction, 512 rows per chunk
  const int tid = threadIdx.x;
  const long long lane_off = (long long)(tid / 8) * a.row_bytes + (tid % 8) * 16;
  const long long step = (long long)RPI * a.row_bytes;
  const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
  const int member = wg & 1, g = wg >> 1;
  __shared__ unsigned next_s;
  unsigned* cnt = a.counter;
  unsigned* mbox = a.counter + 256 + 16 * g;
  char* mine_in = scratch + (size_t)wg * 65536, *mine_out = mine_in + 32768;
  const char* theirs_in = scratch + (size_t)(wg ^ 1) * 65536, *theirs_out = theirs_in + 32768;
  f4 v[U];
#pragma unroll
  for (int q = 0; q < U; ++q) v[q] = f4{1.f, 2.f, 3.f, 4.f};
  for (int it = 0;; ++it) {
    if (tid == 0) {
      unsigned tk;
      if (member == 0) {
        tk = atomicAdd(cnt, 1u);
        __hip_atomic_store(mbox + (it & 15), ((unsigned)(it + 1) << 16) | tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        tk = 0xffffu;
        for (int sp = 0; sp < (1 << 20); ++sp) {
          const unsigned w = __hip_atomic_load(mbox + (it & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((w >> 16) == (unsigned)(it + 1)) { tk = w & 0xffffu; break; }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      next_s = tk;
    }
    __syncthreads(); const int t = (int)next_s; __syncthreads();
    if (t >= a.n_tiles) break;                  // n_tiles = 128-byte columns x batch elements
    const long long base = ((long long)(t / a.cols) * a.tile_rows + (long long)member * (a.tile_rows / 2)) * a.row_bytes + (long long)(t % a.cols) * 128;
    const int chunks = a.tile_rows / 2 / (RPI * U);
    for (int c = 0; c < chunks; ++c) {
      const long long off = base + lane_off + (long long)c * U * step;
      if (a.mode != 2) {
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = *reinterpret_cast<const f4*>(a.src + off + q * step);
      }
      if constexpr (XCH >= 1) {                 // half of what was loaded belongs to the partner's tile: out to the scratch, the partner's half in
#pragma unroll
        for (int q = 0; q < U / 2; ++q) *reinterpret_cast<f4*>(mine_in + q * 8192 + tid * 16) = v[q];
#pragma unroll
        for (int q = 0; q < U / 2; ++q) v[q] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(theirs_in + q * 8192 + tid * 16));
      }
      if constexpr (XCH >= 2) {                 // ... and half of the results go back the same way, so that whole lines can be stored
#pragma unroll
        for (int q = 0; q < U / 2; ++q) *reinterpret_cast<f4*>(mine_out + q * 8192 + tid * 16) = v[U / 2 + q];
#pragma unroll
        for (int q = 0; q < U / 2; ++q) v[U / 2 + q] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(theirs_out + q * 8192 + tid * 16));
      }
      if (a.mode != 1) {
#pragma unroll
        for (int q = 0; q < U; ++q) *reinterpret_cast<f4*>(a.dst + off + q * step) = v[q];
      } else {
#pragma unroll
        for (int q = 0; q < U; ++q) if (v[q].x == 1.2345e-30f) *reinterpret_cast<f4*>(a.dst + off + q * step) = v[q];
      }
    }
  }
}

// wl_mixed<OVF>  (round 6) the I/O of the bf16-rows-in / fp32-rows-out kernel in copy form: a tile = 16 channels x 4096 rows = 32 bytes per row of
// the bf16 source (row stride 1536) and 64 bytes per row of the fp32 destination (row stride 3072).  OVF = 0: what the product does — one ticket
// per QUAD of workgroups, 32-byte load segments.  OVF = 1: one ticket per PAIR, every workgroup requests the whole 64-byte segment its 32 bytes
// lie in (its partner requests the same segment: an L2 hit) — half of what arrives is not used.
template <int OVF>
__global__ void __launch_bounds__(512) wl_mixed(const LabArgs a) {
  constexpr int GANG = OVF ? 2 : 4, LSEG = OVF ? 64 : 32, LLPS = LSEG / 16, LRPI = 512 / LLPS, SRPI = 512 / 4, U = 8;
  const int tid = threadIdx.x;
  const long long src_row = a.row_bytes / 2, dst_row = a.row_bytes;
  const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
  const int member = wg % GANG, g = wg / GANG;
  __shared__ unsigned next_s;
  unsigned* cnt = a.counter;
  unsigned* mbox = a.counter + 256 + 16 * g;
  f4 v[U];
#pragma unroll
  for (int q = 0; q < U; ++q) v[q] = f4{1.f, 2.f, 3.f, 4.f};
  for (int it = 0;; ++it) {
    if (tid == 0) {
      unsigned tk;
      if (member == 0) {
        tk = atomicAdd(cnt, 1u);
        __hip_atomic_store(mbox + (it & 15), ((unsigned)(it + 1) << 16) | tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        tk = 0xffffu;
        for (int sp = 0; sp < (1 << 20); ++sp) {
          const unsigned w = __hip_atomic_load(mbox + (it & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((w >> 16) == (unsigned)(it + 1)) { tk = w & 0xffffu; break; }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      next_s = tk;
    }
    __syncthreads(); int t = (int)next_s; __syncthreads();
    t = t * GANG + member;
    if (t >= a.n_tiles) break;                  // tiles = 16-channel columns x batch elements (cols = 48)
    const long long b = t / a.cols, col = t % a.cols;
    const long long sbase = b * a.tile_rows * src_row + (OVF ? (col / 2) * 64 : col * 32);
    const long long dbase = b * a.tile_rows * dst_row + col * 64;
    const long long l_off = (long long)(tid / LLPS) * src_row + (tid % LLPS) * 16, s_off = (long long)(tid / 4) * dst_row + (tid % 4) * 16;
    for (int c = 0; c < a.tile_rows / (SRPI * U); ++c) {           // 1024 rows per chunk
      constexpr int LI = SRPI * U / LRPI;                          // load instructions per chunk: 4 (32-byte segments) or 8 (64-byte)
      if (a.mode != 2) {
#pragma unroll
        for (int q = 0; q < LI; ++q) v[q] = *reinterpret_cast<const f4*>(a.src + sbase + l_off + ((long long)c * LI + q) * LRPI * src_row);
      }
      if (a.mode == 0) {
#pragma unroll
        for (int q = 0; q < U; ++q) *reinterpret_cast<f4*>(a.dst + dbase + s_off + ((long long)c * U + q) * SRPI * dst_row) = v[q];
      } else if (a.mode == 1) {
#pragma unroll
        for (int q = 0; q < LI; ++q) if (v[q].x == 1.2345e-30f) *reinterpret_cast<f4*>(a.dst + dbase + s_off + q * SRPI * dst_row) = v[q];
      } else {
#pragma unroll
        for (int q = 0; q < U; ++q) *reinterpret_cast<f4*>(a.dst + dbase + s_off + ((long long)c * U + q) * SRPI * dst_row) = v[q];
      }
    }
  }
}

__global__ void __launch_bounds__(256) wl_flat(const f4* __restrict__ src, f4* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  dst[i] = src[i];
}
__global__ void __launch_bounds__(256) wl_flat_load(const f4* __restrict__ src, f4* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const f4 v = src[i];
  if (v.x == 1.2345e-30f) dst[i] = v;
}
__global__ void __launch_bounds__(256) wl_flat_store(f4* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  dst[i] = f4{1.f, 2.f, 3.f, 4.f};
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const char* only_arg = argc > 2 ? argv[2] : "";                       // substrings separated by '|': a variant runs if its name contains one of them
  std::vector<std::string> only_list;
  { std::string t = only_arg; size_t p0 = 0; for (;;) { const size_t p1 = t.find('|', p0); only_list.push_back(t.substr(p0, p1 == std::string::npos ? p1 : p1 - p0)); if (p1 == std::string::npos) break; p0 = p1 + 1; } }
  auto wanted = [&](const char* name) {                                 // ... every '&'-separated part of an alternative must occur
    for (auto& o : only_list) {
      bool all = true; size_t p0 = 0;
      for (;;) { const size_t p1 = o.find('&', p0); if (!strstr(name, o.substr(p0, p1 == std::string::npos ? p1 : p1 - p0).c_str())) all = false; if (p1 == std::string::npos) break; p0 = p1 + 1; }
      if (all) return true;
    }
    return false; };
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const long long B = 256, N = 4096, D = 768, row_bytes = D * 4, rows = B * N;
  const size_t bytes = (size_t)rows * row_bytes;
  char *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  unsigned* counter; CK(hipMalloc(&counter, 65536)); CK(hipMemset(counter, 0, 65536));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto f) {
    for (int i = 0; i < (reps > 2 ? 4 : 1); ++i) f();
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
  const dim3 fg((unsigned)(bytes / 4096));
  for (int i = 0; i < (reps > 2 ? 40 : 1); ++i) hipLaunchKernelGGL(wl_flat, fg, dim3(256), 0, 0, (const f4*)a, (f4*)b);
  CK(hipDeviceSynchronize());
  auto report = [&](const char* name, float ms, double nb) { printf("%-44s %.4f ms  %7.1f GB/s\n", name, ms, nb / ms / 1e6); fflush(stdout); };
  if (wanted("flat copy")) report("flat copy", time([&] { hipLaunchKernelGGL(wl_flat, fg, dim3(256), 0, 0, (const f4*)a, (f4*)b); }), 2.0 * bytes);
  if (wanted("flat load")) report("flat load", time([&] { hipLaunchKernelGGL(wl_flat_load, fg, dim3(256), 0, 0, (const f4*)a, (f4*)b); }), 1.0 * bytes);
  if (wanted("flat store")) report("flat store", time([&] { hipLaunchKernelGGL(wl_flat_store, fg, dim3(256), 0, 0, (f4*)b); }), 1.0 * bytes);

  auto run = [&](auto kern, const char* name, int seg, bool dyn, int mode, int gx = 0) {
    char full[96];
    static const char* moden[3] = {"copy", "load", "store"};
    snprintf(full, sizeof full, "seg %3d %-26s %s", seg, name, moden[mode]);
    if (!wanted(full)) return;
    LabArgs x{};
    x.src = a; x.dst = b; x.row_bytes = row_bytes; x.tile_rows = 4096; x.mode = mode;
    const int gang = gx ? gx : 128 / seg;
    x.cols = (int)(row_bytes / seg); x.n_tiles = (int)(rows / 4096) * x.cols;
    const int slots = cus / gang * gang;
    x.tpw = (x.n_tiles + slots - 1) / slots;
    x.n_wg = gang * ((x.n_tiles + gang * x.tpw - 1) / (gang * x.tpw));
    x.counter = counter;
    auto go = [&] {
      if (dyn) CK(hipMemsetAsync(counter, 0, 65536, 0));
      hipLaunchKernelGGL(kern, dim3(x.n_wg), dim3(512), 0, 0, x);
    };
    report(full, time(go), (mode == 0 ? 2.0 : 1.0) * bytes);
  };
  char* scratch; CK(hipMalloc(&scratch, (size_t)1024 * 65536)); CK(hipMemset(scratch, 0, (size_t)1024 * 65536));
  auto runx = [&](auto kern, const char* name, int mode) {
    char full[96];
    static const char* moden[3] = {"copy", "load", "store"};
    snprintf(full, sizeof full, "two-CU  %-26s %s", name, moden[mode]);
    if (!wanted(full)) return;
    LabArgs x{};
    x.src = a; x.dst = b; x.row_bytes = row_bytes; x.tile_rows = 4096; x.mode = mode;
    x.cols = (int)(row_bytes / 128); x.n_tiles = (int)(rows / 4096) * x.cols;      // tickets = 128-byte columns
    x.n_wg = cus / 2 * 2; x.tpw = 0; x.counter = counter;
    auto go = [&] { CK(hipMemsetAsync(counter, 0, 65536, 0)); hipLaunchKernelGGL(kern, dim3(x.n_wg), dim3(512), 0, 0, x, scratch); };
    report(full, time(go), (mode == 0 ? 2.0 : 1.0) * bytes);
  };
  for (int mode : {0, 1, 2}) {
    runx(wl_pairx<0>, "whole lines, half the rows", mode);
    runx(wl_pairx<1>, "+ input-side exchange", mode);
    runx(wl_pairx<2>, "+ both exchanges", mode);
  }
  // round 6: how many requests should a lane keep in flight?  (the masked C4 probe of rounds 4-5 had ONE per wave — hipcc had serialised its
  // predicated loads — and was 25 % FASTER than the same copy with sixteen)
  // round 6: bf16 rows in / fp32 rows out — 32-byte load segments in quads (the product) against over-fetched 64-byte segments in pairs
  auto runm = [&](auto kern, const char* name, int gang, int mode) {
    char full[96];
    static const char* moden[3] = {"copy", "load", "store"};
    snprintf(full, sizeof full, "bf16->f32 %-30s %s", name, moden[mode]);
    if (!wanted(full)) return;
    LabArgs x{};
    x.src = a; x.dst = b; x.row_bytes = row_bytes; x.tile_rows = 4096; x.mode = mode;
    x.cols = (int)(row_bytes / 64); x.n_tiles = (int)(rows / 4096) * x.cols;
    x.n_wg = cus / gang * gang; x.counter = counter;
    auto go = [&] { CK(hipMemsetAsync(counter, 0, 65536, 0)); hipLaunchKernelGGL(kern, dim3(x.n_wg), dim3(512), 0, 0, x); };
    report(full, time(go), (mode == 0 ? 1.5 : mode == 1 ? 0.5 : 1.0) * bytes);
  };
  for (int mode : {0, 1, 2}) {
    runm(wl_mixed<0>, "32-byte loads, quads", 4, mode);
    runm(wl_mixed<1>, "64-byte over-fetch, pairs", 2, mode);
  }
  // round 6: one ticket per FOUR / EIGHT workgroups = 2 / 4 adjacent lines of a row (64-byte tiles)
  for (int mode : {0, 1, 2}) {
    run(wl_tile<DYNG, 64, 8, 4>, "ticket per 4 wgs (2 lines)", 64, true, mode, 4);
    run(wl_tile<DYNG, 64, 8, 8>, "ticket per 8 wgs (4 lines)", 64, true, mode, 8);
    run(wl_tile<DYNG, 64, 4, 4>, "ticket per 4 wgs, 4 in flight", 64, true, mode, 4);
  }
  for (int mode : {0, 1, 2}) {
    run(wl_tile<FAR, 64, 1>, "static far, 1 in flight", 64, false, mode);
    run(wl_tile<FAR, 64, 2>, "static far, 2 in flight", 64, false, mode);
    run(wl_tile<FAR, 64, 4>, "static far, 4 in flight", 64, false, mode);
    run(wl_tile<FAR, 64, 16>, "static far, 16 in flight", 64, false, mode);
    run(wl_tile<DYNG, 64, 1>, "per PAIR, 1 in flight", 64, true, mode);
    run(wl_tile<DYNG, 64, 2>, "per PAIR, 2 in flight", 64, true, mode);
    run(wl_tile<DYNG, 64, 4>, "per PAIR, 4 in flight", 64, true, mode);
    run(wl_tile<DYNG, 64, 16>, "per PAIR, 16 in flight", 64, true, mode);
  }
  for (int mode : {0, 1, 2}) {
    run(wl_tile<FAR, 128>, "static far", 128, false, mode);
    run(wl_tile<COMPACT, 128>, "static compact", 128, false, mode);
    run(wl_tile<SKEW, 128>, "static compact skewed", 128, false, mode);
    run(wl_tile<DYN, 128>, "dynamic per workgroup", 128, true, mode);
    run(wl_tile<DYNX, 128>, "dynamic per wg, per-XCD ctr", 128, true, mode);
    run(wl_tile<FAR, 64>, "static far", 64, false, mode);
    run(wl_tile<COMPACT, 64>, "static compact", 64, false, mode);
    run(wl_tile<SKEW, 64>, "static compact skewed", 64, false, mode);
    run(wl_tile<DYN, 64>, "dynamic per workgroup", 64, true, mode);
    run(wl_tile<DYNG, 64>, "dynamic per PAIR", 64, true, mode);
    run(wl_tile<DYNX, 64>, "dynamic per PAIR, per-XCD ctr", 64, true, mode);
    run(wl_tile<FAR, 32>, "static far", 32, false, mode);
    run(wl_tile<DYN, 32>, "dynamic per workgroup", 32, true, mode);
    run(wl_tile<DYNG, 32>, "dynamic per QUAD", 32, true, mode);
    run(wl_tile<DYNX, 32>, "dynamic per QUAD, per-XCD ctr", 32, true, mode);
  }
  return 0;
}
