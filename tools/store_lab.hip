// store_lab — why does the L2 take 64-byte (half-line) stores at ~3.5 TB/s and 128-byte (full-line) stores at ~5.4 TB/s, and is there a
// form of half-line store it takes faster?  Store-only and load-only passes over a (256 x 4096 rows) x 3072-byte matrix in the product's
// tile shape (a column of W bytes x 4096 rows per workgroup, persistent workgroups, XCD-contiguous order, 16 bytes per lane):
//   full        W = 128: one workgroup writes whole lines
//   pair        W = 64, two neighbouring workgroups (same XCD) write the two halves of the lines, in step (the product kernel's form)
//   self K      W = 128 owned by ONE workgroup that writes the left halves of a 128-row block, then — K blocks later — the right halves
//               (K = 0: back to back): is the price per REQUEST (then self 0 = pair) or per first touch of a line / per cross-CU merge?
//   only-left   W = 64, the right halves are never written (true partial lines all the way to HBM)
//   lane width  the pair form with 8-byte and 4-byte lanes (8 / 16 lanes per 64-byte segment)
// usage: store_lab [reps]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_lab.hip -o tools/store_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int xcd_contiguous(int wg, int n) {
  const int nx = 8;
  const int q = n / nx, rem = n % nx;
  const int xcd = wg % nx, idx = wg / nx;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
}

struct A {
  char* buf; long long row_bytes; int cols128, tiles128, tpw, form, K, load;   // cols128 = 128-byte columns per row; tiles = B * cols128
  float* sink;
  unsigned* cnt;   // one arrival counter per pair (form 6)
};

// form 0 full, 1 pair, 2 self K, 3 only-left, 4 pair 8-byte lanes, 5 pair 4-byte lanes
template <int FORM>
__global__ void __launch_bounds__(512) k_store(const A a) {
  const int tid = threadIdx.x;
  const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int GANG = (FORM == 1 || FORM == 3 || FORM == 4 || FORM == 5 || FORM == 6) ? 2 : 1;
  const int member = wg % GANG, g = wg / GANG;
  [[maybe_unused]] unsigned nsync = 0;
  for (int it = 0; it < a.tpw; ++it) {
    const int t = g * a.tpw + it;                       // 128-byte column tile
    if (t >= a.tiles128) break;
    char* base = a.buf + (long long)(t / a.cols128) * 4096 * a.row_bytes + (long long)(t % a.cols128) * 128;
    if constexpr (FORM == 6) {                          // pair; the two workgroups MEET (device-scope counter) at the top of every tile and every a.K row blocks
      char* p = base + member * 64 + (long long)(tid >> 2) * a.row_bytes + (tid & 3) * 16;
      for (int blk = 0; blk < 32; ++blk) {
        if (blk == 0 || (a.K > 0 && blk % a.K == 0)) {
          __syncthreads();
          if (tid == 0) {
            ++nsync;
            __hip_atomic_fetch_add(a.cnt + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(a.cnt + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2u * nsync && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
          }
          __syncthreads();
        }
        *reinterpret_cast<f4*>(p + (long long)blk * 128 * a.row_bytes) = v;
      }
    } else if constexpr (FORM == 0) {                          // 8 lanes x 16 B per row, 64 rows per instruction
      char* p = base + (long long)(tid >> 3) * a.row_bytes + (tid & 7) * 16;
#pragma unroll 8
      for (int r = 0; r < 4096; r += 64) { if (a.load) acc += *reinterpret_cast<const f4*>(p + (long long)r * a.row_bytes); else *reinterpret_cast<f4*>(p + (long long)r * a.row_bytes) = v; }
    } else if constexpr (FORM == 1 || FORM == 3) {      // 4 lanes x 16 B per row, 128 rows per instruction; member = which half
      if (FORM == 3 && member == 1) continue;
      char* p = base + member * 64 + (long long)(tid >> 2) * a.row_bytes + (tid & 3) * 16;
#pragma unroll 8
      for (int r = 0; r < 4096; r += 128) { if (a.load) acc += *reinterpret_cast<const f4*>(p + (long long)r * a.row_bytes); else *reinterpret_cast<f4*>(p + (long long)r * a.row_bytes) = v; }
    } else if constexpr (FORM == 2) {                   // one workgroup, both halves, the right half K row blocks behind the left
      char* p = base + (long long)(tid >> 2) * a.row_bytes + (tid & 3) * 16;
      const int nb = 4096 / 128;
      for (int blk = 0; blk < nb + a.K; ++blk) {
        if (blk < nb) { if (a.load) acc += *reinterpret_cast<const f4*>(p + (long long)blk * 128 * a.row_bytes); else *reinterpret_cast<f4*>(p + (long long)blk * 128 * a.row_bytes) = v; }
        if (blk >= a.K) { if (a.load) acc += *reinterpret_cast<const f4*>(p + 64 + (long long)(blk - a.K) * 128 * a.row_bytes); else *reinterpret_cast<f4*>(p + 64 + (long long)(blk - a.K) * 128 * a.row_bytes) = v; }
      }
    } else if constexpr (FORM == 4) {                   // 8 lanes x 8 B per 64-byte half, 64 rows per instruction
      char* p = base + member * 64 + (long long)(tid >> 3) * a.row_bytes + (tid & 7) * 8;
      const f2 w = {1.f, 2.f};
#pragma unroll 8
      for (int r = 0; r < 4096; r += 64) { if (a.load) { const f2 q = *reinterpret_cast<const f2*>(p + (long long)r * a.row_bytes); acc.x += q.x; acc.y += q.y; } else *reinterpret_cast<f2*>(p + (long long)r * a.row_bytes) = w; }
    } else {                                            // 16 lanes x 4 B, 32 rows per instruction
      char* p = base + member * 64 + (long long)(tid >> 4) * a.row_bytes + (tid & 15) * 4;
#pragma unroll 8
      for (int r = 0; r < 4096; r += 32) { if (a.load) acc.x += *reinterpret_cast<const float*>(p + (long long)r * a.row_bytes); else *reinterpret_cast<float*>(p + (long long)r * a.row_bytes) = 1.f; }
    }
  }
  if (a.load && acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) *a.sink = acc.x;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const long long B = 256, row_bytes = 3072;
  const size_t bytes = (size_t)B * 4096 * row_bytes;
  char* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  float* sink; CK(hipMalloc(&sink, 4));
  unsigned* cnt; CK(hipMalloc(&cnt, 4096 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto f) {
    for (int i = 0; i < 4; ++i) f();
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
  auto run = [&](const char* name, int form, int K, int per_cu, int load, double frac = 1.0) {
    A a{}; a.buf = buf; a.row_bytes = row_bytes; a.cols128 = (int)(row_bytes / 128); a.tiles128 = (int)B * a.cols128; a.form = form; a.K = K; a.load = load; a.sink = sink; a.cnt = cnt;
    const int gang = (form == 1 || form == 3 || form == 4 || form == 5 || form == 6) ? 2 : 1;
    const int groups = cus * per_cu / gang;
    a.tpw = (a.tiles128 + groups - 1) / groups;
    const int n_wg = gang * ((a.tiles128 + a.tpw - 1) / a.tpw);
    auto go = [&] {
      if (form == 6) CK(hipMemsetAsync(cnt, 0, 4096 * 4, 0));
      switch (form) {
        case 0: hipLaunchKernelGGL(k_store<0>, dim3(n_wg), dim3(512), 0, 0, a); break;
        case 1: hipLaunchKernelGGL(k_store<1>, dim3(n_wg), dim3(512), 0, 0, a); break;
        case 2: hipLaunchKernelGGL(k_store<2>, dim3(n_wg), dim3(512), 0, 0, a); break;
        case 3: hipLaunchKernelGGL(k_store<3>, dim3(n_wg), dim3(512), 0, 0, a); break;
        case 4: hipLaunchKernelGGL(k_store<4>, dim3(n_wg), dim3(512), 0, 0, a); break;
        case 6: hipLaunchKernelGGL(k_store<6>, dim3(n_wg), dim3(512), 0, 0, a); break;
        default: hipLaunchKernelGGL(k_store<5>, dim3(n_wg), dim3(512), 0, 0, a); break;
      }
    };
    const float ms = time(go);
    printf("%-5s %-46s WG/CU %d   %.4f ms  %7.1f GB/s\n", load ? "load" : "store", name, per_cu, ms, frac * bytes / ms / 1e6); fflush(stdout);
  };
  for (int i = 0; i < 30; ++i) { A a{}; a.buf = buf; a.row_bytes = row_bytes; a.cols128 = 24; a.tiles128 = 256 * 24; a.tpw = 24; hipLaunchKernelGGL(k_store<0>, dim3(256), dim3(512), 0, 0, a); }
  CK(hipDeviceSynchronize());
  for (int load : {0, 1})
    for (int per_cu : {1, 2, 4}) {
      run("full lines (128 B per row and workgroup)", 0, 0, per_cu, load);
      run("pair: two workgroups, 64 B each, in step", 1, 0, per_cu, load);
      run("self: one workgroup, left half then right, K=0", 2, 0, per_cu, load);
      run("self, right half 1 block (128 rows) behind", 2, 1, per_cu, load);
      run("self, right half 8 blocks behind", 2, 8, per_cu, load);
      run("self, right half 32 blocks (a whole tile) behind", 2, 32, per_cu, load);
      run("only the left halves (half the bytes)", 3, 0, per_cu, load, 0.5);
      if (!load && per_cu == 1) {
        run("pair, meeting at the top of every tile", 6, 0, per_cu, load);
        run("pair, meeting every 8 row blocks", 6, 8, per_cu, load);
        run("pair, meeting every 2 row blocks", 6, 2, per_cu, load);
        run("pair, meeting every row block", 6, 1, per_cu, load);
      }
      run("pair, 8-byte lanes", 4, 0, per_cu, load);
      run("pair, 4-byte lanes", 5, 0, per_cu, load);
    }
  return 0;
}
