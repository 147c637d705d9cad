// fold_lab.hip — LOAD-ONLY skeleton of a gate-gradient idea (round 6, third session): is the 32-byte row segment of the shipped gate gradient
// (8 channels of V and of dOut per tile: 3.7 TB/s load-only) worth trading for 64-byte segments read TWICE from the L2?
//   pattern A (shipped geometry): a tile = 8 channels x 4096 rows of both tensors (256 KiB), four tiles share a 128-byte line
//   pattern B (folded): a tile = 16 channels x 4096 rows of both tensors (512 KiB), requested by TWO workgroups (the even-bin and the odd-bin
//              half of a radix-2 decimation in frequency would each fold rows n and n + 2048 on arrival and keep 256 KiB), two such pairs
//              share a 128-byte line: every 64-byte segment is requested twice, from HBM once if the pair stays in step
// Both as one tile per workgroup, 512 threads, 16 16-byte requests in flight per lane, LDS-limited to `wgs_per_cu` workgroups per CU; the
// blockIdx -> tile map keeps the workgroups of a line on one XCD (ids congruent mod 8) and adjacent in dispatch order.
//   hipcc --offload-arch=gfx950 -O3 -o tools/fold_lab tools/fold_lab.hip && tools/fold_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// SEG = bytes per row per tensor a workgroup requests (32 or 64); DUP = workgroups that request the same segments (1 or 2)
template <int SEG, int DUP>
__global__ __launch_bounds__(512) void load_only(const char* __restrict__ v, const char* __restrict__ d, float* sink, int B, int N, int row_bytes, int resident, int swap) {
  extern __shared__ char occupancy_limiter[];
  const int per_line = 128 / SEG * DUP;                       // workgroups that touch one 128-byte line
  const int n_wg = gridDim.x;
  // XCD-contiguous order: hardware deals workgroup ids round-robin over 8 XCDs; consecutive `w` share an XCD
  const int w = (blockIdx.x % 8) * (n_wg / 8) + blockIdx.x / 8;
  const int line_slot = w / per_line, member = w % per_line;
  const int seg_idx = member / DUP;                            // which SEG-byte piece of the line
  const int lines_per_row = row_bytes / 128;
  int b = line_slot / lines_per_row, line = line_slot % lines_per_row;
  if (b >= B) return;
  if (resident) { b = 0; line = line_slot % 3; }             // every workgroup on the same 3 MiB: answered by the L2, the CU's own line rate shows
  const long long base = (long long)b * N * row_bytes + (long long)line * 128 + seg_idx * SEG;
  const int lanes_per_row = SEG / 16, rows_per_inst = 512 / lanes_per_row;
  const int r0 = threadIdx.x / lanes_per_row, c16 = (threadIdx.x % lanes_per_row) * 16;
  f4 acc = {0, 0, 0, 0};
  for (int r = r0; r < N; r += rows_per_inst * 8) {
    f4 x[8], y[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int rr = r + k * rows_per_inst;
      if (swap && (member & 1)) rr ^= swap;                    // the partner takes the row groups of `swap` rows in pairwise-swapped order
      const long long off = base + (long long)(rr < N ? rr : r0) * row_bytes + c16;
      x[k] = *reinterpret_cast<const f4*>(v + off);
      y[k] = *reinterpret_cast<const f4*>(d + off);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += x[k] * y[k];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[blockIdx.x] = acc.x;   // never true: keeps the loads
}


// STORE-ONLY: SEG bytes per row of one tensor; solo != 0: only member 0 of every line's gang writes (the other half of the line is never written)
template <int SEG>
__global__ __launch_bounds__(512) void store_only(char* __restrict__ d, int B, int N, int row_bytes, int solo, int swap) {
  extern __shared__ char occupancy_limiter[];
  const int per_line = 128 / SEG;
  const int n_wg = gridDim.x;
  const int w = (blockIdx.x % 8) * (n_wg / 8) + blockIdx.x / 8;
  const int line_slot = w / per_line, member = w % per_line;
  const int lines_per_row = row_bytes / 128;
  const int b = line_slot / lines_per_row, line = line_slot % lines_per_row;
  if (b >= B || (solo && member)) return;
  const long long base = (long long)b * N * row_bytes + (long long)line * 128 + member * SEG;
  const int lanes_per_row = SEG / 16, rows_per_inst = 512 / lanes_per_row;
  const int r0 = threadIdx.x / lanes_per_row, c16 = (threadIdx.x % lanes_per_row) * 16;
  const f4 val = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (int r = r0; r < N; r += rows_per_inst) {
    int rr = r;
    if (swap && (member & 1)) rr ^= swap;
    *reinterpret_cast<f4*>(d + base + (long long)rr * row_bytes + c16) = val;
  }
}

int main(int argc, char** argv) {
  const int B = 256, N = 4096, D = 768, row_bytes = D * 4;
  const size_t bytes = (size_t)B * N * row_bytes;
  char *v, *d; float* sink;
  CK(hipMalloc(&v, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 1 << 20));
  CK(hipMemset(v, 1, bytes)); CK(hipMemset(d, 2, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto kernel, int seg, int dup, int lds_bytes, int resident = 0, int swap = 0) {
    const int n_wg = B * (row_bytes / 128) * (128 / seg) * dup;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    std::vector<float> t;
    for (int it = 0; it < 7; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kernel, dim3(n_wg), dim3(512), lds_bytes, 0, v, d, sink, B, N, row_bytes, resident, swap);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("%-58s swap %4d LDS %3d KiB/wg%s: %.3f ms  = %.2f TB/s of the 6.44 GB both tensors hold\n", name, swap, lds_bytes / 1024, resident ? " L2-resident" : "", t[t.size() / 2], 2.0 * bytes / t[t.size() / 2] * 1e-9);
  };
  for (int lds : {150 * 1024, 72 * 1024, 36 * 1024}) {       // 1, 2, 4 workgroups per CU
    run("A: 32-byte segments, 8 channels, each once (shipped)", load_only<32, 1>, 32, 1, lds);
    run("B: 64-byte segments, 16 channels, each TWICE (folded)", load_only<64, 2>, 64, 2, lds);
    run("C: 64-byte segments, 16 channels, each once (reference)", load_only<64, 1>, 64, 1, lds);
  }
  for (int lds : {150 * 1024, 72 * 1024}) {
    run("A: 32-byte segments (requests / CU limit)", load_only<32, 1>, 32, 1, lds, 1);
    run("C: 64-byte segments", load_only<64, 1>, 64, 1, lds, 1);
    run("D: 128-byte segments (whole lines)", load_only<128, 1>, 128, 1, lds, 1);
  }
  run("D: 128-byte segments (whole lines), from HBM", load_only<128, 1>, 128, 1, 72 * 1024, 0);
  for (int lds : {150 * 1024, 72 * 1024})
    for (int swap : {0, 128, 256, 512, 1024, 2048}) {
      run("C: 64-byte segments, partner's row groups swapped", load_only<64, 1>, 64, 1, lds, 0, swap);
      run("A: 32-byte segments, odd members' row groups swapped", load_only<32, 1>, 32, 1, lds, 0, swap);
    }
  auto runs = [&](const char* name, auto kernel, int seg, int lds_bytes, int solo, int swap) {
    const int n_wg = B * (row_bytes / 128) * (128 / seg);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    std::vector<float> t;
    for (int it = 0; it < 7; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kernel, dim3(n_wg), dim3(512), lds_bytes, 0, d, B, N, row_bytes, solo, swap);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double moved = (double)bytes * (solo ? (double)seg / 128 : 1.0);
    printf("STORE %-46s solo %d swap %4d LDS %3d KiB/wg: %.3f ms  = %.2f TB/s of the %.2f GB written\n", name, solo, swap, lds_bytes / 1024, t[t.size() / 2],
           moved / t[t.size() / 2] * 1e-9, moved * 1e-9);
  };
  for (int lds : {150 * 1024, 72 * 1024}) {
    runs("whole lines", store_only<128>, 128, lds, 0, 0);
    runs("64-byte halves, pairs in step", store_only<64>, 64, lds, 0, 0);
    runs("64-byte halves, partner's row groups swapped", store_only<64>, 64, lds, 0, 512);
    runs("64-byte halves, ONLY one half of every line", store_only<64>, 64, lds, 1, 0);
    runs("32-byte quarters, gangs in step", store_only<32>, 32, lds, 0, 0);
    runs("32-byte quarters, ONLY one quarter of every line", store_only<32>, 32, lds, 1, 0);
  }
  return 0;
}
