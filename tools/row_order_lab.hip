// row_order_lab — does the ORDER in which a workgroup walks the 4096 rows of its tile decide what an "allocation class" costs?
// (profiles/r05_pmc_allocation_classes.txt: a store-only column walk over a typical buffer takes 1.37 x the time of the same walk over a fast
// one, with the same requests and 1.5 x the DRAM-credit stalls: the memory side takes the same stores more slowly.)  Persistent workgroups
// as in the product's static map: workgroup w owns tiles [w * tpw, (w + 1) * tpw) of 64-byte row segments (ST) or whole 128-byte lines
// (the f4 per 8 lanes below covers 128 bytes: seg = 128; seg = 64: 4 lanes per row), row stride 3072 bytes, 4096 rows; stores only.
// Orders of the 64 row blocks (one store instruction of the whole workgroup = one block: 128 rows of 64-byte segments, 64 rows of 128-byte ones):
//   0 sequential   1 bit-reversed   2 four interleaved quarters (consecutive instructions 1024 rows apart)   3 eight interleaved eighths (512 rows apart)
//   4 rows of one instruction 64 rows apart (instruction k = rows k, k + 64, ...: every store of a wave lands in another 192-KiB region)
// usage: row_order_lab [n_buffers = 24]
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/row_order_lab.hip -o tools/row_order_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NI> __device__ __forceinline__ int block_of(int order, int k) {
  constexpr int LG = NI == 64 ? 6 : 5;
  switch (order) {
    case 1: return (int)(__brev((unsigned)k) >> (32 - LG));
    case 2: return (k % 4) * (NI / 4) + k / 4;
    case 3: return (k % 8) * (NI / 8) + k / 8;
    default: return k;
  }
}
template <int ORDER, int SEG>
__global__ void __launch_bounds__(512) st_tiles(char* __restrict__ dst, int n_tiles, int tpw) {
  constexpr int LPR = SEG / 16, RPI = 512 / LPR, NI = 4096 / RPI;       // lanes per row, rows per instruction, instructions per tile
  const int tid = threadIdx.x, wg = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
  const long long row_bytes = 3072;
  for (int it = 0; it < tpw; ++it) {
    const int t = wg * tpw + it;
    if (t >= n_tiles) break;
    constexpr int TPR = 3072 / SEG;
    const long long base = (long long)(t / TPR) * 4096 * row_bytes + (long long)(t % TPR) * SEG + (tid % LPR) * 16;
#pragma unroll 8
    for (int k = 0; k < NI; ++k) {
      long long row;
      if constexpr (ORDER == 4) row = (long long)(tid / LPR) * NI + k;
      else row = (long long)block_of<NI>(ORDER, k) * RPI + tid / LPR;
      *reinterpret_cast<f4*>(dst + base + row * row_bytes) = f4{1.f, 2.f, 3.f, 4.f};
    }
  }
}
template <int ORDER, int SEG> float run(char* buf, size_t bytes, hipEvent_t e0, hipEvent_t e1) {
  const int n_tiles = (int)(bytes / (4096ll * SEG)), tpw = (n_tiles + 255) / 256;
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((st_tiles<ORDER, SEG>), dim3(256), dim3(512), 0, 0, buf, n_tiles, tpw);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  return best;
}
int main(int argc, char** argv) {
  const int want = argc > 1 ? atoi(argv[1]) : 24;
  const size_t bytes = (size_t)256 * 4096 * 3072;
  std::vector<char*> bufs;
  for (int i = 0; i < want; ++i) { char* p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; } bufs.push_back(p); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 40; ++i) run<0, 64>(bufs[0], bytes, e0, e1);
  printf("store-only column walks, ms (best of 3), %zu buffers of %.2f GB; columns: 64-byte segments order 0 1 2 3 4 | 128-byte segments order 0 1 2 3 4\n", bufs.size(), bytes / 1e9);
  std::vector<std::vector<float>> all;
  for (size_t i = 0; i < bufs.size(); ++i) {
    std::vector<float> m = {run<0, 64>(bufs[i], bytes, e0, e1), run<1, 64>(bufs[i], bytes, e0, e1), run<2, 64>(bufs[i], bytes, e0, e1), run<3, 64>(bufs[i], bytes, e0, e1), run<4, 64>(bufs[i], bytes, e0, e1),
                            run<0, 128>(bufs[i], bytes, e0, e1), run<1, 128>(bufs[i], bytes, e0, e1), run<2, 128>(bufs[i], bytes, e0, e1), run<3, 128>(bufs[i], bytes, e0, e1), run<4, 128>(bufs[i], bytes, e0, e1)};
    printf("buffer %2zu:", i); for (size_t j = 0; j < m.size(); ++j) printf(" %s%.3f", j == 5 ? "| " : "", m[j]); printf("\n");
    all.push_back(m);
  }
  printf("median:   "); for (size_t j = 0; j < 10; ++j) { std::vector<float> c; for (auto& m : all) c.push_back(m[j]); std::sort(c.begin(), c.end()); printf(" %s%.3f", j == 5 ? "| " : "", c[c.size() / 2]); } printf("\n");
  printf("fastest:  "); for (size_t j = 0; j < 10; ++j) { float b = 1e9f; for (auto& m : all) b = std::min(b, m[j]); printf(" %s%.3f", j == 5 ? "| " : "", b); } printf("\n");
  printf("slowest:  "); for (size_t j = 0; j < 10; ++j) { float b = 0; for (auto& m : all) b = std::max(b, m[j]); printf(" %s%.3f", j == 5 ? "| " : "", b); } printf("\n");
  return 0;
}
