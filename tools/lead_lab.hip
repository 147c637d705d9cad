// lead_lab.hip — COPY-FORM test of one idea (round 6, third session): the two workgroups of a pair ask for the two halves of the same 128-byte lines
// at the same time, so the L2's miss path sees TWO requests per fetched line (5.3 TB/s load-only against 6.15 for one, tools/fold_lab.hip).  If the
// LEADER's loads ran `lead` row blocks further ahead than the follower's — both still STORING the same rows at the same time, which the L2 needs to
// merge the halves — the follower's requests would find their lines already there (true hits) and the miss path would see one request per line.
//   persistent workgroups (one or two per CU), each walks through `tpw` half-line tiles (64 bytes x 4096 rows) of its pair's column of tiles;
//   loads run `dist` blocks of 128 rows ahead of the stores through a register ring (across tile boundaries); leader: dist + lead, follower: dist.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lead_lab tools/lead_lab.hip && tools/lead_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 4096, kBlockRows = 128, kBlocksPerTile = kRows / kBlockRows;   // 512 threads = 128 rows x 4 lanes of 16 bytes

// RING = register blocks per thread (max distance); the workgroup's blocks form one sequence over its tiles
template <int RING>
__global__ __launch_bounds__(512) void pair_copy(const char* __restrict__ src, char* __restrict__ dst, int n_tiles, int tiles_per_row, int row_bytes,
                                                 int tpw, int dist, int lead) {
  extern __shared__ char occupancy_limiter[];
  const int n_wg = gridDim.x;
  const int w = (blockIdx.x % 8) * (n_wg / 8) + blockIdx.x / 8;          // XCD-contiguous: w and w ^ 1 share an XCD and are dispatch neighbours
  const int pair = w >> 1, member = w & 1;
  const int my_dist = dist + (member == 0 ? lead : 0);
  const int first_tile = pair * tpw * 2 + member;                           // tiles first_tile, +2, +4, ... (the partner takes the other halves)
  const int r0 = threadIdx.x >> 2, c16 = (threadIdx.x & 3) * 16;
  const int total_blocks = tpw * kBlocksPerTile;
  auto addr = [&](int blk) -> long long {
    const int t = first_tile + 2 * (blk / kBlocksPerTile), rb = blk % kBlocksPerTile;
    const int b = t / tiles_per_row, ct = t % tiles_per_row;
    return ((long long)b * kRows + rb * kBlockRows + r0) * row_bytes + ct * 64 + c16;
  };
  if (first_tile + 2 * (tpw - 1) >= n_tiles) return;
  f4 ring[RING];
  // prologue: my_dist blocks in flight
#pragma unroll
  for (int k = 0; k < RING; ++k)
    if (k < my_dist && k < total_blocks) ring[k] = *reinterpret_cast<const f4*>(src + addr(k));
  for (int base = 0; base < total_blocks; base += RING) {
#pragma unroll
    for (int k = 0; k < RING; ++k) {
      const int blk = base + k;
      if (blk < total_blocks) {
        *reinterpret_cast<f4*>(dst + addr(blk)) = ring[k];
        // (ring slot k = blocks k, k + RING, ...; the load that refills it is the block my_dist ahead of the one just stored only when my_dist == RING:
        //  for smaller distances the slot of block blk + my_dist is (k + my_dist) % RING — handled by issuing that load here)
      }
      const int nxt = blk + my_dist;
      if (nxt < total_blocks && nxt >= my_dist) {
        const int slot = (k + my_dist) % RING;
        // static slot indexing needs compile-time indices: my_dist is uniform, so select by switch over the RING possibilities
#pragma unroll
        for (int sidx = 0; sidx < RING; ++sidx)
          if (sidx == slot) ring[sidx] = *reinterpret_cast<const f4*>(src + addr(nxt));
      }
    }
  }
}

int main() {
  const int B = 256, D = 768, row_bytes = D * 4, tiles_per_row = row_bytes / 64, n_tiles = B * tiles_per_row;
  const size_t bytes = (size_t)B * kRows * row_bytes;
  char *src, *dst;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
  CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int wgs_per_cu, int dist, int lead) {
    const int n_wg = 256 * wgs_per_cu, tpw = n_tiles / n_wg;
    const int lds = wgs_per_cu == 1 ? 150 * 1024 : 72 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_copy<16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    std::vector<float> t;
    for (int it = 0; it < 7; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(pair_copy<16>, dim3(n_wg), dim3(512), lds, 0, src, dst, n_tiles, tiles_per_row, row_bytes, tpw, dist, lead);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(t.begin(), t.end());
    printf("%d workgroup(s) per CU, follower %2d blocks ahead, leader +%2d (= %4d rows): %.3f ms = %.2f TB/s\n", wgs_per_cu, dist, lead, lead * kBlockRows,
           t[t.size() / 2], 2.0 * bytes / t[t.size() / 2] * 1e-9);
  };
  for (int wpc : {1, 2})
    for (int dist : {4, 8})
      for (int lead : {0, 2, 4, 8}) if (dist + lead <= 16) run(wpc, dist, lead);
  // correctness of the ring: dst must equal src
  std::vector<char> h(1 << 20);
  CK(hipMemcpy(h.data(), dst + (bytes / 2), h.size(), hipMemcpyDeviceToHost));
  bool ok = true; for (char c : h) ok = ok && c == 1;
  printf("copy check: %s\n", ok ? "ok" : "MISMATCH");
  return 0;
}
