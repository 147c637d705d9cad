// copy_probe.hip — pure-copy probes behind the C ABI (spectre_probe_copy): what does the memory system of THIS box deliver to a kernel
// that moves the spectral mix's bytes and does nothing else?  bench.py runs them next to the product kernel so that the ceiling the
// product is compared with is measured in the same process, on the same device, by the driver (roofline.dense_copy_GBps /
// roofline.pattern_copy_GBps).  No arithmetic; measurement only — nothing in the product path calls these kernels.
//
//   seg_bytes = 0   dense: persistent workgroups stream 256-KiB chunks of the flat buffer (16 bytes per lane, coalesced)
//   seg_bytes = S   the product's access pattern: the buffer is a (rows x row_bytes) matrix (row_bytes = D * element size); a tile is
//                   S bytes of `tile_rows` consecutive rows (S = 64: the 16 fp32 channels x 4096 rows one workgroup of the 4096 kernel
//                   owns; 32 = its bf16 rows; 128 = a whole L2 line per row).  Tiles that share 128-byte lines are neighbours in the
//                   XCD-contiguous order and are walked in step by 128 / S neighbouring workgroups, exactly like the product kernels.
//   mode 0 copy, 1 load only (the values are consumed by a never-true store), 2 store only
// Round 4 ("a ceiling that agrees with the hardware guide"): the persistent forms above top out at 5.0-5.8 TB/s
// where MI355X_MICROARCH.md records 6.29 TB/s for "a float4 copy".  tools/copy_ceiling.hip found the form that reaches it on these
// boxes — the plain NON-persistent copy, one 256-thread workgroup per 4 KiB, one 16-byte load and one 16-byte store per lane —
// and bench.py now measures it beside the product (copy mode only):
//   seg_bytes = -1  flat float4 copy (6.25-6.31 TB/s on 3 GiB -> 3 GiB)
//   seg_bytes = -2  the same with non-temporal loads and stores (6.56-6.57 TB/s)
//   seg_bytes = -3  hipMemcpyAsync device-to-device (4.9-5.2 TB/s)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernel_regtile.h"
#include "../../include/spectre_hip.h"

namespace sfft {

typedef float probe_f32x4 __attribute__((ext_vector_type(4)));

struct ProbeArgs {
  const char* src; char* dst;
  long long rows, row_bytes;
  int seg, tile_rows, mode;
  int n_tiles, tpw, gang;       // tiles in total, tiles per workgroup, workgroups per 128-byte line
  int cols;                     // tiles per row block (row_bytes / seg)
};

constexpr int kProbeThreads = 512, kProbeInFlight = 16;   // 16 x 16 bytes per lane in flight = 128 KiB per workgroup

// RAGGED = a tile is not a whole number of 128-KiB chunks (C4: 64 bytes x 3000 rows): the last chunk's accesses beyond the tile's rows are
// masked off per lane.  The whole-chunk shapes keep the unpredicated instantiation (their numbers are compared across rounds).
template <bool RAGGED>
__global__ void __launch_bounds__(kProbeThreads) spectre_probe_copy_kernel(const ProbeArgs a) {
  const int tid = threadIdx.x;
  const int wg = xcd_contiguous(blockIdx.x, gridDim.x);
  const int base_tile = (wg / a.gang) * a.tpw * a.gang + (wg % a.gang);
  probe_f32x4 v[kProbeInFlight];
#pragma unroll
  for (int q = 0; q < kProbeInFlight; ++q) v[q] = probe_f32x4{1.f, 2.f, 3.f, 4.f};
  // lane -> (row inside an instruction's row block, 16-byte piece of the segment)
  const int lps = a.seg > 0 ? a.seg / 16 : 0;                         // lanes per segment
  const long long lane_off = a.seg > 0 ? (long long)(tid / lps) * a.row_bytes + (tid % lps) * 16 : (long long)tid * 16;
  const long long step = a.seg > 0 ? (long long)(kProbeThreads / lps) * a.row_bytes : (long long)kProbeThreads * 16;   // bytes between instructions
  const long long tile_bytes = a.seg > 0 ? (long long)a.seg * a.tile_rows : 256 * 1024;
  const int chunks = (int)((tile_bytes + kProbeThreads * 16 * kProbeInFlight - 1) / (kProbeThreads * 16 * kProbeInFlight));   // 128-KiB chunks per tile
  [[maybe_unused]] const int rpi = lps > 0 ? kProbeThreads / lps : 1, row0 = lps > 0 ? tid / lps : 0;                    // rows per instruction, this lane's row in it
  for (int it = 0; it < a.tpw; ++it) {
    const int t = base_tile + a.gang * it;
    if (t >= a.n_tiles) break;
    const long long base = a.seg > 0 ? (long long)(t / a.cols) * a.tile_rows * a.row_bytes + (long long)(t % a.cols) * a.seg : (long long)t * tile_bytes;
    for (int c = 0; c < chunks; ++c) {
      const long long off = base + lane_off + (long long)c * kProbeInFlight * step;
      if constexpr (RAGGED) {
        // The product's own mechanism (kernel_regtile64p.h): raw buffer resources whose range ends with the tile's last row — a request
        // beyond it costs no memory traffic (loads return 0, stores are dropped) and EVERY request is issued unconditionally.  Written as
        // `if (row < tile_rows) v[q] = load`, hipcc gives each load a branch of its own with an `s_waitcnt vmcnt(0)` in front — one
        // request in flight per wave: that is what rounds 4-5 measured as C4's "pattern copy" (found by fft_amd/isa_lint.py, round 6).
        typedef unsigned int probe_u32x4 __attribute__((ext_vector_type(4)));
        const int extent = (int)((long long)a.tile_rows * a.row_bytes);        // (bytes from the tile's first row; < 2^31 for every shape bench.py uses)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.src) + base, 0, extent, kRsrcFlags);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(a.dst + base, 0, extent, kRsrcFlags);
        const uint32_t o32 = (uint32_t)(lane_off + (long long)c * kProbeInFlight * step);
        if (a.mode != 2) {
#pragma unroll
          for (int q = 0; q < kProbeInFlight; ++q) {
            const probe_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, o32 + (uint32_t)(q * step), 0, 0);
            v[q] = probe_f32x4{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
          }
        }
        if (a.mode != 1) {
#pragma unroll
          for (int q = 0; q < kProbeInFlight; ++q) {
            probe_u32x4 t; t.x = __float_as_uint(v[q].x); t.y = __float_as_uint(v[q].y); t.z = __float_as_uint(v[q].z); t.w = __float_as_uint(v[q].w);
            __builtin_amdgcn_raw_buffer_store_b128(t, rd, o32 + (uint32_t)(q * step), 0, 0);
          }
        } else {
#pragma unroll
          for (int q = 0; q < kProbeInFlight; ++q)
            if (v[q].x == 1.2345e-30f) *reinterpret_cast<probe_f32x4*>(a.dst + off + q * step) = v[q];
        }
        continue;
      }
      if (a.mode != 2) {
#pragma unroll
        for (int q = 0; q < kProbeInFlight; ++q) v[q] = *reinterpret_cast<const probe_f32x4*>(a.src + off + q * step);
      }
      if (a.mode != 1) {
#pragma unroll
        for (int q = 0; q < kProbeInFlight; ++q) *reinterpret_cast<probe_f32x4*>(a.dst + off + q * step) = v[q];
      } else {
#pragma unroll
        for (int q = 0; q < kProbeInFlight; ++q)
          if (v[q].x == 1.2345e-30f) *reinterpret_cast<probe_f32x4*>(a.dst + off + q * step) = v[q];   // keeps the loads alive, never true
      }
    }
  }
}

template <bool NT>
__global__ void __launch_bounds__(256) spectre_probe_flat_kernel(const probe_f32x4* __restrict__ src, probe_f32x4* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if constexpr (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
  else dst[i] = src[i];
}

// returns a SPECTRE_E_* code; *why = a static message on failure (the extern "C" wrapper in spectre_hip.hip stores it for spectre_last_error)
int probe_copy(const SpectreProbeArgs* p, int warmup, int iters, float* ms_per_launch, const char** why) {
  *why = "";
  if (!p || !ms_per_launch || iters < 1 || warmup < 0 || !p->src || !p->dst || p->rows < 1 || p->row_bytes < 16) { *why = "NULL pointer or bad size"; return SPECTRE_E_INVALID; }
  const int seg = p->seg_bytes;
  if (seg < 0) {          // the flat (non-persistent) copies and the runtime's own device-to-device copy
    const long long bytes = p->rows * p->row_bytes;
    if (seg < -3 || p->mode != 0 || bytes % 4096 || bytes / 4096 > 0x7fffffffLL) { *why = "seg_bytes -1 / -2 / -3: copy mode only, whole 4-KiB chunks"; return SPECTRE_E_INVALID; }
    *why = "HIP runtime call failed";
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(p->device) != hipSuccess) return SPECTRE_E_HIP;
    hipStream_t stream = reinterpret_cast<hipStream_t>(p->stream);
    const probe_f32x4* s = static_cast<const probe_f32x4*>(p->src);
    probe_f32x4* d = static_cast<probe_f32x4*>(p->dst);
    const dim3 grid((unsigned)(bytes / 4096));
    auto go = [&]() -> hipError_t {
      if (seg == -1) hipLaunchKernelGGL(spectre_probe_flat_kernel<false>, grid, dim3(256), 0, stream, s, d);
      else if (seg == -2) hipLaunchKernelGGL(spectre_probe_flat_kernel<true>, grid, dim3(256), 0, stream, s, d);
      else return hipMemcpyAsync(p->dst, p->src, (size_t)bytes, hipMemcpyDeviceToDevice, stream);
      return hipSuccess;
    };
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) { (void)hipSetDevice(prev); return SPECTRE_E_HIP; }
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); (void)hipSetDevice(prev); return SPECTRE_E_HIP; }
    hipError_t e = hipSuccess;
    for (int i = 0; i < warmup && e == hipSuccess; ++i) e = go();
    if (e == hipSuccess) e = hipEventRecord(e0, stream);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = go();
    if (e == hipSuccess) e = hipEventRecord(e1, stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess) e = hipGetLastError();
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return SPECTRE_E_HIP;
    *ms_per_launch = ms / (float)iters;
    *why = "";
    return SPECTRE_OK;
  }
  if (seg != 0 && (seg < 16 || (seg & (seg - 1)) || seg > 1024 || p->row_bytes % seg || p->tile_rows < 1 || p->rows % p->tile_rows)) {
    *why = "seg_bytes must be 0 or a power of two in 16..1024 that divides row_bytes; tile_rows must divide rows"; return SPECTRE_E_INVALID; }
  const bool ragged = seg != 0 && ((long long)seg * p->tile_rows) % (kProbeThreads * 16 * kProbeInFlight) != 0;   // (C4: 3000 rows; masked last chunk)
  if (seg == 0 && (p->rows * p->row_bytes) % (256 * 1024)) { *why = "dense copy: the buffer must be whole 256-KiB chunks"; return SPECTRE_E_INVALID; }
  if (p->mode < 0 || p->mode > 2) { *why = "mode must be 0 (copy), 1 (load) or 2 (store)"; return SPECTRE_E_INVALID; }
  *why = "HIP runtime call failed";
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(p->device) != hipSuccess) return SPECTRE_E_HIP;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, p->device) != hipSuccess) { (void)hipSetDevice(prev); return SPECTRE_E_HIP; }
  ProbeArgs a{};
  a.src = static_cast<const char*>(p->src); a.dst = static_cast<char*>(p->dst);
  a.rows = p->rows; a.row_bytes = p->row_bytes; a.seg = seg; a.tile_rows = p->tile_rows; a.mode = p->mode;
  a.gang = seg > 0 && seg < 128 ? 128 / seg : 1;
  a.cols = seg > 0 ? (int)(p->row_bytes / seg) : 1;
  const long long tile_bytes = seg > 0 ? (long long)seg * p->tile_rows : 256 * 1024;
  a.n_tiles = (int)(p->rows * p->row_bytes / tile_bytes);
  const int per_cu = p->wgs_per_cu > 0 ? p->wgs_per_cu : 2;
  int slots = prop.multiProcessorCount * per_cu / a.gang * a.gang;
  if (slots < a.gang) slots = a.gang;
  a.tpw = (a.n_tiles + slots - 1) / slots;
  const int n_wg = a.gang * ((a.n_tiles + a.gang * a.tpw - 1) / (a.gang * a.tpw));
  hipStream_t stream = reinterpret_cast<hipStream_t>(p->stream);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess) { (void)hipSetDevice(prev); return SPECTRE_E_HIP; }
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); (void)hipSetDevice(prev); return SPECTRE_E_HIP; }
  auto go = [&] {
    if (ragged) hipLaunchKernelGGL(spectre_probe_copy_kernel<true>, dim3(n_wg), dim3(kProbeThreads), 0, stream, a);
    else hipLaunchKernelGGL(spectre_probe_copy_kernel<false>, dim3(n_wg), dim3(kProbeThreads), 0, stream, a);
  };
  for (int i = 0; i < warmup; ++i) go();
  hipError_t e = hipEventRecord(e0, stream);
  for (int i = 0; i < iters; ++i) go();
  if (e == hipSuccess) e = hipEventRecord(e1, stream);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  if (e == hipSuccess) e = hipGetLastError();
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipSetDevice(prev);
  if (e != hipSuccess) return SPECTRE_E_HIP;
  *ms_per_launch = ms / (float)iters;
  *why = "";
  return SPECTRE_OK;
}

}  // namespace sfft
