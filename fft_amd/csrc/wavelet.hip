// wavelet.hip — host side of the wavelet refinement launches (kernel_wavelet.h); the C-ABI entry points spectre_wavelet_refine /
// spectre_wavelet_gate_grad (spectre_hip.hip) validate nothing themselves and report this unit's `why` through spectre_last_error().
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "kernel_wavelet.h"
#include "../../include/spectre_hip.h"

namespace sfft {

namespace {
struct DeviceScope {
  int prev = -1;
  bool ok = true;
  explicit DeviceScope(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
  }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

// channels per workgroup: the widest power of two (<= 16 = one 64-byte fp32 segment per row) whose tile fits the LDS budget
int wavelet_channels(int64_t N) {
  int C = 16;
  while (C > 1 && N * C > kWaveletMaxFloats) C >>= 1;
  return N * C <= kWaveletMaxFloats ? C : 0;
}

int wavelet_refine(const SpectreWaveletArgs* p, const char** why) {
  if (!p) { *why = "args is NULL"; return SPECTRE_E_INVALID; }
  if (p->B < 0 || p->N < 1 || p->D < 1) { *why = "bad sizes"; return SPECTRE_E_INVALID; }
  if (p->dtype != SPECTRE_F32 && p->dtype != SPECTRE_BF16) { *why = "dtype must be SPECTRE_F32 or SPECTRE_BF16"; return SPECTRE_E_UNSUPPORTED; }
  if (p->N & (p->N - 1)) {
    // the reference's own analysis / synthesis pair fails with a size mismatch as soon as a level has odd length (spectre.py:271)
    *why = "the Haar round trip needs a power-of-two sequence length (the reference raises for any other, spectre.py:271)";
    return SPECTRE_E_UNSUPPORTED;
  }
  int C = wavelet_channels(p->N);
  if (!C) { *why = "sequence too long for the LDS-resident Haar round trip (N <= 32768)"; return SPECTRE_E_UNSUPPORTED; }
  if (p->B == 0) return SPECTRE_OK;
  if (p->B * ((p->D + 3) / 4) >= ((int64_t)1 << 31)) { *why = "too many tiles for one launch"; return SPECTRE_E_UNSUPPORTED; }
  if (!p->v || !p->out || !p->mask || !p->gate) { *why = "v, out, mask and gate must be non-NULL device pointers"; return SPECTRE_E_INVALID; }
  if (p->vref && (p->vref == p->v || p->vref == p->out)) { *why = "vref must not alias v or out"; return SPECTRE_E_INVALID; }
  DeviceScope g(p->device);
  if (!g.ok) { *why = "cannot select the device"; return SPECTRE_E_HIP; }
  WaveletArgs a{};
  a.v = p->v; a.out = p->out; a.vref = p->vref; a.mask = static_cast<const unsigned char*>(p->mask); a.gate = static_cast<const float*>(p->gate);
  a.B = (int)p->B; a.N = (int)p->N; a.D = (int)p->D;
  a.levels = 0;
  while (((int64_t)1 << a.levels) < p->N) ++a.levels;          // int(log2(N)) levels, down to one approximation sample (spectre.py:296, :307)
  a.v_sb = p->v_sb; a.v_sn = p->v_sn; a.out_sb = p->out_sb; a.out_sn = p->out_sn; a.ref_sb = p->ref_sb; a.ref_sn = p->ref_sn;
  hipStream_t stream = reinterpret_cast<hipStream_t>(p->stream);
  // 16-byte (8-byte for bf16) accesses where every one of them stays aligned: whole tiles, channel strides and bases multiples of four
  const int64_t esz = p->dtype == SPECTRE_BF16 ? 2 : 4;
  auto al = [&](const void* q, int64_t sb, int64_t sn) { return !q || (reinterpret_cast<uintptr_t>(q) % (4 * esz) == 0 && sb % 4 == 0 && sn % 4 == 0); };
  const bool aligned = al(p->v, p->v_sb, p->v_sn) && al(p->out, p->out_sb, p->out_sn) && al(p->vref, p->ref_sb, p->ref_sn);
  hipError_t e = hipSuccess;
  size_t lds = 0;
  dim3 grid, block(kWaveletThreads);
  auto go = [&](auto kernel) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) hipLaunchKernelGGL(kernel, grid, block, lds, stream, a);
  };
  // level 0 in registers: the approximation band of a tile of Cr channels is N / 2 * Cr floats (<= 128 KiB), a thread holds P = N * Cr / 4096 pairs
  int Cr = 16;
  while (Cr > 4 && (p->N / 2) * Cr > kWaveletMaxFloats) Cr >>= 1;
  const int64_t P = p->N * Cr / 4096;
  if (aligned && p->N >= 256 && (p->N / 2) * Cr <= kWaveletMaxFloats && p->D % Cr == 0 && P >= 1 && P <= 16) {
    a.C = Cr;
    lds = (size_t)(p->N / 2) * Cr * sizeof(float);
    a.tiles = (int)(p->D / Cr);
    a.gang = (int)std::max<int64_t>(1, 128 / (Cr * esz));
    grid = dim3((unsigned)(a.tiles * p->B));
    const bool bf = p->dtype == SPECTRE_BF16;
    switch ((int)P) {
      case 1: if (bf) go(spectre_wavelet_refine_regs_kernel<true, 1>); else go(spectre_wavelet_refine_regs_kernel<false, 1>); break;
      case 2: if (bf) go(spectre_wavelet_refine_regs_kernel<true, 2>); else go(spectre_wavelet_refine_regs_kernel<false, 2>); break;
      case 4: if (bf) go(spectre_wavelet_refine_regs_kernel<true, 4>); else go(spectre_wavelet_refine_regs_kernel<false, 4>); break;
      case 8: if (bf) go(spectre_wavelet_refine_regs_kernel<true, 8>); else go(spectre_wavelet_refine_regs_kernel<false, 8>); break;
      default: if (bf) go(spectre_wavelet_refine_regs_kernel<true, 16>); else go(spectre_wavelet_refine_regs_kernel<false, 16>); break;
    }
  } else {
    a.C = C;
    lds = (size_t)p->N * C * sizeof(float);
    a.tiles = (int)((p->D + C - 1) / C);
    a.gang = (int)std::max<int64_t>(1, 128 / (C * esz));
    grid = dim3((unsigned)(a.tiles * p->B));
    const bool vec = C >= 4 && p->D % C == 0 && aligned;
    if (p->dtype == SPECTRE_BF16) { if (vec) go(spectre_wavelet_refine_kernel<true, 4>); else go(spectre_wavelet_refine_kernel<true, 1>); }
    else { if (vec) go(spectre_wavelet_refine_kernel<false, 4>); else go(spectre_wavelet_refine_kernel<false, 1>); }
  }
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) { *why = hipGetErrorString(e); return SPECTRE_E_HIP; }
  return SPECTRE_OK;
}

int wavelet_gate_grad(const SpectreWaveletGradArgs* p, const char** why) {
  if (!p) { *why = "args is NULL"; return SPECTRE_E_INVALID; }
  if (p->B < 0 || p->N < 1 || p->D < 1) { *why = "bad sizes"; return SPECTRE_E_INVALID; }
  if (p->dtype != SPECTRE_F32 && p->dtype != SPECTRE_BF16) { *why = "dtype must be SPECTRE_F32 or SPECTRE_BF16"; return SPECTRE_E_UNSUPPORTED; }
  if (p->B == 0) return SPECTRE_OK;
  if (p->B * ((p->D + 63) / 64) >= ((int64_t)1 << 31)) { *why = "too many tiles for one launch"; return SPECTRE_E_UNSUPPORTED; }
  if (!p->dout || !p->vref || !p->mask || !p->dgate) { *why = "dout, vref, mask and dgate must be non-NULL device pointers"; return SPECTRE_E_INVALID; }
  DeviceScope g(p->device);
  if (!g.ok) { *why = "cannot select the device"; return SPECTRE_E_HIP; }
  WaveletGradArgs a{};
  a.dout = p->dout; a.vref = p->vref; a.mask = static_cast<const unsigned char*>(p->mask); a.dgate = static_cast<float*>(p->dgate);
  a.B = (int)p->B; a.N = (int)p->N; a.D = (int)p->D;
  a.d_sb = p->d_sb; a.d_sn = p->d_sn; a.ref_sb = p->ref_sb; a.ref_sn = p->ref_sn;
  const dim3 grid((unsigned)(((p->D + 63) / 64) * p->B)), block(512);
  hipStream_t stream = reinterpret_cast<hipStream_t>(p->stream);
  if (p->dtype == SPECTRE_BF16) hipLaunchKernelGGL(spectre_wavelet_gate_grad_kernel<true>, grid, block, 0, stream, a);
  else hipLaunchKernelGGL(spectre_wavelet_gate_grad_kernel<false>, grid, block, 0, stream, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { *why = hipGetErrorString(e); return SPECTRE_E_HIP; }
  return SPECTRE_OK;
}

}  // namespace sfft
