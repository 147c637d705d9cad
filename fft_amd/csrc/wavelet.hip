// wavelet.hip — host side of the wavelet refinement launches (kernel_wavelet.h); the C-ABI entry points spectre_wavelet_refine /
// spectre_wavelet_gate_grad (spectre_hip.hip) validate nothing themselves and report this unit's `why` through spectre_last_error().
#include <hip/hip_runtime.h>
#include <stdint.h>
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
  const int C = wavelet_channels(p->N);
  if (!C) { *why = "sequence too long for the LDS-resident Haar round trip (N <= 32768)"; return SPECTRE_E_UNSUPPORTED; }
  if (p->B == 0) return SPECTRE_OK;
  if (p->B > 65535) { *why = "B > 65535"; return SPECTRE_E_UNSUPPORTED; }
  if (!p->v || !p->out || !p->mask || !p->gate) { *why = "v, out, mask and gate must be non-NULL device pointers"; return SPECTRE_E_INVALID; }
  if (p->vref && (p->vref == p->v || p->vref == p->out)) { *why = "vref must not alias v or out"; return SPECTRE_E_INVALID; }
  DeviceScope g(p->device);
  if (!g.ok) { *why = "cannot select the device"; return SPECTRE_E_HIP; }
  WaveletArgs a{};
  a.v = p->v; a.out = p->out; a.vref = p->vref; a.mask = static_cast<const unsigned char*>(p->mask); a.gate = static_cast<const float*>(p->gate);
  a.B = (int)p->B; a.N = (int)p->N; a.D = (int)p->D; a.C = C;
  a.levels = 0;
  while (((int64_t)1 << a.levels) < p->N) ++a.levels;          // int(log2(N)) levels, down to one approximation sample (spectre.py:296, :307)
  a.v_sb = p->v_sb; a.v_sn = p->v_sn; a.out_sb = p->out_sb; a.out_sn = p->out_sn; a.ref_sb = p->ref_sb; a.ref_sn = p->ref_sn;
  const size_t lds = (size_t)p->N * C * sizeof(float);
  const dim3 grid((unsigned)((p->D + C - 1) / C), (unsigned)p->B), block(kWaveletThreads);
  hipStream_t stream = reinterpret_cast<hipStream_t>(p->stream);
  hipError_t e;
  if (p->dtype == SPECTRE_BF16) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(spectre_wavelet_refine_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) hipLaunchKernelGGL(spectre_wavelet_refine_kernel<true>, grid, block, lds, stream, a);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(spectre_wavelet_refine_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) hipLaunchKernelGGL(spectre_wavelet_refine_kernel<false>, grid, block, lds, stream, a);
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
  if (p->B > 65535) { *why = "B > 65535"; return SPECTRE_E_UNSUPPORTED; }
  if (!p->dout || !p->vref || !p->mask || !p->dgate) { *why = "dout, vref, mask and dgate must be non-NULL device pointers"; return SPECTRE_E_INVALID; }
  DeviceScope g(p->device);
  if (!g.ok) { *why = "cannot select the device"; return SPECTRE_E_HIP; }
  WaveletGradArgs a{};
  a.dout = p->dout; a.vref = p->vref; a.mask = static_cast<const unsigned char*>(p->mask); a.dgate = static_cast<float*>(p->dgate);
  a.B = (int)p->B; a.N = (int)p->N; a.D = (int)p->D;
  a.d_sb = p->d_sb; a.d_sn = p->d_sn; a.ref_sb = p->ref_sb; a.ref_sn = p->ref_sn;
  const dim3 grid((unsigned)((p->D + 63) / 64), (unsigned)p->B), block(512);
  hipStream_t stream = reinterpret_cast<hipStream_t>(p->stream);
  if (p->dtype == SPECTRE_BF16) hipLaunchKernelGGL(spectre_wavelet_gate_grad_kernel<true>, grid, block, 0, stream, a);
  else hipLaunchKernelGGL(spectre_wavelet_gate_grad_kernel<false>, grid, block, 0, stream, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { *why = hipGetErrorString(e); return SPECTRE_E_HIP; }
  return SPECTRE_OK;
}

}  // namespace sfft
