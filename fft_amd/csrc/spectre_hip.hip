// spectre_hip.hip — C ABI (include/spectre_hip.h), plan cache and kernel dispatch of libspectre_hip.so.
//
// Host side of the MI355X spectral mix: picks the register-resident R x R kernel when the shape and the
// buffers allow it, the LDS Stockham / Bluestein kernel otherwise, and fails loudly for anything else.
// No computation happens on the host beyond twiddle/chirp tables (double precision, once per plan).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/spectre_hip.h"
#include "kernel_regtile_grad.h"
#include "kernel_regtile64p.h"   // (the ticket-slice layout constants)
#include "kernel_tickets.h"
#include "kernel_regtile_wide.h"
#include "kernel_regtile_mixed_grad.h"
#include "kernel_stockham.h"
#include "kernel_gate.h"
#include "kernel_gate_grad_twopass.h"
#include "kernel_decode.h"

namespace sfft {
// defined in regtile_n*.hip (one translation unit per length)
template <> hipError_t launch_regtile<16, 16>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile<32, 16>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile<32, 32>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile<64, 32>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile<64, 64>(const RegtileArgs&, bool, bool, int, hipStream_t);
int probe_copy(const SpectreProbeArgs*, int warmup, int iters, float* ms_per_launch, const char** why);                  // copy_probe.hip
int wavelet_refine(const SpectreWaveletArgs*, const char** why);                                                           // wavelet.hip
int wavelet_gate_grad(const SpectreWaveletGradArgs*, const char** why);
hipError_t launch_regtile64p(const RegtileArgs&, bool in_bf16, bool out_bf16, bool burst, hipStream_t);                           // regtile_n4096p.hip (persistent, pipelined)
template <int RF, int RS> hipError_t launch_regtile_mixedp(const RegtileArgs&, hipStream_t);        // kernel_regtile_mixedp.h (persistent, deferred row blocks)
template <> hipError_t launch_regtile_mixedp<60, 50>(const RegtileArgs&, hipStream_t);               // regtile_mixedp.hip
template <> hipError_t launch_regtile_mixedp<64, 40>(const RegtileArgs&, hipStream_t);
template <> hipError_t launch_regtile_mixedp<60, 40>(const RegtileArgs&, hipStream_t);
template <> hipError_t launch_regtile_mixedp<64, 48>(const RegtileArgs&, hipStream_t);
template <> hipError_t launch_regtile_mixedp<60, 60>(const RegtileArgs&, hipStream_t);
template <> hipError_t launch_regtile_mixedp<64, 60>(const RegtileArgs&, hipStream_t);
hipError_t launch_regtile_long_8192(const RegtileArgs&, bool, bool, int, hipStream_t);   // regtile_n8192.hip, regtile_n6144.hip
hipError_t launch_regtile_long_6144(const RegtileArgs&, bool, bool, int, hipStream_t);
hipError_t launch_regtile_quad_16384(const RegtileArgs&, bool, bool, int, hipStream_t);   // regtile_n16384.hip, regtile_n12288.hip
hipError_t launch_regtile_quad_12288(const RegtileArgs&, bool, bool, int, hipStream_t);
hipError_t launch_gate_grad_long_8192(const GateGradArgs&, bool, bool, hipStream_t);
hipError_t launch_gate_grad_long_6144(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_regtile_mixed<60, 50>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<32, 24>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<48, 32>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<64, 48>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<40, 25>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<50, 40>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<40, 32>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<64, 40>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<64, 60>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<8, 8>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<16, 8>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<14, 14>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<24, 16>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<32, 20>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<32, 30>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<40, 30>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<48, 40>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<60, 40>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_regtile_mixed<60, 60>(const RegtileArgs&, bool, bool, int, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<8, 8>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<16, 8>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<14, 14>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<24, 16>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<32, 20>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<32, 30>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<40, 30>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<48, 40>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<60, 40>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<60, 60>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_regtile<16, 16>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_regtile<32, 16>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_regtile<32, 32>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_regtile<64, 32>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_regtile<64, 64>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<60, 50>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<25, 40>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<32, 24>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<48, 32>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<64, 48>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<50, 40>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<40, 32>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<64, 40>(const GateGradArgs&, bool, bool, hipStream_t);
template <> hipError_t launch_gate_grad_mixed<64, 60>(const GateGradArgs&, bool, bool, hipStream_t);
}  // namespace sfft

namespace {

// register-resident kernels: n_fft = RF * RS
using TileLauncher = hipError_t (*)(const sfft::RegtileArgs&, bool, bool, int, hipStream_t);
using GradLauncher = hipError_t (*)(const sfft::GateGradArgs&, bool, bool, hipStream_t);
struct TileSize {
  int n, RF, RS;
  bool mixed;        // one tile per workgroup, plain tile order (kernel_regtile_mixed.h, kernel_regtile_long.h)
  bool same_dtype;   // built for f32->f32 and bf16->bf16 only
  TileLauncher launch;
  GradLauncher grad;
  int tile_ch = 16;  // channels per tile (8 for the lane-pair kernels, 4 for the lane-quad kernels)   // register-resident gate gradient, or nullptr (LDS Stockham path)
};
const TileSize kTileSizes[] = {
    {256, 16, 16, false, false, &sfft::launch_regtile<16, 16>, &sfft::launch_gate_grad_regtile<16, 16>},
    {512, 32, 16, false, false, &sfft::launch_regtile<32, 16>, &sfft::launch_gate_grad_regtile<32, 16>},
    {1024, 32, 32, false, false, &sfft::launch_regtile<32, 32>, &sfft::launch_gate_grad_regtile<32, 32>},
    {2048, 64, 32, false, false, &sfft::launch_regtile<64, 32>, &sfft::launch_gate_grad_regtile<64, 32>},
    {4096, 64, 64, false, false, &sfft::launch_regtile<64, 64>, &sfft::launch_gate_grad_regtile<64, 64>},
    {16384, 64, 256, true, true, &sfft::launch_regtile_quad_16384, nullptr, 4},   // 4-channel tiles, lane-quad 256-point transform
    {12288, 48, 256, true, true, &sfft::launch_regtile_quad_12288, nullptr, 4},
    {8192, 64, 128, true, false, &sfft::launch_regtile_long_8192, &sfft::launch_gate_grad_long_8192, 8},
    {6144, 48, 128, true, false, &sfft::launch_regtile_long_6144, &sfft::launch_gate_grad_long_6144, 8},
    {3000, 60, 50, true, false, &sfft::launch_regtile_mixed<60, 50>, &sfft::launch_gate_grad_mixed<60, 50>},
    {768, 32, 24, true, true, &sfft::launch_regtile_mixed<32, 24>, &sfft::launch_gate_grad_mixed<32, 24>},
    {1536, 48, 32, true, true, &sfft::launch_regtile_mixed<48, 32>, &sfft::launch_gate_grad_mixed<48, 32>},
    {3072, 64, 48, true, true, &sfft::launch_regtile_mixed<64, 48>, &sfft::launch_gate_grad_mixed<64, 48>},
    {1000, 40, 25, true, true, &sfft::launch_regtile_mixed<40, 25>, &sfft::launch_gate_grad_mixed<25, 40>},   // (gradient: 25 x 40, even RS)
    {2000, 50, 40, true, true, &sfft::launch_regtile_mixed<50, 40>, &sfft::launch_gate_grad_mixed<50, 40>},
    {1280, 40, 32, true, true, &sfft::launch_regtile_mixed<40, 32>, &sfft::launch_gate_grad_mixed<40, 32>},
    {2560, 64, 40, true, true, &sfft::launch_regtile_mixed<64, 40>, &sfft::launch_gate_grad_mixed<64, 40>},
    {3840, 64, 60, true, true, &sfft::launch_regtile_mixed<64, 60>, &sfft::launch_gate_grad_mixed<64, 60>},
    {64, 8, 8, true, true, &sfft::launch_regtile_mixed<8, 8>, &sfft::launch_gate_grad_mixed<8, 8>},
    {128, 16, 8, true, true, &sfft::launch_regtile_mixed<16, 8>, &sfft::launch_gate_grad_mixed<16, 8>},
    {196, 14, 14, true, true, &sfft::launch_regtile_mixed<14, 14>, &sfft::launch_gate_grad_mixed<14, 14>},
    {384, 24, 16, true, true, &sfft::launch_regtile_mixed<24, 16>, &sfft::launch_gate_grad_mixed<24, 16>},
    {640, 32, 20, true, true, &sfft::launch_regtile_mixed<32, 20>, &sfft::launch_gate_grad_mixed<32, 20>},
    {960, 32, 30, true, true, &sfft::launch_regtile_mixed<32, 30>, &sfft::launch_gate_grad_mixed<32, 30>},
    {1200, 40, 30, true, true, &sfft::launch_regtile_mixed<40, 30>, &sfft::launch_gate_grad_mixed<40, 30>},
    {1920, 48, 40, true, true, &sfft::launch_regtile_mixed<48, 40>, &sfft::launch_gate_grad_mixed<48, 40>},
    {2400, 60, 40, true, true, &sfft::launch_regtile_mixed<60, 40>, &sfft::launch_gate_grad_mixed<60, 40>},
    {3600, 60, 60, true, true, &sfft::launch_regtile_mixed<60, 60>, &sfft::launch_gate_grad_mixed<60, 60>},
};
const TileSize* find_tile_size(int64_t n) {
  for (const TileSize& t : kTileSizes)
    if (t.n == n) return &t;
  return nullptr;
}

thread_local std::string g_err;

// Tuning aids (A/B switches for measurements): the SPECTRE_* environment variables below are read ONLY when the master
// switch SPECTRE_TUNING=1 is set — the shipped library's dispatch cannot be changed from the environment by accident — and every
// override that is in force is named by spectre_mix_describe ("[tuning: ...]").
const char* tuning_env(const char* name) {
  static const bool on = [] { const char* e = getenv("SPECTRE_TUNING"); return e && atoi(e) == 1; }();
  return on ? getenv(name) : nullptr;
}
std::string tuning_overrides() {
  std::string r;
  for (const char* n : {"SPECTRE_P64", "SPECTRE_P64_BF16", "SPECTRE_P64_BURST", "SPECTRE_P64_TICKETS", "SPECTRE_MIXEDP_TICKETS", "SPECTRE_TILE_ORDER", "SPECTRE_WIDE", "SPECTRE_WIDE_MAX", "SPECTRE_WIDE_NT", "SPECTRE_MIXEDP", "SPECTRE_STOCKHAM_PMAX", "SPECTRE_TPW", "SPECTRE_P64_TPW", "SPECTRE_GATE_GRAD", "SPECTRE_DGATE_PREFETCH", "SPECTRE_DGATE_GRID"})
    if (const char* e = tuning_env(n)) r += std::string(r.empty() ? "" : " ") + n + "=" + e;
  return r;
}

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

constexpr int kDgatePrefetch = 1;          // 64 x 64 gate gradient: the persistent form with prefetch registers (kernel_regtile_grad.h, PN / EARLY)
constexpr size_t kLdsBytes = 160 * 1024;   // gfx950: 160 KiB per CU, one workgroup may take all of it

constexpr int kTicketSlices = 128;
constexpr int kOrderWarmLaunches = 24;
constexpr int kOrderClasses = 32;          // shape classes a plan keeps a decision for (SPECTRE_ORDER_AUTO)
constexpr int kOrderPairs = 64;            // (V, out) pairs it keeps one for (SPECTRE_ORDER_AUTO_PAIR)
constexpr int kOrderExploreCap = 1024;     // event-timed launches a plan may spend on measuring, all classes / pairs together
constexpr float kOrderMargin = 0.99f;      // the static map replaces the default (tickets) only where it measures at least 1 % faster

struct Plan {
  int device = 0;
  int64_t n = 0;
  // length-n table exp(-2 pi i m / n): regtile twiddle bases and smooth-n Stockham passes
  float2* tw_n = nullptr;
  // Stockham factorisation of n (empty if n has a prime factor > 13)
  std::vector<int> radix_n;
  // Bluestein
  bool bluestein = false;
  int64_t m = 0;
  std::vector<int> radix_m;
  float2* tw_m = nullptr;
  float2* chirp = nullptr;
  float2* bhat = nullptr;
  // n_fft = 4096 / 3000 / 3600 / 3840: ticket slices for the dynamic tile order of kernel_regtile64p.h / kernel_tickets.h (TICKETS): UNCACHED
  // device memory (scalar atomics carry no scope bits, so only memory the L2 does not keep is coherent between XCDs for them).
  // WHO MAY USE A SLICE (round 6): a slice belongs to ONE stream for the life of the plan (slice_of_stream) — launches on
  // one stream execute in order, and every ticket launch is preceded on its stream by the reset of its slice, so a slice is never
  // zeroed under a running kernel and two kernels never draw from one counter, whatever the other streams do.  (Round 5 handed the
  // slices out round-robin: a stream that stalled behind an event while another issued 64 launches, or a graph replay beside eager
  // launches, could wrap the ring onto a slice still in use — missing tiles.)  A launch on a CAPTURING stream gets a slice of its own,
  // for good: the graph replays it on whatever stream it is launched into, beside anything.  (One hipGraphExec never overlaps itself;
  // two execs instantiated from ONE captured graph share its kernel arguments and must not be launched concurrently — or pin the
  // static order for the capture.)  No free slice left, or no ring (allocation refused): the static tile map.
  unsigned* tk_ring = nullptr;
  mutable std::unordered_map<hipStream_t, int> slice_of_stream;
  mutable int slices_used = 0;
  // WHICH order a launch takes (SPECTRE_ORDER_*, spectre_plan_set_tile_order; default AUTO).  The static map wins by 2-6 % where the
  // driver has placed the two tensors in memory of the fast class and loses by 3-5 % elsewhere — nothing a library
  // can see from a pointer, so it is measured: behind the first 24 launches of a CLASS (n_fft, shape, dtypes, strides — round 6: no
  // pointers, so the decision survives an allocator that hands out a fresh `out` every call; tickets meanwhile: the chip needs ~25
  // launches to leave its idle power state, and a measurement taken while the clock ramps up picked the wrong order in bench.py)
  // sixteen launches take the two orders in the pattern T S S T (what is left of a drift cancels out) with a HIP event pair around
  // each (recorded on the caller's stream, looked at later with hipEventQuery: nothing ever waits), and once all sixteen have finished
  // the class takes the static map if its median (behind the first sample of each) is at least 1 % below the ticket order's, tickets
  // otherwise — a decision inside the noise (round 5: bf16 -> bf16 at 1.1275 against 1.1278 ms) stays with the default.  One decision
  // per class, for good (spectre_plan_set_tile_order starts over).  SPECTRE_ORDER_AUTO_PAIR keys the same procedure on the (V, out)
  // pointer pair as well (round 5's behaviour: worth 2-6 % where a process reuses a few buffers of different allocation classes; an
  // LRU of 64 pairs; a plan stops measuring behind 1024 timed launches in all).  Same bits either way.  Nothing is measured under
  // stream capture (the decision so far, or tickets).
  mutable std::atomic<int> order_policy{SPECTRE_ORDER_AUTO};
  mutable int explore_spent = 0;
  struct OrderPending { hipEvent_t e0, e1; int mode; };
  struct OrderEntry {
    uint64_t key[6] = {0, 0, 0, 0, 0, 0};
    int decided = -1;                              // -1 exploring, 0 static, 1 tickets
    int launches = 0;                              // of this class / pair so far (the measurement starts behind kOrderWarmLaunches of them)
    int issued[2] = {0, 0}, samples[2] = {0, 0};
    float ms[2][8] = {{0}, {0}};                   // the samples of each order, in issue order
    float med[2] = {0.f, 0.f};                     // medians behind the first sample of each (what the decision was taken on)
    char name[72] = "auto";
    bool pair = false;                             // keyed on the (V, out) pointers as well (SPECTRE_ORDER_AUTO_PAIR)
    std::vector<OrderPending> pending;
    uint64_t last_use = 0;
  };
  mutable std::mutex order_mu;
  mutable std::vector<OrderEntry> orders;
  mutable uint64_t order_clock = 0;

  ~Plan() {
    // best effort: the owning device must be current for hipFree
    int cur = 0;
    if (hipGetDevice(&cur) == hipSuccess) {
      (void)hipSetDevice(device);
      if (tw_n) (void)hipFree(tw_n);
      if (tw_m) (void)hipFree(tw_m);
      if (chirp) (void)hipFree(chirp);
      if (bhat) (void)hipFree(bhat);
      if (tk_ring) (void)hipFree(tk_ring);
      for (auto& en : orders) for (auto& pd : en.pending) { (void)hipEventDestroy(pd.e0); (void)hipEventDestroy(pd.e1); }
      (void)hipSetDevice(cur);
    }
  }
};

std::mutex g_mu;
std::map<std::pair<int, int64_t>, std::unique_ptr<Plan>> g_plans;
// Plans taken out of service by spectre_plan_destroy.  They are NOT freed: a launch path uses its plan (host object and device tables)
// after the registry lock is released, and a kernel reads the tables until it retires, so freeing here would be a use-after-free the
// library cannot see coming (round 2 documented it as a contract; round 3 removes it).  A retired plan keeps its device tables (32 KiB
// at n_fft = 4096; a long Bluestein length carries four tables, about 1 MiB), is put back into service by the next spectre_plan_create /
// launch for the same (device, n_fft), and is released by spectre_plans_release_retired (caller-synchronised) or at process exit.
std::map<std::pair<int, int64_t>, std::unique_ptr<Plan>> g_retired;

bool factorize(int64_t n, std::vector<int>& out) {
  out.clear();
  static const int radices[] = {16, 8, 4, 2, 25, 15, 5, 3, 7, 11, 13};
  for (int r : radices)
    while (n % r == 0 && n > 1) { out.push_back(r); n /= r; }
  return n == 1;
}

std::vector<float2> unit_table(int64_t n) {   // exp(-2 pi i m / n)
  std::vector<float2> t((size_t)n);
  const long double two_pi = 6.283185307179586476925286766559L;
  for (int64_t m = 0; m < n; ++m) {
    const long double a = two_pi * (long double)m / (long double)n;
    t[(size_t)m] = make_float2((float)cosl(a), (float)(-sinl(a)));
  }
  return t;
}

hipError_t upload(const std::vector<float2>& h, float2** d) {
  hipError_t e = hipMalloc(reinterpret_cast<void**>(d), h.size() * sizeof(float2));
  if (e != hipSuccess) return e;
  return hipMemcpy(*d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
}

// naive-free host FFT is not needed: the Bluestein filter spectrum is computed by an O(M log M) radix-2
// recursion in double precision (M is a power of two)
void fft_pow2(std::vector<std::pair<double, double>>& a) {
  const size_t n = a.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    for (size_t i = 0; i < n; i += len) {
      for (size_t k = 0; k < len / 2; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)len;
        const double wr = cos(ang), wi = sin(ang);
        auto& u = a[i + k];
        auto& v = a[i + k + len / 2];
        const double tr = v.first * wr - v.second * wi, ti = v.first * wi + v.second * wr;
        v = {u.first - tr, u.second - ti};
        u = {u.first + tr, u.second + ti};
      }
    }
  }
}

int build_plan(int device, int64_t n, Plan** out) {
  auto plan = std::make_unique<Plan>();
  plan->device = device;
  plan->n = n;
  hipError_t e = upload(unit_table(n), &plan->tw_n);
  if (e != hipSuccess) return fail(SPECTRE_E_HIP, "plan(n_fft=%lld): %s", (long long)n, hipGetErrorString(e));
  if (!factorize(n, plan->radix_n)) {
    plan->radix_n.clear();
    plan->bluestein = true;
    int64_t m = 1;
    while (m < 2 * n - 1) m <<= 1;
    plan->m = m;
    factorize(m, plan->radix_m);
    if ((e = upload(unit_table(m), &plan->tw_m)) != hipSuccess)
      return fail(SPECTRE_E_HIP, "plan(bluestein M=%lld): %s", (long long)m, hipGetErrorString(e));
    // chirp w[j] = exp(-i pi j^2 / n), angle reduced exactly through j^2 mod 2n
    std::vector<float2> w((size_t)n);
    std::vector<std::pair<double, double>> bt((size_t)m, {0.0, 0.0});
    for (int64_t j = 0; j < n; ++j) {
      const int64_t q = (j * j) % (2 * n);
      const double ang = M_PI * (double)q / (double)n;
      w[(size_t)j] = make_float2((float)cos(ang), (float)(-sin(ang)));
      bt[(size_t)j] = {cos(ang), sin(ang)};                 // conj(w[j])
      if (j > 0) bt[(size_t)(m - j)] = {cos(ang), sin(ang)};
    }
    fft_pow2(bt);
    std::vector<float2> bh((size_t)m);
    for (int64_t j = 0; j < m; ++j) bh[(size_t)j] = make_float2((float)bt[(size_t)j].first, (float)bt[(size_t)j].second);
    if ((e = upload(w, &plan->chirp)) != hipSuccess || (e = upload(bh, &plan->bhat)) != hipSuccess)
      return fail(SPECTRE_E_HIP, "plan(bluestein tables): %s", hipGetErrorString(e));
  }
  if (n == 4096 || n == 3000 || n == 3600 || n == 3840) {   // ticket ring (optional: without it the kernels keep the static tile map)
    void* ring = nullptr;
    if (hipExtMallocWithFlags(&ring, (size_t)kTicketSlices * sfft::kP64TkSliceWords * 4, hipDeviceMallocUncached) == hipSuccess) plan->tk_ring = static_cast<unsigned*>(ring);
    else (void)hipGetLastError();
  }
  *out = plan.get();
  g_plans[{device, n}] = std::move(plan);
  return SPECTRE_OK;
}

// The launch paths keep the raw pointer after the mutex is released: plans are therefore never freed while the process lives —
// spectre_plan_destroy retires them (g_retired above).
// Building a plan allocates and copies synchronously, which would invalidate a stream capture: refuse instead of corrupting
// the capture (the caller creates the plan first with spectre_plan_create or one eager call).
int get_plan(int device, int64_t n, Plan** out, hipStream_t stream = nullptr) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_plans.find({device, n});
  if (it != g_plans.end()) { *out = it->second.get(); return SPECTRE_OK; }
  auto rt = g_retired.find({device, n});
  if (rt != g_retired.end()) {                     // destroyed earlier: the tables are still there
    *out = rt->second.get();
    g_plans[{device, n}] = std::move(rt->second);
    g_retired.erase(rt);
    return SPECTRE_OK;
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
    return fail(SPECTRE_E_INVALID, "no plan for n_fft=%lld on device %d yet and the stream is capturing: call spectre_plan_create "
                                   "(or run the op once eagerly) before the capture", (long long)n, device);
  return build_plan(device, n, out);
}

struct Choice {
  bool regtile = false;
  const TileSize* tile = nullptr;
  int RF = 0, RS = 0;      // n_fft = RF * RS
  bool wide = false;       // 32-channel tiles: whole 128-byte lines per row (kernel_regtile_wide.h; n_fft <= 1024, fast mode)
  int mode = 0;            // 0 fast, 1 general (row predicates / gate from global), 2 general + memory_fft, 3 row predicates only
  bool mixedp = false;     // n_fft = 3000 / 2560 / 2400 / 3072 / 3600 / 3840, fp32 in/out, gate in LDS, 16-byte aligned input rows, 8-byte aligned output rows: persistent kernel with deferred and LDS-staged row blocks (kernel_regtile_mixedp.h)
  bool pipelined = false;  // n_fft = 4096 fast mode, fp32, 16-byte aligned rows: persistent software-pipelined kernel (kernel_regtile64p.h)
  // stockham
  int P = 0, S = 0, solo = 0;
  const char* why_not_regtile = "";
};

int validate(const SpectreMixArgs* a) {
  if (!a) return fail(SPECTRE_E_INVALID, "args is NULL");
  if (a->B == 0 && a->N_in >= 1 && a->n_fft >= 1 && a->D >= 1 && a->G_tot >= 1 && a->D % a->G_tot == 0)
    return SPECTRE_OK;   // empty batch: nothing to read or write, pointers may be NULL
  if (!a->v || !a->gate || !a->out) return fail(SPECTRE_E_INVALID, "v, gate and out must be non-NULL device pointers");
  if (a->B < 0 || a->N_in < 1 || a->n_fft < 1 || a->D < 1 || a->G_tot < 1)
    return fail(SPECTRE_E_INVALID, "bad sizes B=%lld N_in=%lld n_fft=%lld D=%lld G_tot=%lld", (long long)a->B,
                (long long)a->N_in, (long long)a->n_fft, (long long)a->D, (long long)a->G_tot);
  if (a->D % a->G_tot) return fail(SPECTRE_E_INVALID, "D=%lld not divisible by G_tot=%lld (spectre.py:422)", (long long)a->D, (long long)a->G_tot);
  if ((a->in_dtype != SPECTRE_F32 && a->in_dtype != SPECTRE_BF16) || (a->out_dtype != SPECTRE_F32 && a->out_dtype != SPECTRE_BF16))
    return fail(SPECTRE_E_UNSUPPORTED, "dtype must be SPECTRE_F32 or SPECTRE_BF16");
  if (a->algo < SPECTRE_ALGO_AUTO || a->algo > SPECTRE_ALGO_REGTILE) return fail(SPECTRE_E_INVALID, "bad algo %d", a->algo);
  const int64_t n_out = a->N_in < a->n_fft ? a->N_in : a->n_fft;
  if (a->v_sn < a->D || a->out_sn < a->D || a->v_sb < 0 || a->out_sb < 0)
    return fail(SPECTRE_E_INVALID, "row strides must be >= D and batch strides >= 0");
  (void)n_out;
  if (a->B > 0 && (a->B * a->D > (int64_t)1 << 40)) return fail(SPECTRE_E_INVALID, "B*D too large");
  return SPECTRE_OK;
}

int choose(const SpectreMixArgs* a, const Plan* plan, Choice* c) {
  const int64_t n = a->n_fft, D = a->D, d_g = D / a->G_tot;
  const int es_in = a->in_dtype == SPECTRE_BF16 ? 2 : 4, es_out = a->out_dtype == SPECTRE_BF16 ? 2 : 4;
  const TileSize* ts = find_tile_size(n);
  const char* why = "";
  if (!ts) why = "no register-tile kernel for this n_fft";
  else if (ts->same_dtype && a->in_dtype != a->out_dtype) why = "storage dtypes differ (not built for this n_fft)";
  else if (d_g % 2) why = "odd group width";
  else if ((reinterpret_cast<uintptr_t>(a->v) % (2 * es_in)) || (a->v_sn % 2) || (a->v_sb % 2)) why = "v not pair-aligned";
  else if ((reinterpret_cast<uintptr_t>(a->out) % (2 * es_out)) || (a->out_sn % 2) || (a->out_sb % 2)) why = "out not pair-aligned";
  else if (a->mem && (reinterpret_cast<uintptr_t>(a->mem) % 16)) why = "mem not 16-byte aligned";
  else if (reinterpret_cast<uintptr_t>(a->gate) % 8) why = "gate not 8-byte aligned";
  else if (a->v_sn * 255 * 4 + 64 >= ((int64_t)1 << 31) || a->out_sn * 255 * 4 + 64 >= ((int64_t)1 << 31)) why = "row stride too large";
  else if (ts->tile_ch == 16 && (n * a->v_sn * es_in >= ((int64_t)1 << 31) || n * a->out_sn * es_out >= ((int64_t)1 << 31)))
    why = "row stride too large for the 32-bit buffer offsets of the register-tile kernels";
  else if (ts->tile_ch < 16 && (a->v_sn * n * es_in + 64 >= ((int64_t)1 << 31) || a->out_sn * n * es_out + 64 >= ((int64_t)1 << 31)))
    why = "row stride too large for the 32-bit row offsets of the lane-pair / lane-quad kernels";
  else if (a->B * ((D + 3) / 4) >= ((int64_t)1 << 31)) why = "too many tiles";
  c->why_not_regtile = why;
  const bool can_regtile = why[0] == 0;
  if (a->algo == SPECTRE_ALGO_REGTILE && !can_regtile)
    return fail(why[0] == 'v' || why[0] == 'o' || why[0] == 'm' || why[0] == 'g' ? SPECTRE_E_ALIGN : SPECTRE_E_UNSUPPORTED,
                "register-tile kernel not applicable: %s", why);
  int mode = 0;
  if (can_regtile) {
    if (ts->tile_ch < 16) mode = a->mem ? 2 : ((a->N_in < a->n_fft) || (D % ts->tile_ch != 0)) ? 1 : 0;   // gate always from global
    else if (!ts->mixed) mode = (d_g % 16 != 0) ? (a->mem ? 2 : 1) : a->mem ? 4 : (a->N_in < a->n_fft) ? 3 : 0;   // 3, 4: gate still in LDS
    else mode = a->mem ? 2 : (d_g % 16 != 0) ? 1 : (a->N_in < a->n_fft) ? 3 : 0;
  }
  static const bool p64_off = [] { const char* e = tuning_env("SPECTRE_P64"); return e && atoi(e) == 0; }();   // A/B switch (tuning aid)
  // fast mode, padded sequences (mode 3: rows >= N_in are the buffer instructions' out-of-range case) and memory_fft (mode 4);
  // 32-bit byte offsets
  static const bool p64_bf16_off = [] { const char* e = tuning_env("SPECTRE_P64_BF16"); return e && atoi(e) == 0; }();
  const bool in_bf = a->in_dtype == SPECTRE_BF16;     // a lane moves the 4 channels of a row: 16 bytes of fp32, 8 of bf16
  const bool out_bf = a->out_dtype == SPECTRE_BF16;   // built: f32 -> f32 (+ memory_fft), bf16 -> f32, bf16 -> bf16
  const bool pipelined_ok = can_regtile && ts && !ts->mixed && ts->tile_ch == 16 && !p64_off && n == 4096 && (mode == 0 || mode == 3 || mode == 4) &&
                   (!out_bf || in_bf) && (!in_bf || (mode != 4 && !p64_bf16_off)) &&
                   reinterpret_cast<uintptr_t>(a->v) % (in_bf ? 8 : 16) == 0 && reinterpret_cast<uintptr_t>(a->out) % (out_bf ? 8 : 16) == 0 &&
                   a->v_sn % 4 == 0 && a->v_sb % 4 == 0 && a->out_sn % 4 == 0 && a->out_sb % 4 == 0 &&
                   a->v_sn * 4096 * 4 + 64 < ((int64_t)1 << 32) && a->out_sn * 4096 * 4 + 64 < ((int64_t)1 << 32);
  bool can = can_regtile;
  // differing storage dtypes outside the fast mode: bf16 rows in / fp32 rows out (activations under autocast, padded or ragged shapes) is
  // built for the five power-of-two lengths and for 3000; f32 -> bf16 and the secondary lengths take the general path
  const bool mixed_io_modes = in_bf && !out_bf && ts && !ts->same_dtype && ts->tile_ch == 16;
  if (can && mode != 0 && a->in_dtype != a->out_dtype && !pipelined_ok && !mixed_io_modes) {
    c->why_not_regtile = "storage dtypes differ (built for the fast mode only)";
    can = false;
    if (a->algo == SPECTRE_ALGO_REGTILE) return fail(SPECTRE_E_UNSUPPORTED, "register-tile kernel not applicable: %s", c->why_not_regtile);
  }
  if (can && a->algo != SPECTRE_ALGO_STOCKHAM) {
    c->regtile = true;
    c->tile = ts; c->RF = ts->RF; c->RS = ts->RS;
    c->mode = mode;
    c->pipelined = pipelined_ok;
    // whole-line tiles (round 4): a workgroup owns 32 channels, so every request is a full 128-byte line (bf16 rows: 64 bytes) instead of
    // half of one — the L2 takes half-line stores at two thirds of the rate (profiles/r04_store_lab_half_line_stores.log)
    static const bool wide_off = [] { const char* e = tuning_env("SPECTRE_WIDE"); return e && atoi(e) == 0; }();
    static const int wide_max = [] { const char* e = tuning_env("SPECTRE_WIDE_MAX"); return e ? atoi(e) : 1024; }();
    c->wide = !wide_off && !ts->mixed && ts->tile_ch == 16 && n <= wide_max && n <= 2048 && (mode == 0 || mode == 3) && d_g % 32 == 0 && D % 32 == 0 && (!out_bf || in_bf);
    static const bool mixedp_off = [] { const char* e = tuning_env("SPECTRE_MIXEDP"); return e && atoi(e) == 0; }();
    c->mixedp = !mixedp_off && ts->mixed && (n == 3000 || n == 2560 || n == 2400 || n == 3072 || n == 3600 || n == 3840) && (mode == 0 || mode == 3) && a->in_dtype == SPECTRE_F32 && a->out_dtype == SPECTRE_F32 &&
                reinterpret_cast<uintptr_t>(a->v) % 16 == 0 && reinterpret_cast<uintptr_t>(a->out) % 8 == 0 &&      // (input rows: 16-byte LDS-DMA requests, round 4)
                a->v_sn % 4 == 0 && a->v_sb % 4 == 0 && a->out_sn % 2 == 0 && a->out_sb % 2 == 0 &&
                a->v_sn * n * 4 < ((int64_t)1 << 31) && a->out_sn * n * 4 < ((int64_t)1 << 31);
    return SPECTRE_OK;
  }
  // Stockham / Bluestein in LDS: one buffer of L points per slot; P slots per workgroup, limited by the LDS and by
  // every pass having to be register-resident ((L/R)*P butterflies <= 1024 threads * stockham_kmax(R))
  const int64_t L = plan->bluestein ? plan->m : n;
  const std::vector<int>& rad = plan->bluestein ? plan->radix_m : plan->radix_n;
  c->solo = (d_g % 2) ? 1 : 0;
  c->S = (int)(c->solo ? D : D / 2);
  int64_t P = (int64_t)kLdsBytes / (8 * L);
  if (P > 16) P = 16;
  { static const int pmax = [] { const char* e = tuning_env("SPECTRE_STOCKHAM_PMAX"); return e ? atoi(e) : 0; }(); if (pmax > 0 && P > pmax) P = pmax; }
  if (P > c->S) P = c->S;
  for (int r : rad) {
    const int64_t cap = (int64_t)sfft::kStockhamMaxThreads * sfft::stockham_kmax(r) * r / L;
    if (P > cap) P = cap;
  }
  if (P < 1)
    return fail(SPECTRE_E_UNSUPPORTED,
                "n_fft=%lld needs %lld bytes of LDS per sequence (transform length %lld%s); the CU has %zu",
                (long long)n, (long long)(8 * L), (long long)L, plan->bluestein ? ", Bluestein" : "", kLdsBytes);
  c->P = (int)P;
  const int64_t groups = (c->S + P - 1) / P;
  if (a->B * groups >= ((int64_t)1 << 31)) return fail(SPECTRE_E_UNSUPPORTED, "grid too large");
  return SPECTRE_OK;
}

// Tiles per workgroup of the register-tile kernel.  Measured on MI355X (tools/size_sweep.py): looping over 2-4
// tiles inside a workgroup does not help at n_fft = 4096 (the workgroup launch is not what is exposed) and costs
// 5-10 % at n_fft <= 1024, where several workgroups per CU overlap each other and finer dispatch balances better.
// One tile per workgroup therefore; SPECTRE_TPW overrides (tuning aid).
int tiles_per_workgroup(int n_tiles) {
  static const int forced = [] { const char* e = tuning_env("SPECTRE_TPW"); return e ? atoi(e) : 0; }();
  (void)n_tiles;
  return forced > 0 ? forced : 1;
}

int cu_count(int device) {
  static int cached[64] = {};
  if (device >= 0 && device < 64 && cached[device] > 0) return cached[device];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n < 1) n = 256;
  if (device >= 0 && device < 64) cached[device] = n;
  return n;
}

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// n_fft = 4096 pipelined kernel: does this launch take the dynamic tile order?  (SPECTRE_P64_TICKETS=0: the static map, tuning aid.)
// Built for every shipped form (round 6: memory_fft too); the mailboxes of n_wg / gang gangs and one claim bit per tile have to fit the slice.
bool p64_tickets(const SpectreMixArgs* a, const Plan* plan, int n_tiles, int n_wg, int gang) {
  static const bool off = [] { const char* e = tuning_env("SPECTRE_P64_TICKETS"); return e && atoi(e) == 0; }();
#ifdef SPECTRE_P64_LEGACY
  static const bool burst_off = [] { const char* e2 = tuning_env("SPECTRE_P64_BURST"); return e2 && atoi(e2) == 0; }();
#else
  constexpr bool burst_off = false;
#endif
  // (whether an eligible launch then TAKES the ticket order is measured per tensor pair: choose_tile_order.  fp32 rows -3 ... -5 % on slow-class
  //  pairs, +2 ... +6 % on fast ones; bf16 rows in / fp32 out -0.2 ... -3.5 %; bf16 rows out +-0.8 %)
  return plan->tk_ring && !off && !burst_off && !(a->mem && a->in_dtype == SPECTRE_BF16) && n_tiles <= sfft::p64_ticket_capacity() &&
         n_wg / gang <= (sfft::kP64TkClaim - sfft::kP64TkBox) / 8 && n_wg >= gang;
}

// the persistent mixed-radix kernels built with tickets (regtile_mixedp.hip: the lengths whose deferred loads sit in the exchange gaps)
bool mixedp_tickets(const SpectreMixArgs* a, const Plan* plan, int n_tiles, int n_wg) {
  // At (256, 3000, 768) the dynamic order is worth -4.5 % on one box and costs +1.4 ... +3.9 % on four others (profiles/r05_tickets_lab_box*.log):
  // eligible, and taken where it measures faster (choose_tile_order).  SPECTRE_MIXEDP_TICKETS=0: never.
  static const bool off = [] { const char* e = tuning_env("SPECTRE_MIXEDP_TICKETS"); return e && atoi(e) == 0; }();
  const int64_t n = a->n_fft;
  return plan->tk_ring && !off && (n == 3000 || n == 3600 || n == 3840) && n_tiles <= sfft::tk_capacity() && n_wg / 2 <= sfft::tk_max_gangs() && n_wg >= 2;
}

// The reset of a ticket slice: OUR OWN kernel, not hipMemsetAsync.  Captured into a hipGraph, the runtime's memset node did not reliably
// clear the (uncached) slice on the second and later replays — the ticket kernel then found a spent counter and processed nothing
// (tools/graph_dbg.py) — while a kernel node is replayed like any other launch.  Same cost (a 10-KB store), same stream ordering.
__global__ void __launch_bounds__(256) spectre_ticket_reset(unsigned* slice, int words) {
  for (int i = threadIdx.x; i < words; i += 256) __hip_atomic_store(slice + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
hipError_t ticket_reset(unsigned* slice, size_t bytes, hipStream_t stream) {
  hipLaunchKernelGGL(spectre_ticket_reset, dim3(1), dim3(256), 0, stream, slice, (int)(bytes / 4));
  return hipGetLastError();
}

// Event and capture-status calls of the order measurement and of the slice hand-out are made with this thread's capture mode RELAXED:
// while ANOTHER stream is being captured in the (default) global mode, hipEventQuery / hipEventCreate from any thread would otherwise
// invalidate that capture.  Nothing here touches the capturing stream itself.
struct RelaxedCapture {
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  bool ok;
  RelaxedCapture() { ok = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess; if (!ok) (void)hipGetLastError(); }
  ~RelaxedCapture() { if (ok) (void)hipThreadExchangeStreamCaptureMode(&mode); }
};
// a status we cannot read counts as "capturing": no event calls, no measuring
bool stream_capturing(hipStream_t stream) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); return true; }
  return cs != hipStreamCaptureStatusNone;
}

// The ticket slice of this launch (Plan: who may use a slice), or nullptr = take the static map.
unsigned* take_slice(const Plan* plan, hipStream_t stream, bool capturing, size_t slice_words) {
  std::lock_guard<std::mutex> lk(plan->order_mu);
  int idx = -1;
  if (!capturing) {
    auto it = plan->slice_of_stream.find(stream);
    if (it != plan->slice_of_stream.end()) idx = it->second;
  }
  if (idx < 0) {
    if (plan->slices_used >= kTicketSlices) return nullptr;
    idx = plan->slices_used++;
    if (!capturing) plan->slice_of_stream.emplace(stream, idx);
  }
  return plan->tk_ring + (size_t)idx * slice_words;
}

// The order policy in force: SPECTRE_TILE_ORDER = auto | pair | tickets | static under SPECTRE_TUNING=1 (tests, A/B) overrides the plan's
// (spectre_plan_set_tile_order; default SPECTRE_ORDER_AUTO)
int tile_order_policy(const Plan* plan) {
  static const int env = [] { const char* e = tuning_env("SPECTRE_TILE_ORDER");
                              return !e ? -1 : !strcmp(e, "static") ? SPECTRE_ORDER_STATIC : !strcmp(e, "tickets") ? SPECTRE_ORDER_TICKETS
                                         : !strcmp(e, "pair") ? SPECTRE_ORDER_AUTO_PAIR : !strcmp(e, "auto") ? SPECTRE_ORDER_AUTO : -1; }();
  return env >= 0 ? env : plan->order_policy.load(std::memory_order_relaxed);
}

void order_entry_drop(Plan::OrderEntry& en) { for (auto& pd : en.pending) { (void)hipEventDestroy(pd.e0); (void)hipEventDestroy(pd.e1); } en.pending.clear(); }

Plan::OrderEntry* order_entry(const SpectreMixArgs* a, const Plan* plan, bool per_pair, bool create) {     // (plan->order_mu held)
  const uint64_t key[6] = {per_pair ? (uint64_t)(uintptr_t)a->v : 0, per_pair ? (uint64_t)(uintptr_t)a->out : 0, (uint64_t)a->B, (uint64_t)a->D,
                           (uint64_t)a->N_in * 8 + (a->mem ? 4u : 0u) + (uint64_t)a->in_dtype * 2 + (uint64_t)a->out_dtype, (uint64_t)a->v_sn ^ ((uint64_t)a->out_sn << 32)};
  for (auto& en : plan->orders) if (en.pair == per_pair && !memcmp(en.key, key, sizeof key)) { en.last_use = ++plan->order_clock; return &en; }
  if (!create) return nullptr;
  // full: forget the entry of the same kind that has not been used for the longest time (its events with it)
  size_t kind = 0, old = plan->orders.size();
  for (size_t i = 0; i < plan->orders.size(); ++i) {
    if (plan->orders[i].pair != per_pair) continue;
    ++kind;
    if (old == plan->orders.size() || plan->orders[i].last_use < plan->orders[old].last_use) old = i;
  }
  if (kind >= (size_t)(per_pair ? kOrderPairs : kOrderClasses)) {
    order_entry_drop(plan->orders[old]);
    plan->orders.erase(plan->orders.begin() + (long)old);
  }
  plan->orders.emplace_back();
  memcpy(plan->orders.back().key, key, sizeof key);
  plan->orders.back().pair = per_pair;
  if (per_pair) snprintf(plan->orders.back().name, sizeof plan->orders.back().name, "pair");
  plan->orders.back().last_use = ++plan->order_clock;
  return &plan->orders.back();
}

// Which order does THIS launch take (1 = tickets), and does it carry an event pair (returned in *ev, recorded by the caller around the launch)?
// dflt = the order this kernel takes until (and unless) the other one has measured at least 1 % faster: tickets at n_fft = 4096 (-3 ... -5 % on
// typical buffers, fp32 rows; bf16 rows in -0.2 ... -4.4 %), the static map for the persistent mixed-radix lengths (3000: tickets -4.5 % on one box
// of six, +1.4 ... +5.3 % on the others)
int choose_tile_order(const SpectreMixArgs* a, const Plan* plan, bool capturing, int dflt, Plan::OrderPending* ev, bool* timed) {
  *timed = false;
  const int pol = tile_order_policy(plan);
  if (pol == SPECTRE_ORDER_STATIC) return 0;
  if (pol == SPECTRE_ORDER_TICKETS) return 1;
  const bool per_pair = pol == SPECTRE_ORDER_AUTO_PAIR;
  std::lock_guard<std::mutex> lk(plan->order_mu);
  Plan::OrderEntry* en = order_entry(a, plan, per_pair, true);
  // a stream that is being captured: no event calls at all (a query of an outside event during the capture left the captured launch
  // without its slice reset on replay); the decision so far, or tickets
  if (capturing) return en->decided >= 0 ? en->decided : dflt;
  if (en->decided >= 0) return en->decided;
  RelaxedCapture rc;
  // harvest what has finished (in issue order; nothing waits)
  while (!en->pending.empty() && hipEventQuery(en->pending.front().e1) == hipSuccess) {
    Plan::OrderPending pd = en->pending.front();
    en->pending.erase(en->pending.begin());
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pd.e0, pd.e1) == hipSuccess && ms > 0.f) {
      if (en->samples[pd.mode] < 8) en->ms[pd.mode][en->samples[pd.mode]++] = ms;
    }
    (void)hipEventDestroy(pd.e0); (void)hipEventDestroy(pd.e1);
  }
  (void)hipGetLastError();                         // (hipEventQuery's hipErrorNotReady is not an error of ours)
  if (en->samples[0] >= 8 && en->samples[1] >= 8) {
    for (int m = 0; m < 2; ++m) { float t[7]; memcpy(t, en->ms[m] + 1, sizeof t); std::sort(t, t + 7); en->med[m] = t[3]; }
    en->decided = en->med[1 - dflt] < kOrderMargin * en->med[dflt] ? 1 - dflt : dflt;
    snprintf(en->name, sizeof en->name, "%s:%s (%.4f ms against %.4f)", per_pair ? "pair" : "auto", en->decided ? "tickets" : "static", en->med[en->decided], en->med[1 - en->decided]);
    return en->decided;
  }
  if (++en->launches <= kOrderWarmLaunches) return dflt;
  // measuring: T S S T T S S T T S S T T S S T, eight timed launches per order
  const int k = en->issued[0] + en->issued[1];
  const int mode = (k & 3) == 0 || (k & 3) == 3 ? 1 : 0;
  if (k >= 16) {
    // every sample is issued; if they do not all come back (a failed event: fewer than 8 samples an order) the class keeps the default
    if (en->pending.empty()) { en->decided = dflt; snprintf(en->name, sizeof en->name, "%s:%s (default: %d + %d samples)", per_pair ? "pair" : "auto", dflt ? "tickets" : "static", en->samples[1], en->samples[0]); }
    return dflt;
  }
  if (plan->explore_spent >= kOrderExploreCap) {   // this plan has measured enough: the default for whatever is still undecided
    en->decided = dflt; snprintf(en->name, sizeof en->name, "%s:%s (default: measuring budget spent)", per_pair ? "pair" : "auto", dflt ? "tickets" : "static");
    order_entry_drop(*en);
    return dflt;
  }
  if (hipEventCreate(&ev->e0) != hipSuccess) { (void)hipGetLastError(); return dflt; }
  if (hipEventCreate(&ev->e1) != hipSuccess) { (void)hipEventDestroy(ev->e0); (void)hipGetLastError(); return dflt; }
  ev->mode = mode;
  ++en->issued[mode];
  ++plan->explore_spent;
  *timed = true;
  return mode;
}

void tile_order_timed(const SpectreMixArgs* a, const Plan* plan, const Plan::OrderPending& ev) {     // the event pair is on the stream: remember it
  std::lock_guard<std::mutex> lk(plan->order_mu);
  const int pol = tile_order_policy(plan);
  Plan::OrderEntry* en = pol == SPECTRE_ORDER_AUTO || pol == SPECTRE_ORDER_AUTO_PAIR ? order_entry(a, plan, pol == SPECTRE_ORDER_AUTO_PAIR, false) : nullptr;
  if (en && en->decided < 0) en->pending.push_back(ev);
  else { (void)hipEventDestroy(ev.e0); (void)hipEventDestroy(ev.e1); }
}

const char* tile_order_name(const SpectreMixArgs* a, const Plan* plan) {
  const int pol = tile_order_policy(plan);
  if (pol == SPECTRE_ORDER_STATIC) return "static";
  if (pol == SPECTRE_ORDER_TICKETS) return "tickets";
  std::lock_guard<std::mutex> lk(plan->order_mu);
  const Plan::OrderEntry* en = order_entry(a, plan, pol == SPECTRE_ORDER_AUTO_PAIR, false);
  static thread_local char buf[80];
  snprintf(buf, sizeof buf, "%s", en ? en->name : pol == SPECTRE_ORDER_AUTO_PAIR ? "pair" : "auto");
  return buf;
}

// One ticket launch's bookkeeping, shared by the 4096 kernel and the persistent mixed-radix kernels: the order of this launch, its slice
// (reset on the launch's own stream), the event pair of a measured launch.  Returns the slice (nullptr: static map).
int ticket_launch_begin(const SpectreMixArgs* a, const Plan* plan, hipStream_t stream, int dflt, size_t slice_words, size_t used_bytes, Plan::OrderPending* ev, bool* timed, unsigned** slice) {
  *slice = nullptr;
  bool capturing;
  { RelaxedCapture rc; capturing = stream_capturing(stream); }
  {                                                // no slice to be had for this stream: the static map, and nothing to measure
    std::lock_guard<std::mutex> lk(plan->order_mu);
    if (plan->slices_used >= kTicketSlices && (capturing || !plan->slice_of_stream.count(stream))) return SPECTRE_OK;
  }
  const int tickets = choose_tile_order(a, plan, capturing, dflt, ev, timed);
  if (*timed) (void)hipEventRecord(ev->e0, stream);
  if (!tickets) return SPECTRE_OK;
  unsigned* sl = take_slice(plan, stream, capturing, slice_words);
  if (!sl) return SPECTRE_OK;                      // (another thread took the last one in between)
  const hipError_t e = ticket_reset(sl, used_bytes, stream);
  if (e != hipSuccess) return fail(SPECTRE_E_HIP, "ticket slice reset: %s", hipGetErrorString(e));
  *slice = sl;
  return SPECTRE_OK;
}

int launch(const SpectreMixArgs* a, const Plan* plan, const Choice& c, bool conj_gate = false) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(a->stream);
  if (a->B == 0) return SPECTRE_OK;
  hipError_t e;
  if (c.regtile) {
    sfft::RegtileArgs k{};
    k.v = a->v; k.gate = reinterpret_cast<const float2*>(a->gate); k.mem = reinterpret_cast<const float*>(a->mem);
    k.out = a->out; k.tw = plan->tw_n;
    k.B = (int)a->B; k.N_in = (int)std::min<int64_t>(a->N_in, (int64_t)1 << 30); k.D = (int)a->D; k.G = (int)a->G_tot;
    k.d_g = (int)(a->D / a->G_tot); k.F = (int)(a->n_fft / 2 + 1);
    const int64_t tch = c.tile->tile_ch;
    k.tiles_per_row = (int)((a->D + tch - 1) / tch); k.n_tiles = (int)(a->B * ((a->D + tch - 1) / tch));
    k.v_sb = a->v_sb; k.v_sn = a->v_sn; k.out_sb = a->out_sb; k.out_sn = a->out_sn;
    k.conj_gate = conj_gate ? 1 : 0;
    k.rows_in = k.rows_out = (int)std::min<int64_t>(a->N_in, a->n_fft);   // spectre.py:506 pads / truncates to n_fft, :553 keeps min(N, n_fft) rows
    k.tpw = tiles_per_workgroup(k.n_tiles);
    k.n_wg = 2 * ((k.n_tiles + 2 * k.tpw - 1) / (2 * k.tpw));
    const bool ib = a->in_dtype == SPECTRE_BF16, ob = a->out_dtype == SPECTRE_BF16;
    if (c.tile->mixed) { k.tpw = 1; k.n_wg = 2 * ((k.n_tiles + 1) / 2); }
    if (c.wide) {          // one whole-line tile per workgroup (kernel_regtile_wide.h)
      k.tiles_per_row = (int)(a->D / 32); k.n_tiles = (int)(a->B * (a->D / 32)); k.tpw = 1; k.n_wg = k.n_tiles;
      const int64_t n = a->n_fft;
      e = n == 2048 ? sfft::launch_regtile_wide<64, 32>(k, ib, ob, c.mode == 3, stream) : n == 1024 ? sfft::launch_regtile_wide<32, 32>(k, ib, ob, c.mode == 3, stream) : n == 512 ? sfft::launch_regtile_wide<32, 16>(k, ib, ob, c.mode == 3, stream)
                                                                                          : sfft::launch_regtile_wide<16, 16>(k, ib, ob, c.mode == 3, stream);
    } else if (c.pipelined) {   // one workgroup per CU walks through tpw tiles; pairs of workgroups stay on adjacent tiles
      const int ncu = cu_count(a->device);
      static const int forced = [] { const char* e = tuning_env("SPECTRE_P64_TPW"); return e ? atoi(e) : 0; }();
      // round 4: store burst behind a workgroup barrier + phased I/O + requests spread over the arithmetic — every variant (fp32 rows,
      // bf16 rows in and / or out, memory_fft).  SPECTRE_P64_BURST=0: the round-3 order (tuning aid)
#ifdef SPECTRE_P64_LEGACY
      static const bool burst_off = [] { const char* e2 = tuning_env("SPECTRE_P64_BURST"); return e2 && atoi(e2) == 0; }();
#else
      constexpr bool burst_off = false;   // (the round-3 order is built only with -DSPECTRE_P64_LEGACY: regtile_n4096p.hip)
#endif
      const int gang = (ib || ob) ? 4 : 2;   // = p64_gang() of kernel_regtile64p.h: workgroups that walk in step
      const int slots = std::max(gang, ncu / gang * gang);
      k.tpw = forced > 0 ? forced : std::max(1, (k.n_tiles + slots - 1) / slots);
      k.n_wg = gang * ((k.n_tiles + gang * k.tpw - 1) / (gang * k.tpw));
      // round 5: dynamic tile order (one ticket per gang from a chip-wide counter) where the ring exists and the launch fits a slice
      Plan::OrderPending ev{}; bool timed = false;
      if (p64_tickets(a, plan, k.n_tiles, k.n_wg, gang)) {
        unsigned* slice = nullptr;
        if (int rc = ticket_launch_begin(a, plan, stream, 1, sfft::kP64TkSliceWords, ((size_t)sfft::kP64TkClaim + (size_t)(k.n_tiles + 31) / 32) * 4, &ev, &timed, &slice)) return rc;
        k.tickets = slice;
      }
      e = sfft::launch_regtile64p(k, ib, ob, !burst_off, stream);
      if (timed) { (void)hipEventRecord(ev.e1, stream); tile_order_timed(a, plan, ev); }
    } else if (c.mixedp) {   // one workgroup per CU, pairs of workgroups on adjacent tiles (kernel_regtile_mixedp.h)
      const int ncu = cu_count(a->device);
      const int slots = std::max(2, ncu / 2 * 2);
      k.tpw = std::max(1, (k.n_tiles + slots - 1) / slots);
      k.n_wg = 2 * ((k.n_tiles + 2 * k.tpw - 1) / (2 * k.tpw));
      const int64_t n = a->n_fft;
      Plan::OrderPending ev{}; bool timed = false;
      if (mixedp_tickets(a, plan, k.n_tiles, k.n_wg)) {      // round 5: dynamic tile order, as at 4096
        unsigned* slice = nullptr;
        if (int rc = ticket_launch_begin(a, plan, stream, 0, sfft::kTkSliceWords, ((size_t)sfft::kTkClaim + (size_t)(k.n_tiles + 31) / 32) * 4, &ev, &timed, &slice)) return rc;
        k.tickets = slice;
      }
      e = n == 3000 ? sfft::launch_regtile_mixedp<60, 50>(k, stream) : n == 2560 ? sfft::launch_regtile_mixedp<64, 40>(k, stream)
        : n == 2400 ? sfft::launch_regtile_mixedp<60, 40>(k, stream) : n == 3072 ? sfft::launch_regtile_mixedp<64, 48>(k, stream)
        : n == 3600 ? sfft::launch_regtile_mixedp<60, 60>(k, stream) : sfft::launch_regtile_mixedp<64, 60>(k, stream);
      if (timed) { (void)hipEventRecord(ev.e1, stream); tile_order_timed(a, plan, ev); }
    } else {
      e = c.tile->launch(k, ib, ob, c.mode, stream);
    }
  } else {
    sfft::StockhamArgs k{};
    k.v = a->v; k.gate = reinterpret_cast<const float2*>(a->gate); k.mem = reinterpret_cast<const float*>(a->mem); k.out = a->out;
    k.B = (int)a->B; k.N_in = (int)std::min<int64_t>(a->N_in, (int64_t)1 << 30); k.N = (int)a->n_fft; k.D = (int)a->D;
    k.G = (int)a->G_tot; k.d_g = (int)(a->D / a->G_tot); k.F = (int)(a->n_fft / 2 + 1);
    k.P = c.P; k.S = c.S; k.solo = c.solo; k.groups_per_batch = (c.S + c.P - 1) / c.P;
    k.in_bf16 = a->in_dtype == SPECTRE_BF16; k.out_bf16 = a->out_dtype == SPECTRE_BF16;
    k.v_sb = a->v_sb; k.v_sn = a->v_sn; k.out_sb = a->out_sb; k.out_sn = a->out_sn;
    const std::vector<int>& rad = plan->bluestein ? plan->radix_m : plan->radix_n;
    k.L = (int)(plan->bluestein ? plan->m : a->n_fft);
    k.n_pass = (int)rad.size();
    if (k.n_pass > sfft::kMaxPasses) return fail(SPECTRE_E_UNSUPPORTED, "too many Stockham passes (%d)", k.n_pass);
    for (int i = 0; i < k.n_pass; ++i) k.radix_packed[i / 8] |= (unsigned long long)rad[(size_t)i] << (8 * (i % 8));
    k.tw = plan->bluestein ? plan->tw_m : plan->tw_n;
    k.bluestein = plan->bluestein ? 1 : 0;
    k.chirp = plan->chirp; k.bhat = plan->bhat;
    k.conj_gate = conj_gate ? 1 : 0;
    const size_t lds = (size_t)k.L * k.P * sizeof(float2);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(sfft::spectre_mix_stockham),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) {
      // enough threads to hold every pass in registers, and at least one per 8 points for the pointwise loops
      int64_t need = ((int64_t)k.L * k.P + 7) / 8;
      for (int i = 0; i < k.n_pass; ++i) {
        const int r = rad[(size_t)i];
        need = std::max<int64_t>(need, (((int64_t)k.L / r) * k.P + sfft::stockham_kmax(r) - 1) / sfft::stockham_kmax(r));
      }
      const int threads = (int)std::min<int64_t>(sfft::kStockhamMaxThreads, std::max<int64_t>(64, (need + 63) / 64 * 64));
      hipLaunchKernelGGL(sfft::spectre_mix_stockham, dim3((unsigned)(a->B * k.groups_per_batch)), dim3(threads), lds, stream, k);
      e = hipGetLastError();
    }
  }
  if (e != hipSuccess) return fail(SPECTRE_E_HIP, "kernel launch failed: %s", hipGetErrorString(e));
  return SPECTRE_OK;
}

int prepare(const SpectreMixArgs* a, Plan** plan, Choice* c) {
  int rc = validate(a);
  if (rc) return rc;
  rc = get_plan(a->device, a->n_fft, plan, reinterpret_cast<hipStream_t>(a->stream));
  if (rc) return rc;
  return choose(a, *plan, c);
}

}  // namespace

extern "C" {

int spectre_version(void) { return SPECTRE_ABI_VERSION; }

int spectre_probe_copy(const SpectreProbeArgs* a, int warmup, int iters, float* ms_per_launch) {   // copy_probe.hip: measurement only
  const char* why = "";
  const int rc = sfft::probe_copy(a, warmup, iters, ms_per_launch, &why);
  return rc == SPECTRE_OK ? SPECTRE_OK : fail(rc, "spectre_probe_copy: %s", why);
}

int spectre_wavelet_refine(const SpectreWaveletArgs* a) {                     // wavelet.hip
  const char* why = "";
  const int rc = sfft::wavelet_refine(a, &why);
  return rc == SPECTRE_OK ? SPECTRE_OK : fail(rc, "spectre_wavelet_refine: %s", why);
}

int spectre_wavelet_gate_grad(const SpectreWaveletGradArgs* a) {
  const char* why = "";
  const int rc = sfft::wavelet_gate_grad(a, &why);
  return rc == SPECTRE_OK ? SPECTRE_OK : fail(rc, "spectre_wavelet_gate_grad: %s", why);
}

const char* spectre_last_error(void) { return g_err.c_str(); }

int spectre_plan_create(int device, int64_t n_fft) {
  if (n_fft < 1) return fail(SPECTRE_E_INVALID, "n_fft must be >= 1");
  DeviceGuard g(device);
  if (!g.ok) return fail(SPECTRE_E_HIP, "cannot select device %d", device);
  Plan* p = nullptr;
  return get_plan(device, n_fft, &p);
}

int spectre_plan_destroy(int device, int64_t n_fft) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_plans.find({device, n_fft});
  if (it == g_plans.end()) return fail(SPECTRE_E_INVALID, "no plan for device %d n_fft %lld", device, (long long)n_fft);
  g_retired[{device, n_fft}] = std::move(it->second);   // out of service, not freed (see g_retired): safe against launches in flight
  g_plans.erase(it);
  return SPECTRE_OK;
}

int spectre_plan_set_tile_order(int device, int64_t n_fft, int order) {
  if (order < SPECTRE_ORDER_AUTO || order > SPECTRE_ORDER_AUTO_PAIR) return fail(SPECTRE_E_INVALID, "bad tile order %d", order);
  if (n_fft < 1) return fail(SPECTRE_E_INVALID, "n_fft must be >= 1");
  DeviceGuard g(device);
  if (!g.ok) return fail(SPECTRE_E_HIP, "cannot select device %d", device);
  Plan* p = nullptr;
  if (int rc = get_plan(device, n_fft, &p)) return rc;
  std::lock_guard<std::mutex> lk(p->order_mu);
  p->order_policy.store(order, std::memory_order_relaxed);
  // start over: what has been measured belongs to the old policy (events of launches still in flight are destroyed, which HIP allows)
  RelaxedCapture rc;
  for (auto& en : p->orders) order_entry_drop(en);
  p->orders.clear();
  p->explore_spent = 0;
  return SPECTRE_OK;
}

int spectre_plan_get_tile_order(int device, int64_t n_fft, int* order) {
  if (!order) return fail(SPECTRE_E_INVALID, "order is NULL");
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_plans.find({device, n_fft});
  if (it == g_plans.end()) return fail(SPECTRE_E_INVALID, "no plan for device %d n_fft %lld", device, (long long)n_fft);
  *order = tile_order_policy(it->second.get());
  return SPECTRE_OK;
}

int spectre_plans_release_retired(int device) {
  // Frees what spectre_plan_destroy only retired.  The caller vouches that no launch that used those plans is still in flight and no
  // other thread is inside a spectre_* call for them (typically: right after a device synchronisation) — exactly the knowledge the
  // library does not have, which is why spectre_plan_destroy itself never frees.
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  for (auto it = g_retired.begin(); it != g_retired.end();) {
    if (device < 0 || it->first.first == device) { it = g_retired.erase(it); ++n; }
    else ++it;
  }
  return n;
}

int spectre_mix_fwd(const SpectreMixArgs* a) {
  if (!a) return fail(SPECTRE_E_INVALID, "args is NULL");
  DeviceGuard g(a->device);
  if (!g.ok) return fail(SPECTRE_E_HIP, "cannot select device %d", a->device);
  Plan* plan = nullptr;
  Choice c;
  int rc = prepare(a, &plan, &c);
  if (rc) return rc;
  return launch(a, plan, c);
}

int spectre_mix_describe(const SpectreMixArgs* a, char* buf, size_t cap) {
  if (!a || !buf || cap == 0) return fail(SPECTRE_E_INVALID, "bad describe arguments");
  DeviceGuard g(a->device);
  if (!g.ok) return fail(SPECTRE_E_HIP, "cannot select device %d", a->device);
  Plan* plan = nullptr;
  Choice c;
  int rc = prepare(a, &plan, &c);
  if (rc) return rc;
  const char* in = a->in_dtype == SPECTRE_BF16 ? "bf16" : "f32";
  const char* out = a->out_dtype == SPECTRE_BF16 ? "bf16" : "f32";
  if (c.regtile) {
    snprintf(buf, cap, "regtile%s %dx%d in=%s out=%s mode=%d tiles=%lld", c.wide ? "-wide" : c.pipelined ? "-pipelined" : c.mixedp ? "-mixed-pipelined" : c.tile->tile_ch == 4 ? "-quad" : c.tile->tile_ch == 8 ? "-long" : c.tile->mixed ? "-mixed" : "", c.RF, c.RS,
             in, out, c.mode, (long long)(a->B * ((a->D + (c.wide ? 32 : c.tile->tile_ch) - 1) / (c.wide ? 32 : c.tile->tile_ch))));
  } else {
    std::string r;
    const std::vector<int>& rad = plan->bluestein ? plan->radix_m : plan->radix_n;
    for (size_t i = 0; i < rad.size(); ++i) r += (i ? "," : "") + std::to_string(rad[i]);
    snprintf(buf, cap, "stockham P=%d solo=%d L=%lld radices=%s bluestein=%d in=%s out=%s (regtile: %s)", c.P, c.solo,
             (long long)(plan->bluestein ? plan->m : a->n_fft), r.c_str(), plan->bluestein ? 1 : 0, in, out,
             c.why_not_regtile[0] ? c.why_not_regtile : "not selected");
  }
  if (c.regtile && c.pipelined && !c.wide) {         // the tile order of the persistent 4096 kernel (as launch() decides it)
    const bool ib = a->in_dtype == SPECTRE_BF16, ob = a->out_dtype == SPECTRE_BF16;
    const int gang = (ib || ob) ? 4 : 2, ncu = cu_count(a->device), n_tiles = (int)(a->B * ((a->D + 15) / 16));
    const int slots = std::max(gang, ncu / gang * gang), tpw = std::max(1, (n_tiles + slots - 1) / slots);
    const int n_wg = gang * ((n_tiles + gang * tpw - 1) / (gang * tpw));
    const size_t l = strlen(buf);
    snprintf(buf + l, cap - l, " order=%s", p64_tickets(a, plan, n_tiles, n_wg, gang) ? tile_order_name(a, plan) : "static");
  }
  if (c.regtile && c.mixedp) {
    const int ncu = cu_count(a->device), n_tiles = (int)(a->B * ((a->D + 15) / 16));
    const int slots = std::max(2, ncu / 2 * 2), tpw = std::max(1, (n_tiles + slots - 1) / slots), n_wg = 2 * ((n_tiles + 2 * tpw - 1) / (2 * tpw));
    const size_t l = strlen(buf);
    snprintf(buf + l, cap - l, " order=%s", mixedp_tickets(a, plan, n_tiles, n_wg) ? tile_order_name(a, plan) : "static");
  }
  const std::string ov = tuning_overrides();
  if (!ov.empty()) { const size_t l = strlen(buf); snprintf(buf + l, cap - l, " [tuning: %s]", ov.c_str()); }
  return SPECTRE_OK;
}

int spectre_mix_time(const SpectreMixArgs* a, int warmup, int iters, float* ms_per_launch) {
  if (!a || !ms_per_launch || iters < 1 || warmup < 0) return fail(SPECTRE_E_INVALID, "bad timing arguments");
  DeviceGuard g(a->device);
  if (!g.ok) return fail(SPECTRE_E_HIP, "cannot select device %d", a->device);
  Plan* plan = nullptr;
  Choice c;
  int rc = prepare(a, &plan, &c);
  if (rc) return rc;
  hipStream_t stream = reinterpret_cast<hipStream_t>(a->stream);
  for (int i = 0; i < warmup; ++i)
    if ((rc = launch(a, plan, c))) return rc;
  // the warm-up launches have been ISSUED, not run: let them finish, so that a tile order that is still being measured (choose_tile_order:
  // launches 25-40 on a tensor pair) is settled by what they measured before the timed launches start, instead of by whoever calls next
  if (warmup > 0 && hipStreamSynchronize(stream) != hipSuccess) return fail(SPECTRE_E_HIP, "hipStreamSynchronize failed");
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(SPECTRE_E_HIP, "hipEventCreate failed");
  hipError_t e = hipEventRecord(e0, stream);
  for (int i = 0; i < iters && e == hipSuccess; ++i)
    if ((rc = launch(a, plan, c))) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
  if (e == hipSuccess) e = hipEventRecord(e1, stream);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess) return fail(SPECTRE_E_HIP, "timing failed: %s", hipGetErrorString(e));
  *ms_per_launch = ms / (float)iters;
  return SPECTRE_OK;
}

// partial sums of the gate gradient: Stockham path (B, G, n_fft) spectrum sums; register-tile path up to 8 partial half spectra per (batch, group)
static int64_t dgate_partials_bytes(int64_t B, int64_t n_fft, int64_t G_tot) {
  return B * G_tot * std::max<int64_t>(n_fft, 8 * (n_fft / 2 + 1)) * (int64_t)sizeof(float2);
}
// Transform length the LDS Stockham gate gradient would work on (n_fft, or the Bluestein convolution length), and whether two slots of it
// still fit the LDS next to each other; if not, the two-pass form needs 2 x (batch chunk) x (F, D) complex spectra of scratch.
static bool dgate_may_need_two_pass(int64_t n_fft) {
  std::vector<int> rad;
  int64_t L = n_fft;
  if (!factorize(n_fft, rad)) { L = 1; while (L < 2 * n_fft - 1) L <<= 1; }
  return 2 * L * 8 > (int64_t)kLdsBytes && !(find_tile_size(n_fft) && find_tile_size(n_fft)->grad);
}
constexpr int64_t kTwoPassTarget = (int64_t)256 << 20;   // bytes of spectra per pass and tensor the query asks for (more is used if given)

int64_t spectre_mix_bwd_workspace_bytes(int64_t B, int64_t n_fft, int64_t D, int64_t G_tot) {
  if (B < 0 || n_fft < 1 || G_tot < 1 || D < 1) return 0;
  int64_t bytes = dgate_partials_bytes(B, n_fft, G_tot);
  if (dgate_may_need_two_pass(n_fft)) {
    const int64_t per_b = (n_fft / 2 + 1) * D * (int64_t)sizeof(float2);
    const int64_t Bc = std::max<int64_t>(1, std::min<int64_t>(B, kTwoPassTarget / per_b));
    bytes = std::max<int64_t>(bytes, 2 * Bc * per_b);
  }
  return bytes;
}

int spectre_mix_bwd(const SpectreMixBwdArgs* a) {
  if (!a) return fail(SPECTRE_E_INVALID, "args is NULL");
  if (a->B < 0 || a->N_in < 1 || a->n_fft < 1 || a->D < 1 || a->G_tot < 1 || a->D % a->G_tot)
    return fail(SPECTRE_E_INVALID, "bad sizes");
  if (a->io_dtype != SPECTRE_F32 && a->io_dtype != SPECTRE_BF16) return fail(SPECTRE_E_UNSUPPORTED, "bad io_dtype");
  if (a->B == 0) return SPECTRE_OK;
  if (!a->v || !a->gate || !a->dout) return fail(SPECTRE_E_INVALID, "v, gate and dout must be non-NULL device pointers");
  if (a->dgate && !a->workspace) return fail(SPECTRE_E_INVALID, "dgate requested without a workspace");
  if (a->dgate && a->workspace_bytes < spectre_mix_bwd_workspace_bytes(a->B, a->n_fft, a->D, a->G_tot))
    return fail(SPECTRE_E_INVALID, "workspace of %lld bytes, spectre_mix_bwd_workspace_bytes() asks for %lld", (long long)a->workspace_bytes,
                (long long)spectre_mix_bwd_workspace_bytes(a->B, a->n_fft, a->D, a->G_tot));
  DeviceGuard g(a->device);
  if (!g.ok) return fail(SPECTRE_E_HIP, "cannot select device %d", a->device);
  hipStream_t stream = reinterpret_cast<hipStream_t>(a->stream);
  const int64_t n_out = std::min(a->N_in, a->n_fft);
  const int es = a->io_dtype == SPECTRE_BF16 ? 2 : 4;
  int rc;
  if (a->dv) {
    // dV = mix(dOut, conj(gate)): rows [0, n_out); rows n_out..N_in (input truncated by rfft's n=) get zero gradient
    SpectreMixArgs f{};
    f.v = a->dout; f.gate = a->gate; f.mem = nullptr; f.out = a->dv;
    f.B = a->B; f.N_in = n_out; f.n_fft = a->n_fft; f.D = a->D; f.G_tot = a->G_tot;
    f.v_sb = a->dout_sb; f.v_sn = a->dout_sn; f.out_sb = a->dv_sb; f.out_sn = a->dv_sn;
    f.in_dtype = a->io_dtype; f.out_dtype = a->io_dtype; f.algo = SPECTRE_ALGO_AUTO; f.device = a->device; f.stream = a->stream;
    Plan* plan = nullptr;
    Choice c;
    if ((rc = prepare(&f, &plan, &c))) return rc;
    if ((rc = launch(&f, plan, c, /*conj_gate=*/true))) return rc;
    if (a->N_in > n_out) {
      for (int64_t b = 0; b < a->B; ++b) {
        char* p = reinterpret_cast<char*>(a->dv) + (b * a->dv_sb + n_out * a->dv_sn) * es;
        hipError_t e = hipMemset2DAsync(p, (size_t)a->dv_sn * es, 0, (size_t)a->D * es, (size_t)(a->N_in - n_out), stream);
        if (e != hipSuccess) return fail(SPECTRE_E_HIP, "hipMemset2DAsync: %s", hipGetErrorString(e));
      }
    }
  }
  if (a->dgate) {
    Plan* plan = nullptr;
    if ((rc = get_plan(a->device, a->n_fft, &plan))) return rc;
    const int64_t n = a->n_fft, D = a->D, d_g = D / a->G_tot;
    const TileSize* ts = find_tile_size(n);
    static const bool force_stockham = [] { const char* e = tuning_env("SPECTRE_GATE_GRAD"); return e && !strcmp(e, "stockham"); }();
    // (32-bit buffer offsets: a tile's last row has to lie below 2^31 bytes from its first; wider views take the Stockham path)
    if (ts && ts->grad && !force_stockham && std::max<int64_t>(n, 128) * a->v_sn * 4 < ((int64_t)1 << 31) &&
        std::max<int64_t>(n, 128) * a->dout_sn * 4 < ((int64_t)1 << 31) && a->B * a->G_tot * 8 < ((int64_t)1 << 31)) {
      // register-tile gate gradient (kernel_regtile_grad.h / kernel_regtile_mixed_grad.h): 8-channel tiles, S workgroups
      // per (batch, group)
      sfft::GateGradArgs k{};
      k.v = a->v; k.dout = a->dout; k.part = reinterpret_cast<float2*>(a->workspace); k.tw = plan->tw_n;
      k.B = (int)a->B; k.N_in = (int)std::min<int64_t>(a->N_in, (int64_t)1 << 30); k.D = (int)D; k.G = (int)a->G_tot;
      k.d_g = (int)d_g; k.F = (int)(n / 2 + 1);
      const int gch = ts->tile_ch == 8 ? 4 : 8;       // channels per gradient tile (one per lane p)
      k.T = (int)((d_g + gch - 1) / gch);
      int S = std::min(k.T, gch == 4 ? 8 : 4);        // the tiles of one 128-byte line go to neighbouring workgroups
      while (a->B * a->G_tot * S < 1024 && 2 * S <= std::min(k.T, 8)) S *= 2;
      k.S = S; k.n_wg = (int)(a->B * a->G_tot * S);
      k.prefetch = kDgatePrefetch;
      k.grid = cu_count(a->device) / S * S;           // (persistent form: whole (batch, group) gangs)
      if (const char* e = tuning_env("SPECTRE_DGATE_GRID")) k.grid = atoi(e);
      if (const char* e = tuning_env("SPECTRE_DGATE_PREFETCH")) k.prefetch = atoi(e);
      k.v_sb = a->v_sb; k.v_sn = a->v_sn; k.dout_sb = a->dout_sb; k.dout_sn = a->dout_sn;
      const bool bf = a->io_dtype == SPECTRE_BF16, general = (a->N_in < n) || (d_g % gch != 0);
      hipError_t e = ts->grad(k, bf, general, stream);
      if (e != hipSuccess) return fail(SPECTRE_E_HIP, "gate-gradient launch failed: %s", hipGetErrorString(e));
      const int64_t total = a->B * a->G_tot * (n / 2 + 1);
      hipLaunchKernelGGL(sfft::spectre_gate_grad_regtile_finish<0>, dim3((unsigned)std::min<int64_t>(4096, (total + 255) / 256)), dim3(256), 0,
                         stream, reinterpret_cast<const float2*>(a->workspace), reinterpret_cast<float2*>(a->dgate), S, (int)(n / 2 + 1),
                         (int)n, (long long)total);
      if ((e = hipGetLastError()) != hipSuccess) return fail(SPECTRE_E_HIP, "gate-gradient finish failed: %s", hipGetErrorString(e));
      return SPECTRE_OK;
    }
    const int64_t L = plan->bluestein ? plan->m : n;
    const std::vector<int>& rad = plan->bluestein ? plan->radix_m : plan->radix_n;
    const int solo = (d_g % 2) ? 1 : 0;
    const int64_t S = solo ? D : D / 2, Sg = solo ? d_g : d_g / 2;
    // P slots of v + P slots of dOut share the LDS buffer; P must divide the slots of one gate group
    int64_t P = 0;
    for (int64_t cand = std::min<int64_t>(8, Sg); cand >= 1; --cand) {
      if (Sg % cand) continue;
      if (2 * cand * L * 8 > (int64_t)kLdsBytes) continue;
      bool ok = true;
      for (int r : rad) ok = ok && ((L / r) * 2 * cand <= (int64_t)sfft::kStockhamMaxThreads * sfft::stockham_kmax(r));
      if (ok) { P = cand; break; }
    }
    if (P < 1) {
      // two LDS slots do not fit (n_fft = 12288, 16384, long Bluestein lengths): two-pass fallback — spectra of V and dOut with
      // the half-spectrum kernel into the CALLER'S workspace (as many batch elements per pass as it holds), then a reduction over
      // each group's channels (kernel_gate_grad_twopass.h).  Never refuse a length the forward accepts; never allocate.
      const int64_t F = n / 2 + 1, per_b = F * D * (int64_t)sizeof(float2);
      const int64_t Bc = std::min<int64_t>(std::min<int64_t>(a->B, 65535), a->workspace_bytes / (2 * per_b));
      if (Bc < 1) return fail(SPECTRE_E_INVALID, "gate gradient (two-pass): the workspace holds %lld bytes, one batch element needs %lld",
                              (long long)a->workspace_bytes, (long long)(2 * per_b));
      if (F >= ((int64_t)1 << 31) || a->G_tot >= 65536) return fail(SPECTRE_E_UNSUPPORTED, "gate gradient: grid too large");
      float2* X = reinterpret_cast<float2*>(a->workspace);
      float2* R = X + Bc * F * D;
      hipError_t e = hipSuccess;
      int rc2 = SPECTRE_OK;
      for (int64_t b0 = 0; b0 < a->B && rc2 == SPECTRE_OK; b0 += Bc) {
        const int64_t bc = std::min<int64_t>(Bc, a->B - b0);
        SpectreRfftArgs r{};
        r.B = bc; r.n_fft = n; r.D = D; r.in_dtype = a->io_dtype; r.device = a->device; r.stream = a->stream;
        r.v = reinterpret_cast<const char*>(a->v) + b0 * a->v_sb * es; r.spec = X; r.N_in = a->N_in; r.v_sb = a->v_sb; r.v_sn = a->v_sn;
        rc2 = spectre_rfft_fwd(&r);
        if (rc2 == SPECTRE_OK) {
          r.v = reinterpret_cast<const char*>(a->dout) + b0 * a->dout_sb * es; r.spec = R; r.N_in = n_out; r.v_sb = a->dout_sb; r.v_sn = a->dout_sn;
          rc2 = spectre_rfft_fwd(&r);
        }
        if (rc2 == SPECTRE_OK) {
          hipLaunchKernelGGL(sfft::spectre_gate_grad_reduce, dim3((unsigned)F, (unsigned)a->G_tot, (unsigned)bc), dim3(64), 0, stream, X, R,
                             reinterpret_cast<float2*>(a->dgate) + b0 * a->G_tot * F, (int)F, (int)D, (int)a->G_tot, (int)d_g, (int)n);
          if ((e = hipGetLastError()) != hipSuccess) rc2 = fail(SPECTRE_E_HIP, "gate-gradient reduce launch failed: %s", hipGetErrorString(e));
        }
      }
      return rc2;
    }
    sfft::StockhamArgs k{};
    k.v = a->v; k.gate = reinterpret_cast<const float2*>(a->gate); k.mem = nullptr; k.out = nullptr;
    k.B = (int)a->B; k.N_in = (int)std::min<int64_t>(a->N_in, (int64_t)1 << 30); k.N = (int)n; k.D = (int)D; k.G = (int)a->G_tot;
    k.d_g = (int)d_g; k.F = (int)(n / 2 + 1);
    k.P = (int)P; k.S = (int)S; k.solo = solo; k.groups_per_batch = (int)((S + P - 1) / P);
    k.in_bf16 = a->io_dtype == SPECTRE_BF16; k.out_bf16 = 0;
    k.v_sb = a->v_sb; k.v_sn = a->v_sn;
    k.L = (int)L; k.n_pass = (int)rad.size();
    for (int i = 0; i < k.n_pass; ++i) k.radix_packed[i / 8] |= (unsigned long long)rad[(size_t)i] << (8 * (i % 8));
    k.tw = plan->bluestein ? plan->tw_m : plan->tw_n;
    k.bluestein = plan->bluestein ? 1 : 0; k.chirp = plan->chirp; k.bhat = plan->bhat;
    k.dout = a->dout; k.dout_sb = a->dout_sb; k.dout_sn = a->dout_sn; k.ws = reinterpret_cast<float2*>(a->workspace);
    hipError_t e = hipMemsetAsync(a->workspace, 0, (size_t)dgate_partials_bytes(a->B, n, a->G_tot), stream);
    if (e != hipSuccess) return fail(SPECTRE_E_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    const size_t lds = (size_t)2 * k.L * k.P * sizeof(float2);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(sfft::spectre_gate_grad_stockham),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(SPECTRE_E_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(sfft::spectre_gate_grad_stockham, dim3((unsigned)(a->B * k.groups_per_batch)), dim3(sfft::kStockhamMaxThreads),
                       lds, stream, k);
    const int64_t total = a->B * a->G_tot * (n / 2 + 1);
    hipLaunchKernelGGL(sfft::spectre_gate_grad_finish, dim3((unsigned)std::min<int64_t>(4096, (total + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(a->workspace), reinterpret_cast<float2*>(a->dgate), (int)(a->B * a->G_tot), (int)n,
                       (int)(n / 2 + 1));
    if ((e = hipGetLastError()) != hipSuccess) return fail(SPECTRE_E_HIP, "gate-gradient launch failed: %s", hipGetErrorString(e));
  }
  return SPECTRE_OK;
}

int spectre_gate_fwd(const SpectreGateArgs* a) {
  if (!a) return fail(SPECTRE_E_INVALID, "args is NULL");
  if (a->B < 0 || a->G < 1 || a->K < 1 || a->F < 1) return fail(SPECTRE_E_INVALID, "bad sizes");
  if (a->phase_sb != 0 && a->phase_sb != a->F) return fail(SPECTRE_E_INVALID, "phase_sb must be 0 or F");
  if (a->B == 0) return SPECTRE_OK;
  if (!a->anchors || !a->bias || !a->gate) return fail(SPECTRE_E_INVALID, "anchors, bias and gate must be non-NULL device pointers");
  if (a->B * a->G * a->F >= ((int64_t)1 << 40) || a->K >= ((int64_t)1 << 24) || a->F >= ((int64_t)1 << 24))
    return fail(SPECTRE_E_UNSUPPORTED, "gate tensor too large");
  DeviceGuard g(a->device);
  if (!g.ok) return fail(SPECTRE_E_HIP, "cannot select device %d", a->device);
  sfft::GateArgs k{};
  k.anchors = reinterpret_cast<const float2*>(a->anchors); k.bias = reinterpret_cast<const float*>(a->bias);
  k.phase = reinterpret_cast<const float2*>(a->phase); k.gate = reinterpret_cast<float2*>(a->gate);
  k.B = (int)a->B; k.G = (int)a->G; k.K = (int)a->K; k.F = (int)a->F; k.phase_sb = a->phase_sb; k.eps = a->eps;
  k.decode_m = 0; k.decode_n = 1;
  const int64_t total = a->B * a->G * a->F;
  hipLaunchKernelGGL(sfft::spectre_gate_producer, dim3((unsigned)std::min<int64_t>(8192, (total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(a->stream), k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SPECTRE_E_HIP, "gate producer launch failed: %s", hipGetErrorString(e));
  return SPECTRE_OK;
}

int spectre_gate_bwd(const SpectreGateBwdArgs* a) {
  if (!a) return fail(SPECTRE_E_INVALID, "args is NULL");
  if (a->B < 0 || a->G < 1 || a->K < 1 || a->F < 1) return fail(SPECTRE_E_INVALID, "bad sizes");
  if (a->phase_sb != 0 && a->phase_sb != a->F) return fail(SPECTRE_E_INVALID, "phase_sb must be 0 or F");
  if (a->B == 0) return SPECTRE_OK;
  if (!a->anchors || !a->bias || !a->dgate || !a->workspace || !a->danchors || !a->dbias)
    return fail(SPECTRE_E_INVALID, "anchors, bias, dgate, workspace, danchors and dbias must be non-NULL device pointers");
  if (a->dphase && !a->phase) return fail(SPECTRE_E_INVALID, "dphase requested without a phase");
  if (a->B * a->G * a->F >= ((int64_t)1 << 40) || a->K >= 65536 || a->F >= ((int64_t)1 << 24) || 2 * a->G >= 65536 || a->B >= 65536)
    return fail(SPECTRE_E_UNSUPPORTED, "gate tensor too large");
  DeviceGuard g(a->device);
  if (!g.ok) return fail(SPECTRE_E_HIP, "cannot select device %d", a->device);
  hipStream_t stream = reinterpret_cast<hipStream_t>(a->stream);
  sfft::GateBwdArgs k{};
  k.anchors = reinterpret_cast<const float2*>(a->anchors); k.bias = reinterpret_cast<const float*>(a->bias);
  k.phase = reinterpret_cast<const float2*>(a->phase); k.dgate = reinterpret_cast<const float2*>(a->dgate);
  k.dr = reinterpret_cast<float2*>(a->workspace); k.danchors = reinterpret_cast<float2*>(a->danchors);
  k.dbias = reinterpret_cast<float*>(a->dbias); k.dphase = reinterpret_cast<float2*>(a->dphase);
  k.B = (int)a->B; k.G = (int)a->G; k.K = (int)a->K; k.F = (int)a->F; k.phase_sb = a->phase_sb; k.eps = a->eps;
  hipLaunchKernelGGL(sfft::spectre_gate_bwd_elem, dim3((unsigned)((a->G * a->F + 255) / 256)), dim3(256), 0, stream, k);
  hipLaunchKernelGGL(sfft::spectre_gate_bwd_anchors, dim3((unsigned)a->K, (unsigned)(2 * a->G), (unsigned)a->B), dim3(64), 0, stream, k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SPECTRE_E_HIP, "gate backward launch failed: %s", hipGetErrorString(e));
  return SPECTRE_OK;
}

int spectre_rfft_fwd(const SpectreRfftArgs* a) {
  if (!a) return fail(SPECTRE_E_INVALID, "args is NULL");
  if (a->B < 0 || a->N_in < 1 || a->n_fft < 1 || a->D < 1) return fail(SPECTRE_E_INVALID, "bad sizes");
  if (a->in_dtype != SPECTRE_F32 && a->in_dtype != SPECTRE_BF16) return fail(SPECTRE_E_UNSUPPORTED, "bad in_dtype");
  if (a->B == 0) return SPECTRE_OK;
  if (!a->v || !a->spec) return fail(SPECTRE_E_INVALID, "v and spec must be non-NULL device pointers");
  DeviceGuard g(a->device);
  if (!g.ok) return fail(SPECTRE_E_HIP, "cannot select device %d", a->device);
  // slot geometry of the general LDS kernel: one gate channel per batch element stands in for "no filter"
  SpectreMixArgs f{};
  f.v = a->v; f.gate = a->spec; f.mem = nullptr; f.out = a->spec;
  f.B = a->B; f.N_in = a->N_in; f.n_fft = a->n_fft; f.D = a->D; f.G_tot = (a->D % 2) ? a->D : 1;
  f.v_sb = a->v_sb; f.v_sn = a->v_sn; f.out_sb = 0; f.out_sn = a->D;
  f.in_dtype = a->in_dtype; f.out_dtype = SPECTRE_F32; f.algo = SPECTRE_ALGO_STOCKHAM; f.device = a->device; f.stream = a->stream;
  Plan* plan = nullptr;
  Choice c;
  int rc = prepare(&f, &plan, &c);
  if (rc) return rc;
  sfft::StockhamArgs k{};
  k.v = a->v; k.out = a->spec;
  k.B = (int)a->B; k.N_in = (int)std::min<int64_t>(a->N_in, (int64_t)1 << 30); k.N = (int)a->n_fft; k.D = (int)a->D;
  k.G = 1; k.d_g = (int)a->D; k.F = (int)(a->n_fft / 2 + 1);
  k.P = c.P; k.S = c.S; k.solo = c.solo; k.groups_per_batch = (c.S + c.P - 1) / c.P;
  k.in_bf16 = a->in_dtype == SPECTRE_BF16;
  k.v_sb = a->v_sb; k.v_sn = a->v_sn;
  const std::vector<int>& rad = plan->bluestein ? plan->radix_m : plan->radix_n;
  k.L = (int)(plan->bluestein ? plan->m : a->n_fft);
  k.n_pass = (int)rad.size();
  if (k.n_pass > sfft::kMaxPasses) return fail(SPECTRE_E_UNSUPPORTED, "too many Stockham passes (%d)", k.n_pass);
  for (int i = 0; i < k.n_pass; ++i) k.radix_packed[i / 8] |= (unsigned long long)rad[(size_t)i] << (8 * (i % 8));
  k.tw = plan->bluestein ? plan->tw_m : plan->tw_n;
  k.bluestein = plan->bluestein ? 1 : 0; k.chirp = plan->chirp; k.bhat = plan->bhat;
  const size_t lds = (size_t)k.L * k.P * sizeof(float2);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sfft::spectre_rfft_stockham),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return fail(SPECTRE_E_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  int64_t need = ((int64_t)k.L * k.P + 7) / 8;
  for (int i = 0; i < k.n_pass; ++i) {
    const int r = rad[(size_t)i];
    need = std::max<int64_t>(need, (((int64_t)k.L / r) * k.P + sfft::stockham_kmax(r) - 1) / sfft::stockham_kmax(r));
  }
  const int threads = (int)std::min<int64_t>(sfft::kStockhamMaxThreads, std::max<int64_t>(64, (need + 63) / 64 * 64));
  hipLaunchKernelGGL(sfft::spectre_rfft_stockham, dim3((unsigned)(a->B * k.groups_per_batch)), dim3(threads), lds,
                     reinterpret_cast<hipStream_t>(a->stream), k);
  if ((e = hipGetLastError()) != hipSuccess) return fail(SPECTRE_E_HIP, "rfft launch failed: %s", hipGetErrorString(e));
  return SPECTRE_OK;
}

int64_t spectre_decode_workspace_bytes(int64_t n_fft, int64_t d) {
  if (n_fft < 1 || d < 1) return 0;
  const int64_t F = n_fft / 2 + 1;
  return ((F + sfft::kDecodeChunk - 1) / sfft::kDecodeChunk) * d * (int64_t)sizeof(float);
}

int spectre_decode_step(const SpectreDecodeArgs* a) {
  if (!a) return fail(SPECTRE_E_INVALID, "args is NULL");
  if (a->n_fft < 1 || a->d < 1 || a->t < 0 || a->t >= ((int64_t)1 << 24)) return fail(SPECTRE_E_INVALID, "bad sizes (t must be < 2^24)");
  if (!a->prefix || !a->v_old || !a->v_new) return fail(SPECTRE_E_INVALID, "prefix, v_old and v_new must be non-NULL device pointers");
  if (a->gate && (a->G < 1 || a->d % a->G || !a->out || !a->workspace)) return fail(SPECTRE_E_INVALID, "gate given: need G | d, out and workspace");
  if (a->n_fft >= ((int64_t)1 << 24) || a->d >= ((int64_t)1 << 24)) return fail(SPECTRE_E_UNSUPPORTED, "sizes too large");
  DeviceGuard g(a->device);
  if (!g.ok) return fail(SPECTRE_E_HIP, "cannot select device %d", a->device);
  hipStream_t stream = reinterpret_cast<hipStream_t>(a->stream);
  sfft::DecodeArgs k{};
  k.prefix = reinterpret_cast<float2*>(a->prefix); k.v_old = reinterpret_cast<const float*>(a->v_old);
  k.v_new = reinterpret_cast<const float*>(a->v_new); k.gate = reinterpret_cast<const float2*>(a->gate);
  k.partial = reinterpret_cast<float*>(a->workspace);
  k.n = (int)a->n_fft; k.F = (int)(a->n_fft / 2 + 1); k.d = (int)a->d; k.d_g = a->gate ? (int)(a->d / a->G) : (int)a->d;
  k.t = (int)a->t; k.j = (int)(a->t % a->n_fft); k.evict = a->t >= a->n_fft ? 1 : 0; k.chunk = sfft::kDecodeChunk;
  const int chunks = (k.F + k.chunk - 1) / k.chunk;
  hipLaunchKernelGGL(sfft::spectre_decode_step, dim3((unsigned)chunks, (unsigned)((k.d + 63) / 64)), dim3(256), 0, stream, k);
  if (a->gate)
    hipLaunchKernelGGL(sfft::spectre_decode_finish, dim3((unsigned)((k.d + 255) / 256)), dim3(256), 0, stream, k.partial,
                       reinterpret_cast<float*>(a->out), chunks, k.d, k.n, (float*)nullptr, (const float*)nullptr, (float*)nullptr,
                       (const float*)nullptr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SPECTRE_E_HIP, "decode launch failed: %s", hipGetErrorString(e));
  return SPECTRE_OK;
}

int64_t spectre_decode_head_workspace_bytes(int64_t n_fft, int64_t d, int64_t G, int64_t K) {
  if (n_fft < 1 || d < 1 || G < 1 || K < 1) return 0;
  const int64_t F = n_fft / 2 + 1;
  return spectre_decode_workspace_bytes(n_fft, d) + ((2 * G * K * 4 + 15) / 16) * 16 + G * F * 8;   // partials | anchors | gate
}

int spectre_decode_head_step(const SpectreDecodeHeadArgs* a) {
  if (!a) return fail(SPECTRE_E_INVALID, "args is NULL");
  if (a->n_fft < 1 || a->d < 1 || a->G < 1 || a->K < 1 || a->h1 < 1 || a->d % a->G || a->t < 0 || a->t >= ((int64_t)1 << 24))
    return fail(SPECTRE_E_INVALID, "bad sizes (t must be < 2^24)");
  if (!a->prefix || !a->V_buf || !a->Q_buf || !a->sum_q || !a->q_t || !a->v_t || !a->out || !a->workspace || !a->ln_w || !a->ln_b ||
      !a->w1 || !a->b1 || !a->w2 || !a->b2 || !a->modrelu_bias)
    return fail(SPECTRE_E_INVALID, "all pointers must be non-NULL device pointers");
  if ((a->d + a->h1) * 4 > 60 * 1024) return fail(SPECTRE_E_UNSUPPORTED, "d + d_gate too large for the descriptor kernel");
  DeviceGuard g(a->device);
  if (!g.ok) return fail(SPECTRE_E_HIP, "cannot select device %d", a->device);
  hipStream_t stream = reinterpret_cast<hipStream_t>(a->stream);
  const int64_t F = a->n_fft / 2 + 1, j = a->t % a->n_fft;
  const int evict = a->t >= a->n_fft ? 1 : 0;
  char* ws = reinterpret_cast<char*>(a->workspace);
  float* partial = reinterpret_cast<float*>(ws);
  float* anchors = reinterpret_cast<float*>(ws + spectre_decode_workspace_bytes(a->n_fft, a->d));
  float2* gate = reinterpret_cast<float2*>(reinterpret_cast<char*>(anchors) + ((2 * a->G * a->K * 4 + 15) / 16) * 16);
  // 1) running query sum, LayerNorm, gate MLP -> anchors (spectre.py:809-813, :578-580)
  sfft::DecodeMlpArgs m{};
  m.sum_q = reinterpret_cast<float*>(a->sum_q); m.q_t = reinterpret_cast<const float*>(a->q_t);
  m.ln_w = reinterpret_cast<const float*>(a->ln_w); m.ln_b = reinterpret_cast<const float*>(a->ln_b);
  m.w1 = reinterpret_cast<const float*>(a->w1); m.b1 = reinterpret_cast<const float*>(a->b1);
  m.w2 = reinterpret_cast<const float*>(a->w2); m.b2 = reinterpret_cast<const float*>(a->b2);
  m.anchors = anchors; m.d = (int)a->d; m.h1 = (int)a->h1; m.o = (int)(2 * a->G * a->K); m.n = (int)a->n_fft; m.evict = evict;
  m.ln_eps = a->ln_eps;
  hipLaunchKernelGGL(sfft::spectre_decode_mlp, dim3(1), dim3(1024), (size_t)(a->d + a->h1) * 4, stream, m);
  // 2) cubic resample -> modReLU -> decode phase (spectre.py:586-596)
  sfft::GateArgs gk{};
  gk.anchors = reinterpret_cast<const float2*>(anchors); gk.bias = reinterpret_cast<const float*>(a->modrelu_bias); gk.phase = nullptr;
  gk.gate = gate; gk.B = 1; gk.G = (int)a->G; gk.K = (int)a->K; gk.F = (int)F; gk.phase_sb = 0; gk.eps = a->modrelu_eps;
  gk.decode_m = (int)(a->t - j); gk.decode_n = (int)a->n_fft;
  const int64_t total = a->G * F;
  hipLaunchKernelGGL(sfft::spectre_gate_producer, dim3((unsigned)std::min<int64_t>(8192, (total + 255) / 256)), dim3(256), 0, stream, gk);
  // 3) spectrum update + filter + one-row inverse (spectre.py:794-805, :603, :614-655), then the ring writes (:807-810)
  float* v_row = reinterpret_cast<float*>(a->V_buf) + j * a->d;
  float* q_row = reinterpret_cast<float*>(a->Q_buf) + j * a->d;
  sfft::DecodeArgs k{};
  k.prefix = reinterpret_cast<float2*>(a->prefix); k.v_old = v_row; k.v_new = reinterpret_cast<const float*>(a->v_t);
  k.gate = gate; k.partial = partial;
  k.n = (int)a->n_fft; k.F = (int)F; k.d = (int)a->d; k.d_g = (int)(a->d / a->G);
  k.t = (int)a->t; k.j = (int)j; k.evict = evict; k.chunk = sfft::kDecodeChunk;
  const int chunks = (k.F + k.chunk - 1) / k.chunk;
  hipLaunchKernelGGL(sfft::spectre_decode_step, dim3((unsigned)chunks, (unsigned)((k.d + 63) / 64)), dim3(256), 0, stream, k);
  hipLaunchKernelGGL(sfft::spectre_decode_finish, dim3((unsigned)((k.d + 255) / 256)), dim3(256), 0, stream, k.partial,
                     reinterpret_cast<float*>(a->out), chunks, k.d, k.n, v_row, k.v_new, q_row, reinterpret_cast<const float*>(a->q_t));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SPECTRE_E_HIP, "decode launch failed: %s", hipGetErrorString(e));
  return SPECTRE_OK;
}

}  // extern "C"
