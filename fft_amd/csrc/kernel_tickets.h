// kernel_tickets.h — dynamic tile order for the persistent register-tile kernels (round 5): one TICKET per GANG of workgroups (the
// workgroups whose tiles share the 128-byte lines of a row) from ONE chip-wide counter, so that at any time the chip works on a few
// adjacent batch elements instead of on all of them (tools/window_lab.hip: a pure copy in 64-byte tiles runs 40 % faster that way, in
// 32-byte tiles 43 %; PMC, profiles/r05_pmc_static_vs_pair_tickets.txt: the L2's write requests wait half as long for DRAM credits).
// kernel_regtile64p.h carries the same protocol inline (its comment explains the choices); this header is the reusable form.
//
// Memory: a slice of UNCACHED device memory (scalar atomics carry no scope bits: only memory the L2 does not keep is coherent between
// XCDs for them), zeroed on the launch's stream:   [0] counter | [1] tiles claimed so far | [kTkBox + 8 g ...] mailbox of gang g (8 tagged
// slots) | [kTkClaim ...] one claim bit per tile.  LDS: two words of the caller's.
//
// Protocol (wave 0 of every workgroup; every step sits behind a barrier that has waited for lgkmcnt(0), so no step waits for memory):
//   leader (member 0):  draw()  ->  publish() [+ claim]  ->  result()
//   follower:                        ask()    ->  check() [+ claim]  ->  result()
// result() leaves the tile after next (>= 0 a tile index, -1 a PHANTOM tile: run the same instruction stream over empty buffer ranges
// so that the gang stays in step, -2 the end of the stream) in LDS word 0; every wave reads it with handed_over() behind a later barrier.
// A tile is processed by whoever set its claim bit — exactly one workgroup can — so correctness never depends on the members of a gang
// being resident together: a follower whose leader does not publish in time stops following, and every workgroup that runs out of
// tickets calls sweep() until no unclaimed tile is left.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfft {

constexpr int kTkBox = 16, kTkClaim = 2048, kTkSliceWords = 16384;
constexpr unsigned kTkEnd = 0xffffffu;
constexpr int tk_capacity() { return (kTkSliceWords - kTkClaim) * 32; }       // tiles one slice has claim bits for
constexpr int tk_max_gangs() { return (kTkClaim - kTkBox) / 8; }

template <int GANG>
struct GangTickets {
  unsigned* cnt; unsigned* box; unsigned* claim; volatile int* lds;
  unsigned total; int n_tiles, member; bool w0;
  int seq; bool follow;
  unsigned ra, rb, bit; int tile;                 // the state machine's registers (SGPRs: everything here is wave-uniform)

  __device__ __forceinline__ static unsigned tag(int s) { return (unsigned)(s % 255) + 1u; }

  __device__ __forceinline__ void init(unsigned* slice, int gang_index, int member_, int n_tiles_, volatile int* lds_, int tid0) {
    cnt = slice; box = slice + kTkBox + 8 * gang_index; claim = slice + kTkClaim; lds = lds_;
    n_tiles = n_tiles_; member = member_; total = (unsigned)((n_tiles_ + GANG - 1) / GANG);
    w0 = __builtin_amdgcn_readfirstlane(tid0 >> 6) == 0;
    seq = 2; follow = true; ra = 1; rb = 0; bit = 0; tile = -2;
  }
  // the first two tickets of the gang, synchronously (call before anything is in flight; contains __syncthreads)
  __device__ __forceinline__ void first_two(int tid0, int& cur, int& nxt) {
    if (tid0 == 0) {
      for (int sq = 0; sq < 2; ++sq) {
        unsigned t = kTkEnd;
        if (member == 0) {
          t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (t >= total) t = kTkEnd;
          __hip_atomic_store(box + sq, (tag(sq) << 24) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          for (int i = 0; i < (1 << 18); ++i) {
            const unsigned w = __hip_atomic_load(box + sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w >> 24) == tag(sq)) { t = w & 0xffffffu; break; }
            __builtin_amdgcn_s_sleep(8);
          }
        }
        int tl = -2;
        if (t != kTkEnd) {
          const unsigned ti = t * GANG + member;
          tl = -1;
          if (ti < (unsigned)n_tiles && !(__hip_atomic_fetch_or(claim + (ti >> 5), 1u << (ti & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (1u << (ti & 31)))) tl = (int)ti;
        }
        if (tl >= 0) __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds[sq] = tl;
        if (tl == -2) { lds[1] = -2; break; }         // the stream has ended (or the leader is silent): claim nothing behind it
      }
    }
    __syncthreads();
    cur = __builtin_amdgcn_readfirstlane(lds[0]); nxt = __builtin_amdgcn_readfirstlane(lds[1]);
    __syncthreads();
    follow = nxt != -2;
    if (cur == -2) nxt = -2;
    seq = 2;
  }
  // one unclaimed tile, or false when every tile has an owner (call with nothing in flight; contains __syncthreads).  While the stream is
  // running only tiles below counter - 3 * gangs are taken (a gang has at most three tickets on their way, and a ticket stays unclaimed
  // for a moment between the draw and each member's claim); behind the end of the stream, one grace period later, everything counts.
  __device__ __forceinline__ bool sweep(int tid0, int nthreads, int wg_lin, int n_wg, int& cur) {
    bool grace = false;
    for (;;) {
      __syncthreads();
      if (tid0 == 0) { lds[0] = 0x7fffffff; lds[1] = (int)__hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(lds[1]) >= n_tiles) return false;      // every tile has an owner (the usual end of a launch)
      __syncthreads();
      if (tid0 == 0) lds[1] = (int)__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ONE reading for the whole workgroup: every
      __syncthreads();                                                                                    // wave must take the same way out
      const unsigned now = (unsigned)__builtin_amdgcn_readfirstlane(lds[1]);
      __syncthreads();
      const bool all = now >= total && grace;
      const long long old = all ? (long long)total : (long long)now - 3LL * (n_wg / GANG);
      const int lim = (int)(old <= 0 ? 0 : old * GANG > n_tiles ? n_tiles : old * GANG);
      const int n_words = (lim + 31) >> 5, all_words = (n_tiles + 31) >> 5;
      const int sw = n_words ? (int)((long long)wg_lin * n_words / n_wg) : 0;
      if (tid0 == 0) lds[1] = 0;
      __syncthreads();
      for (int w = tid0; w < all_words; w += nthreads) {
        unsigned fr = ~__hip_atomic_load(claim + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (32 * w + 32 > n_tiles) fr &= (1u << (n_tiles - 32 * w)) - 1u;
        if (fr) lds[1] = 1;                         // something is unclaimed somewhere (the usual end of a launch: nothing is)
        if (32 * w + 32 > lim) fr &= 32 * w >= lim ? 0u : (1u << (lim - 32 * w)) - 1u;
        if (fr) atomicMin(const_cast<int*>(lds), 32 * (w >= sw ? w - sw : w - sw + n_words) + (int)__builtin_ctz(fr));
      }
      __syncthreads();
      const int key = __builtin_amdgcn_readfirstlane(lds[0]), any_free = __builtin_amdgcn_readfirstlane(lds[1]);
      __syncthreads();
      if (key == 0x7fffffff) {
        if (all || !any_free) return false;
        for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(127);      // ~ 25 us
        if (now >= total) grace = true;
        continue;
      }
      const int cand = 32 * (((key >> 5) + sw) % n_words) + (key & 31);
      if (tid0 == 0) lds[1] = (__hip_atomic_fetch_or(claim + (cand >> 5), 1u << (cand & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (cand & 31)) & 1u;
      __syncthreads();
      const int lost = __builtin_amdgcn_readfirstlane(lds[1]);
      __syncthreads();
      if (!lost) { if (tid0 == 0) __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); cur = cand; return true; }
    }
  }
  __device__ __forceinline__ bool leading() const { return w0 && member == 0 && follow; }
  __device__ __forceinline__ bool following() const { return w0 && member != 0 && follow; }
  __device__ __forceinline__ unsigned* slot() const { return box + (seq & 7); }
  __device__ __forceinline__ void begin_tile() { ra = 1; rb = 0; bit = 0; tile = -2; }
  __device__ __forceinline__ void draw() { asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(ra) : "s"(cnt) : "memory"); }
  __device__ __forceinline__ void claim_issue(unsigned t) {
    const unsigned ti = t * GANG + member;
    tile = -1;
    if (ti < (unsigned)n_tiles) {
      tile = (int)ti; bit = 1u << (ti & 31); rb = bit;
      unsigned* wp = claim + (ti >> 5);
      asm volatile("s_atomic_or %0, %1, 0x0 glc" : "+s"(rb) : "s"(wp) : "memory");
    }
  }
  // (behind a wait for lgkmcnt(0))
  __device__ __forceinline__ void publish() {
    asm volatile("" : "+s"(ra));
    const unsigned t = ra < total ? ra : kTkEnd;
    const unsigned pub = (tag(seq) << 24) | t;
    unsigned* sp = slot();
    asm volatile("s_atomic_swap %0, %1, 0x0" :: "s"(pub), "s"(sp) : "memory");
    if (t != kTkEnd) claim_issue(t);
  }
  __device__ __forceinline__ void ask() { unsigned* sp = slot(); asm volatile("s_load_dword %0, %1, 0x0 glc" : "=s"(ra) : "s"(sp) : "memory"); }
  __device__ __forceinline__ void check() {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ra) :: "memory");
    unsigned* sp = slot();
    for (int i = 0; i < (1 << 16) && (ra >> 24) != tag(seq); ++i) {
      __builtin_amdgcn_s_sleep(8);
      asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(ra) : "s"(sp) : "memory");
    }
    if ((ra >> 24) == tag(seq) && (ra & 0xffffffu) != kTkEnd) claim_issue(ra & 0xffffffu);
  }
  // (behind a wait for lgkmcnt(0))
  __device__ __forceinline__ void result(int lane) {
    if (tile >= 0) { asm volatile("" : "+s"(rb)); if (rb & bit) tile = -1; }
    if (tile >= 0) { const unsigned one = 1u; unsigned* cp = cnt + 1; asm volatile("s_atomic_add %0, %1, 0x0" :: "s"(one), "s"(cp) : "memory"); }
    if (lane == 0) lds[0] = tile;
  }
  __device__ __forceinline__ int handed_over() const { return follow ? __builtin_amdgcn_readfirstlane(lds[0]) : -2; }
  __device__ __forceinline__ void advance(int fut) { ++seq; if (fut == -2) follow = false; }
};

}  // namespace sfft
