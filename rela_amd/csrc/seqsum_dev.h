// seqsum_dev.h -- device-side view of a built sequential-sum index (see seqsum_core.h) and the
// host entry points that build it.  Shared by seqsum.hip and replay.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "seqsum_core.h"

namespace rela_amd {

// Everything a kernel needs to evaluate the reference's running sum
// (rela/prioritized_replay.h:304-306) at any logical index of the ring.
struct SeqView {
  const float* w;  // weight ring (device)
  int64_t ring, head, size;
  const SeqTab* T1;
  const SeqTab* T2;
  const double* A3;  // exact accumulator before each level-3 node; A3[n3] = total
  int n3;
};

struct SeqRingAt {
  const float* w;
  int64_t ring, head, size;
  __device__ __forceinline__ float operator()(int64_t k) const {
    if (k >= size) return 0.f;
    int64_t p = head + k;
    if (p >= ring) p -= ring;
    return w[p];
  }
};

__device__ __forceinline__ int64_t seq_phys(const SeqView& v, int64_t k) {
  int64_t p = v.head + k;
  return p >= v.ring ? p - v.ring : p;
}

// first logical index whose inclusive sequential prefix reaches `target` (> 0)
__device__ inline SeqHit seq_find(const SeqView& v, double target) {
  int lo = 0, hi = v.n3;  // smallest j with A3[j+1] >= target
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (v.A3[mid + 1] >= target) hi = mid; else lo = mid + 1;
  }
  if (lo >= v.n3) {
    SeqHit h;
    h.k = v.size;
    h.A = v.A3[v.n3];
    h.w = 0.f;
    h.found = false;
    return h;
  }
  int64_t kend = (int64_t)(lo + 1) * kL3;
  if (kend > v.size) kend = v.size;
  return seq_walk(v.A3[lo], (int64_t)lo * kL3, kend, target, v.T1, v.T2, (const SeqTab*)nullptr,
                  SeqRingAt{v.w, v.ring, v.head, v.size});
}

// exact sequential sum of the first k weights (k <= size)
__device__ inline double seq_prefix(const SeqView& v, int64_t k) {
  int j = (int)(k / kL3);
  if (j > v.n3) j = v.n3;
  if (j == v.n3) return v.A3[v.n3];
  return seq_walk(v.A3[j], (int64_t)j * kL3, k, (double)INFINITY, v.T1, v.T2, (const SeqTab*)nullptr,
                  SeqRingAt{v.w, v.ring, v.head, v.size}).A;
}

// ---- wave-cooperative walk -------------------------------------------------------------
// Same arithmetic as seq_walk (seqsum_core.h), executed by all 64 lanes of a wavefront with
// identical scalar state: sibling tables (16 x 24 B) and native weights (64 x 4 B) are fetched
// by one coalesced load each and consumed through lane broadcasts, so a search costs ~3 memory
// round trips instead of ~100 dependent ones.  Every lane must call these with the same
// arguments; every lane gets the same result.
// `src` is wave-uniform: v_readlane_b32 (a few cycles) instead of ds_bpermute_b32 (an LDS round trip)
__device__ __forceinline__ int rl_i(int x, int src) { return __builtin_amdgcn_readlane(x, src); }
__device__ __forceinline__ float rl_f(float x, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src));
}
__device__ __forceinline__ double rl_d(double x, int src) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ SeqTab seq_tab_bcast(const SeqTab& mine, int src) {
  SeqTab t;
  t.d[0] = rl_d(mine.d[0], src);
  t.d[1] = rl_d(mine.d[1], src);
  t.e = rl_i(mine.e, src);
  t.par = rl_i(mine.par, src);
  return t;
}

// Walks [k0, kend) inside ONE level-3 node (k0 is that node's first index).  pad_ok: tables may
// be applied even when their span runs past kend (true for searches up to `size`, where the
// tables were built with zero padding; false when kend is an arbitrary prefix end).
__device__ inline SeqHit seq_walk_wave(const SeqView& v, double A0, int64_t k0, int64_t kend, double target,
                                       bool pad_ok) {
  const int lane = threadIdx.x & 63;
  const SeqRingAt at{v.w, v.ring, v.head, v.size};
  SeqHit h;
  double A = A0;
  int64_t k = k0;
  const int64_t n2 = (v.size + kL2 - 1) / kL2, n1 = n2 * kFan;
  SeqTab my2 = seq_tab_invalid();
  {
    const int64_t i2 = k0 / kL2 + lane;
    if (lane < kFan && i2 < n2) my2 = v.T2[i2];
  }
  for (int c2 = 0; c2 < kFan && k < kend; ++c2) {
    double n;
    const SeqTab t2 = seq_tab_bcast(my2, c2);
    if ((pad_ok || k + kL2 <= kend) && seq_apply(t2, A, &n) && n < target) {
      A = n;
      k += kL2;
      continue;
    }
    SeqTab my1 = seq_tab_invalid();
    {
      const int64_t i1 = k / kL1 + lane;
      if (lane < kFan && i1 < n1) my1 = v.T1[i1];
    }
    const int64_t end2 = (k + kL2 < kend) ? k + kL2 : kend;
    for (int c1 = 0; c1 < kFan && k < end2; ++c1) {
      const SeqTab t1 = seq_tab_bcast(my1, c1);
      if ((pad_ok || k + kL1 <= kend) && seq_apply(t1, A, &n) && n < target) {
        A = n;
        k += kL1;
        continue;
      }
      const float wl = at(k + lane);  // 64 weights, one coalesced load (0 beyond size)
      const int cnt = (int)((k + kL1 <= kend) ? kL1 : (kend - k));
      for (int e = 0; e < cnt; ++e) {
        const float we = rl_f(wl, e);
        A += (double)we;
        if (A >= target) {
          h.k = k + e;
          h.A = A;
          h.w = we;
          h.found = true;
          return h;
        }
      }
      k += cnt;
    }
  }
  h.k = kend;
  h.A = A;
  h.w = 0.f;
  h.found = false;
  return h;
}

// wave-cooperative versions of seq_find / seq_prefix
__device__ inline SeqHit seq_find_wave(const SeqView& v, double target) {
  int lo = 0, hi = v.n3;  // smallest j with A3[j+1] >= target (uniform across the wave)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (v.A3[mid + 1] >= target) hi = mid; else lo = mid + 1;
  }
  if (lo >= v.n3) {
    SeqHit h;
    h.k = v.size;
    h.A = v.A3[v.n3];
    h.w = 0.f;
    h.found = false;
    return h;
  }
  int64_t kend = (int64_t)(lo + 1) * kL3;
  if (kend > v.size) kend = v.size;
  return seq_walk_wave(v, v.A3[lo], (int64_t)lo * kL3, kend, target, true);
}

__device__ inline double seq_prefix_wave(const SeqView& v, int64_t k) {
  int j = (int)(k / kL3);
  if (j >= v.n3) return v.A3[v.n3];
  return seq_walk_wave(v, v.A3[j], (int64_t)j * kL3, k, (double)INFINITY, false).A;
}

// ---- host side -----------------------------------------------------------------------
struct SeqIndex {
  double* bsum2 = nullptr;  // [n2cap]     plain f64 sums of level-2 nodes (guesses only)
  double* S0 = nullptr;     // [n2cap + 1] their exclusive prefix
  SeqTab* T1 = nullptr;     // [n2cap * 16]
  SeqTab* T2 = nullptr;     // [n2cap]
  SeqTab* T3 = nullptr;     // [n3cap]
  double* A3 = nullptr;     // [n3cap + 1]
  int n2cap = 0, n3cap = 0;
};

int seq_index_alloc(SeqIndex* ix, int64_t max_elems);
void seq_index_free(SeqIndex* ix);
// Queues the build (5 small kernels) for the live range [head, head+size) on `stream` and
// fills `view` (plain struct, pass by value to kernels queued on the same stream afterwards).
int seq_index_build(const SeqIndex& ix, const float* ring_dev, int64_t ring, int64_t head,
                    int64_t size, hipStream_t stream, SeqView* view);

}  // namespace rela_amd
