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
