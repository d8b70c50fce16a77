// seqsum_dev.h -- device-side view of a built sequential-sum index (see seqsum_core.h) and the
// host entry points that build it.  Shared by seqsum.hip and replay.hip.
//
// The index gives the EXACT value of the reference's running sum (rela/prioritized_replay.h:304-306)
// before every level-2 node (1024 weights): A2[i] = acc before logical element 1024*i, and A3 for the
// level-3 nodes.  A search for a stratified target is two coalesced loads over the monotone arrays A3 / A2,
// one coalesced load of the 16 level-1 transfer tables of the level-2 node it lands in (applied in order,
// each application verified), and at most 64 native adds inside one level-1 node; a prefix (blockPop's
// diff, :85-95) walks the same way.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "seqsum_core.h"

namespace rela_amd {

// device-resident scalars of one replay (replay.hip); the chain kernel reads `sum` for the stratified targets
struct ReplayDevState {
  double sum;       // ConcurrentQueue::sum_
  float sum_f;      // sum_ narrowed at the last sample_ (:261-262)
  int32_t err;      // sticky device-side error
  double last_pop;  // diff of the last blockPop (diagnostic)
};

struct SeqView {
  const float* w;  // weight ring (device)
  int64_t ring, head, size;
  const SeqTab* T1;  // [16*n2] level-1 transfer tables (zero padded to whole level-2 nodes)
  const double* A2;  // [16*n3 + 1]  exact accumulator before each level-2 node (zero padded past size)
  const double* A3;  // [n3 + 1]     ... before each level-3 node; A3[n3] = total
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

// Native walk of the 64 weights of level-1 node u held one per lane (wl), from the exact accumulator A
// (wave-uniform): stops at the first element whose inclusive sum reaches `target`, or after `cnt` elements.
__device__ __forceinline__ SeqHit seq_native_wave(float wl, double A, int64_t k0, int cnt, double target) {
  SeqHit h;
  for (int e = 0; e < cnt; ++e) {
    const float we = rl_f(wl, e);
    A += (double)we;
    if (A >= target) {
      h.k = k0 + e;
      h.A = A;
      h.w = we;
      h.found = true;
      return h;
    }
  }
  h.k = k0 + cnt;
  h.A = A;
  h.w = 0.f;
  h.found = false;
  return h;
}

// Walks the level-2 node i2 from its exact start value: the 16 level-1 tables arrive in ONE coalesced load and
// are applied through lane broadcasts (verified; a table that cannot be applied, or that would pass `target`,
// is walked natively).  Stops at the first element whose inclusive sum reaches `target`, or at element kend.
__device__ inline SeqHit seq_walk_l2_wave(const SeqView& v, int64_t i2, double A, int64_t kend, double target) {
  const int lane = threadIdx.x & 63;
  const SeqRingAt at{v.w, v.ring, v.head, v.size};
  SeqTab mine = seq_tab_any();
  if (lane < kFan) mine = v.T1[i2 * kFan + lane];
  int64_t k = i2 * kL2;
  for (int c = 0; c < kFan && k < kend; ++c) {
    const SeqTab t1 = seq_tab_bcast(mine, c);
    double n;
    if (k + kL1 <= kend && seq_apply(t1, A, &n) && n < target) {
      A = n;
      k += kL1;
      continue;
    }
    const float wl = at(k + lane);  // 64 weights, one coalesced load (0 beyond size)
    const int cnt = (int)((k + kL1 <= kend) ? kL1 : (kend - k));
    const SeqHit h = seq_native_wave(wl, A, k, cnt, target);
    if (h.found) return h;
    A = h.A;
    k += cnt;
  }
  SeqHit h;
  h.k = kend;
  h.A = A;
  h.w = 0.f;
  h.found = false;
  return h;
}

// Wave-cooperative search: first logical index whose inclusive sequential prefix reaches `target` (> 0).
// Every lane must call with the same arguments; every lane gets the same result.
__device__ inline SeqHit seq_find_wave(const SeqView& v, double target) {
  const int lane = threadIdx.x & 63;
  // level 3: j = number of level-3 nodes whose END value is still below the target
  int j = 0;
  for (int base = 0; base < v.n3; base += 64) {
    const int i = base + lane;
    const bool below = i < v.n3 && v.A3[i + 1] < target;
    j += __popcll(__ballot(below));
  }
  if (j >= v.n3) {  // never reached (:297-302, the reference aborts here)
    SeqHit h;
    h.k = v.size;
    h.A = v.A3[v.n3];
    h.w = 0.f;
    h.found = false;
    return h;
  }
  // level 2: 16 children (the array is padded to whole level-3 nodes)
  const bool b2 = lane < kFan && v.A2[(int64_t)j * kFan + 1 + lane] < target;
  const int64_t i2 = (int64_t)j * kFan + __popcll(__ballot(b2));
  int64_t kend = (i2 + 1) * kL2;
  if (kend > v.size) kend = v.size;
  return seq_walk_l2_wave(v, i2, v.A2[i2], kend, target);
}

// exact sequential sum of the first k weights (k <= size), wave-cooperative
__device__ inline double seq_prefix_wave(const SeqView& v, int64_t k) {
  if (k >= v.size) return v.A3[v.n3];
  const int64_t i2 = k / kL2;
  return seq_walk_l2_wave(v, i2, v.A2[i2], k, (double)INFINITY).A;
}

// ---- host side -----------------------------------------------------------------------
// A level-1 node whose binade guess is invalid (the running sum crosses a power of two inside it) is a
// "crossing node".  The tables kernel records it with a speculative split at the crossing element m:
// B = transfer table of elements [0, m) in the binade before, C = table of (m, 64) in the binade after.
// The chain applies B, adds w[m] natively and applies C, each step verified; any failed check falls back
// to 64 native adds, so the split only decides how much work is skipped.
struct SeqRec {
  int32_t node;  // level-1 node index
  int32_t kind;  // 0 = split (B, wm, C), 1 = walk natively
  int32_t m;
  float wm;
  SeqTab B, C;
};
constexpr int kMaxRec = 256;

struct SeqIndex {
  double* s2 = nullptr;     // [n2cap]       plain f64 sums of level-2 nodes (binade guesses only)
  double* s1 = nullptr;     // [n2cap * 16]  plain f64 sums of level-1 nodes
  SeqTab* T1 = nullptr;     // [n2cap * 16]
  SeqTab* T2 = nullptr;     // [n2cap]
  SeqTab* T3 = nullptr;     // [n3cap]
  double* A3 = nullptr;     // [n3cap + 1]
  double* A2 = nullptr;     // [n3cap * 16 + 1]
  SeqRec* rec = nullptr;    // [kMaxRec]
  int32_t* ctl = nullptr;   // [0] number of crossing records, [1] count of fallback runs (diagnostic)
  int n2cap = 0, n3cap = 0;
};

// Optional second job of the chain kernel (it is a single workgroup with idle lanes): the stratified
// targets of a sample (prioritized_replay.h:261-280).  batch = 0: none.
struct SeqTargetsJob {
  const uint32_t* draws = nullptr;  // raw mt19937 outputs, one per sample
  int batch = 0;
  ReplayDevState* state = nullptr;  // sum -> sum_f
  float* targets = nullptr;  // [batch] clamped targets (diagnostic)
  double* eff = nullptr;     // [batch] effective (prefix-max) targets
};

int seq_index_alloc(SeqIndex* ix, int64_t max_elems);
void seq_index_free(SeqIndex* ix);
// Queues the build (3 kernels) for the live range [head, head+size) on `stream` and fills `view`
// (plain struct, pass by value to kernels queued on the same stream afterwards).
int seq_index_build(const SeqIndex& ix, const float* ring_dev, int64_t ring, int64_t head, int64_t size,
                    hipStream_t stream, SeqView* view, const SeqTargetsJob* targets = nullptr);

}  // namespace rela_amd
