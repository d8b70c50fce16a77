// seqsum_core.h -- exact parallel evaluation of a SEQUENTIALLY ROUNDED f64 running sum.
//
// The reference walks the priority ring with `double accSum += (double)float_w`
// (rela/prioritized_replay.h:304-306) and evicts with `double diff -= w` (:85-95).
// Every step rounds to nearest-even, so the value at index k depends on the order of
// all earlier additions: a tree-ordered parallel sum is NOT bit-identical.  Replay ids
// are decided by `accSum >= rand` (:286), so we reproduce the sequential value exactly:
//
//   While the accumulator A stays inside one binade [2^e, 2^(e+1)) its ulp u = 2^(e-52)
//   is constant, A is an integer multiple n*u, and adding a float w = (q + f)*u gives
//       n' = n + q + [f > 1/2] + [f == 1/2]*((n + q) & 1)            (round-half-even)
//   i.e. an increment that depends on A only through the PARITY of n.  Such maps
//   p -> D_p compose associatively:  (L then R)_p = L_p + R_{(p + L_p) & 1}.
//   A node of the ring (64 / 1024 / 16384 consecutive weights) therefore has a
//   "transfer table" (D_0, D_1) valid for one binade e.  Tables are built in parallel
//   from a GUESS of e; they are only APPLIED after checking, with the exact incoming A,
//   that binade(A) == e and binade(A + D_p) == e (A is monotone, so every intermediate
//   value is then in the binade too).  A failed check falls through to the next finer
//   level and finally to native sequential adds.  The result is exact by construction;
//   the quality of the guess only changes how much work is skipped.
//
// Everything here is __host__ __device__ so the CPU unit tests (tests/cpu_shims) run the
// same arithmetic as the kernels in seqsum.hip.
#pragma once
#include <stdint.h>
#include <string.h>

#ifndef RELA_HD
#ifdef __HIPCC__
#define RELA_HD __host__ __device__ __forceinline__
#else
#define RELA_HD inline
#endif
#endif

namespace rela_amd {

constexpr int kL1 = 64;              // elements per level-1 node (one wavefront, one lane per weight)
constexpr int kFan = 16;             // children per node above level 1
constexpr int kL2 = kL1 * kFan;      // 1024
constexpr int kL3 = kL2 * kFan;      // 16384

constexpr int kTabInvalid = -1;  // never applicable
constexpr int kTabAny = -2;      // all-zero node: identity in every binade
constexpr int kTabExact = -3;    // d[0] = the exact accumulator entering the node, d[1] = leaving it: written by
                                 // the chain for crossing nodes once both are known; applies iff A == d[0]

// Transfer table of one node.  d[p] is the exact increment (already scaled by u, exact
// in f64) for incoming parity p; par bit p is the parity of that increment in units of u.
struct SeqTab {
  double d[2];
  int32_t e;    // biased f64 exponent of the binade, or kTabInvalid / kTabAny
  int32_t par;  // bit0 = parity(D_0), bit1 = parity(D_1)
};

RELA_HD uint64_t dbits(double x) {
#ifdef __HIP_DEVICE_COMPILE__
  return (uint64_t)__double_as_longlong(x);
#else
  uint64_t u;
  memcpy(&u, &x, 8);
  return u;
#endif
}
RELA_HD double bitsd(uint64_t u) {
#ifdef __HIP_DEVICE_COMPILE__
  return __longlong_as_double((long long)u);
#else
  double x;
  memcpy(&x, &u, 8);
  return x;
#endif
}
RELA_HD uint32_t fbits(float x) {
#ifdef __HIP_DEVICE_COMPILE__
  return __float_as_uint(x);
#else
  uint32_t u;
  memcpy(&u, &x, 4);
  return u;
#endif
}
RELA_HD int dexp(double x) { return (int)((dbits(x) >> 52) & 0x7ff); }

// 2^(ed - 1075): the ulp of binade `ed` (biased exponent).  ed > 52 always holds for sums
// of floats (the smallest positive float is 2^-149, biased f64 exponent 874).
RELA_HD double ulp_of(int ed) { return bitsd((uint64_t)(ed - 52) << 52); }

struct SeqElem {
  int64_t D0, D1;
  bool bad;
};

// Increment of one float weight in units of u = ulp_of(ed), for incoming parity 0 and 1.
RELA_HD SeqElem seq_classify(float w, int ed) {
  SeqElem r;
  r.D0 = r.D1 = 0;
  r.bad = false;
  const uint32_t b = fbits(w);
  if ((b << 1) == 0) return r;  // +-0
  const int ef = (int)((b >> 23) & 0xff);
  if ((b >> 31) || ef == 0xff) {  // negative, inf or NaN: only the native path is faithful
    r.bad = true;
    return r;
  }
  const int64_t m = ef ? (int64_t)((b & 0x7fffff) | 0x800000) : (int64_t)(b & 0x7fffff);
  const int ew = ef ? ef - 150 : -149;  // w = m * 2^ew
  const int t = ed - 1075;              // u = 2^t
  const int s = ew - t;
  if (s >= 0) {
    if (s > 29) {  // m << s could reach 2^53: the add leaves the binade by itself
      r.bad = true;
      return r;
    }
    r.D0 = r.D1 = m << s;
    return r;
  }
  const int sh = -s;
  if (sh > 24) return r;  // w < u/2: rounds away for either parity
  const int64_t q = m >> sh;
  const int64_t rem = m & (((int64_t)1 << sh) - 1);
  const int64_t half = (int64_t)1 << (sh - 1);
  if (rem > half) {
    r.D0 = r.D1 = q + 1;
  } else if (rem < half) {
    r.D0 = r.D1 = q;
  } else {  // tie: to even of (n + q)
    r.D0 = q + (q & 1);
    r.D1 = q + ((q + 1) & 1);
  }
  return r;
}

// (L then R) in integer units
RELA_HD SeqElem seq_compose_int(const SeqElem& L, const SeqElem& R) {
  SeqElem o;
  o.D0 = L.D0 + ((L.D0 & 1) ? R.D1 : R.D0);
  o.D1 = L.D1 + (((1 + L.D1) & 1) ? R.D1 : R.D0);
  o.bad = L.bad || R.bad;
  return o;
}

constexpr int64_t kSeqLim = (int64_t)1 << 53;

RELA_HD SeqTab seq_make_tab(const SeqElem& x, int ed) {
  SeqTab t;
  if (x.bad || x.D0 > kSeqLim || x.D1 > kSeqLim || ed <= 52) {
    t.d[0] = t.d[1] = 0;
    t.e = kTabInvalid;
    t.par = 0;
    return t;
  }
  const double u = ulp_of(ed);
  t.d[0] = (double)x.D0 * u;  // exact: D <= 2^53, power-of-two scale
  t.d[1] = (double)x.D1 * u;
  t.e = ed;
  t.par = (int)(x.D0 & 1) | ((int)(x.D1 & 1) << 1);
  return t;
}

RELA_HD SeqTab seq_tab_any() {
  SeqTab t;
  t.d[0] = t.d[1] = 0;
  t.e = kTabAny;
  t.par = 0;
  return t;
}
RELA_HD SeqTab seq_tab_invalid() {
  SeqTab t;
  t.d[0] = t.d[1] = 0;
  t.e = kTabInvalid;
  t.par = 0;
  return t;
}

// Compose two tables (L then R).  kTabAny is the identity; a binade mismatch or an
// increment beyond 2^53 ulps (never applicable anyway) yields an invalid table.
RELA_HD SeqTab seq_compose(const SeqTab& L, const SeqTab& R) {
  if (L.e == kTabInvalid || R.e == kTabInvalid || L.e == kTabExact || R.e == kTabExact) return seq_tab_invalid();
  if (L.e == kTabAny) return R;
  if (R.e == kTabAny) return L;
  if (L.e != R.e) return seq_tab_invalid();
  SeqTab o;
  const double lim = ulp_of(L.e) * 9007199254740992.0;  // 2^53 ulps
  const int p0 = (L.par & 1);         // parity after L for incoming parity 0
  const int p1 = 1 ^ ((L.par >> 1) & 1);  // parity after L for incoming parity 1
  o.d[0] = L.d[0] + R.d[p0];
  o.d[1] = L.d[1] + R.d[p1];
  if (o.d[0] > lim || o.d[1] > lim) return seq_tab_invalid();
  o.e = L.e;
  o.par = ((L.par & 1) ^ ((R.par >> p0) & 1)) | ((((L.par >> 1) & 1) ^ ((R.par >> p1) & 1)) << 1);
  return o;
}

// Try to advance the exact accumulator A across a node.  Returns true and the new value
// if the table is provably applicable; false leaves *out untouched.
RELA_HD bool seq_apply(const SeqTab& t, double A, double* out) {
  if (t.e == kTabAny) {
    *out = A;
    return true;
  }
  if (t.e == kTabInvalid) return false;
  if (t.e == kTabExact) {
    if (dbits(A) != dbits(t.d[0])) return false;
    *out = t.d[1];
    return true;
  }
  const uint64_t ab = dbits(A);
  if ((int)((ab >> 52) & 0x7ff) != t.e) return false;
  const double n = A + t.d[ab & 1];
  if (dexp(n) != t.e) return false;
  *out = n;
  return true;
}

// Binade guess for a node from approximate prefix sums at its two ends.
RELA_HD int seq_guess(double s_begin, double s_end) {
  if (!(s_begin > 0)) return kTabInvalid;  // the first positive weight is a crossing by definition
  const int a = dexp(s_begin), b = dexp(s_end);
  return (a == b && a > 52 && a < 0x7ff) ? a : kTabInvalid;
}

// Exact walk of the sequential accumulator.  Starts at logical index k0 with the exact
// value A0 (the accumulator BEFORE element k0) and advances to kend, skipping whole nodes
// through their tables when `seq_apply` proves that legal.  Stops at the first k whose
// inclusive prefix A_k >= target (the reference's hit test `accSum >= rand`, :286; callers
// map `accSum > 0` onto a tiny positive target) and reports (k, A_k, w_k); if the target is
// never reached it reports (kend, A_{kend-1}, 0).  T3 may be null.  `wat(k)` returns the
// weight at logical index k (callers mask k >= size to 0).
struct SeqHit {
  int64_t k;
  double A;
  float w;
  bool found;
};

template <class WAt>
RELA_HD SeqHit seq_walk(double A0, int64_t k0, int64_t kend, double target, const SeqTab* T1,
                        const SeqTab* T2, const SeqTab* T3, WAt wat) {
  SeqHit h;
  double A = A0;
  int64_t k = k0;
  while (k < kend) {
    double n;
    if (T3 != nullptr && (k % kL3) == 0 && k + kL3 <= kend) {
      if (seq_apply(T3[k / kL3], A, &n) && n < target) {
        A = n;
        k += kL3;
        continue;
      }
    }
    if ((k % kL2) == 0 && k + kL2 <= kend) {
      if (seq_apply(T2[k / kL2], A, &n) && n < target) {
        A = n;
        k += kL2;
        continue;
      }
    }
    if ((k % kL1) == 0 && k + kL1 <= kend) {
      if (seq_apply(T1[k / kL1], A, &n) && n < target) {
        A = n;
        k += kL1;
        continue;
      }
    }
    // native sequential adds up to the next level-1 boundary
    int64_t stop = (k / kL1 + 1) * kL1;
    if (stop > kend) stop = kend;
    for (; k < stop; ++k) {
      const float w = wat(k);
      A += (double)w;
      if (A >= target) {
        h.k = k;
        h.A = A;
        h.w = w;
        h.found = true;
        return h;
      }
    }
  }
  h.k = kend;
  h.A = A;
  h.w = 0.f;
  h.found = false;
  return h;
}

}  // namespace rela_amd
