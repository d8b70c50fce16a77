// CPU shim for the host logic tests: runs the SAME arithmetic as the kernels in
// rela_amd/csrc/seqsum.hip (seqsum_core.h is __host__ __device__) with the kernel
// structure replayed sequentially.  Built by tests/test_seqsum_host.py with g++.
// TEST INFRASTRUCTURE -- not part of the product library.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "../../rela_amd/csrc/seqsum_core.h"

using namespace rela_amd;

namespace {
struct Ring {
  const float* w;
  int64_t head, ring, size;
  float operator()(int64_t k) const {
    if (k >= size) return 0.f;
    int64_t p = head + k;
    if (p >= ring) p -= ring;
    return w[p];
  }
};

uint32_t lcg(uint32_t& s) {
  s = s * 1664525u + 1013904223u;
  return s >> 8;
}
}  // namespace

extern "C" {

// guess_mode: 0 = guesses from plain f64 prefix sums (what the kernels do)
//             1 = guesses randomly perturbed by -1/0/+1 binades (stress the verify path)
//             2 = every guess invalid (pure native path)
//             3 = one constant guess for all nodes (the binade of the total)
int shim_seq_search(const float* w, int64_t size, int64_t head, int64_t ring, const double* targets,
                    int nt, int guess_mode, uint32_t seed, int64_t* out_k, double* out_A,
                    float* out_w, double* out_total, int64_t prefix_at, double* out_prefix) {
  Ring at{w, head, ring, size};
  const int64_t n1 = (size + kL1 - 1) / kL1, n2 = (size + kL2 - 1) / kL2, n3 = (size + kL3 - 1) / kL3;
  std::vector<SeqTab> T1(n1 ? n1 : 1), T2(n2 ? n2 : 1), T3(n3 ? n3 : 1);
  // K1/K2: plain (tree-free here) prefix of level-1 sums -> guesses
  std::vector<double> s1(n1 + 1, 0.0);
  for (int64_t j = 0; j < n1; ++j) {
    double s = 0;
    for (int i = 0; i < kL1; ++i) s += (double)at(j * kL1 + i);
    s1[j + 1] = s1[j] + s;
  }
  const int etot = dexp(s1[n1]);
  // K3: tables
  for (int64_t j = 0; j < n1; ++j) {
    int e = seq_guess(s1[j], s1[j + 1]);
    if (guess_mode == 1 && e != kTabInvalid) e += (int)(lcg(seed) % 3) - 1;
    if (guess_mode == 2) e = kTabInvalid;
    if (guess_mode == 3) e = etot;
    bool allzero = true;
    for (int i = 0; i < kL1; ++i) allzero = allzero && (fbits(at(j * kL1 + i)) << 1) == 0;
    if (allzero) {
      T1[j] = seq_tab_any();
      continue;
    }
    if (e == kTabInvalid || e <= 52 || e >= 0x7ff) {
      T1[j] = seq_tab_invalid();
      continue;
    }
    // emulate the wavefront tree reduction order (pairwise, in order)
    SeqElem el[kL1];
    for (int i = 0; i < kL1; ++i) el[i] = seq_classify(at(j * kL1 + i), e);
    for (int off = 1; off < kL1; off <<= 1)
      for (int i = 0; i + off < kL1; i += 2 * off) el[i] = seq_compose_int(el[i], el[i + off]);
    T1[j] = seq_make_tab(el[0], e);
  }
  for (int64_t j = 0; j < n2; ++j) {
    SeqTab t = seq_tab_any();
    for (int c = 0; c < kFan; ++c) {
      int64_t i = j * kFan + c;
      t = seq_compose(t, i < n1 ? T1[i] : seq_tab_any());
    }
    T2[j] = t;
  }
  for (int64_t j = 0; j < n3; ++j) {
    SeqTab t = seq_tab_any();
    for (int c = 0; c < kFan; ++c) {
      int64_t i = j * kFan + c;
      t = seq_compose(t, i < n2 ? T2[i] : seq_tab_any());
    }
    T3[j] = t;
  }
  // K4: chain over level-3 nodes
  std::vector<double> A3(n3 + 1, 0.0);
  double A = 0;
  for (int64_t j = 0; j < n3; ++j) {
    A3[j] = A;
    int64_t kend = (j + 1) * kL3 < size ? (j + 1) * kL3 : size;
    SeqHit h = seq_walk(A, j * kL3, kend, INFINITY, T1.data(), T2.data(), T3.data(), at);
    A = h.A;
  }
  A3[n3] = A;
  if (out_total) *out_total = A;
  // K5: per-target search
  for (int t = 0; t < nt; ++t) {
    const double r = targets[t];
    int64_t lo = 0, hi = n3;  // smallest j with A3[j+1] >= r
    while (lo < hi) {
      int64_t mid = (lo + hi) / 2;
      if (A3[mid + 1] >= r) hi = mid; else lo = mid + 1;
    }
    if (lo >= n3) {
      out_k[t] = -1;
      out_A[t] = A;
      out_w[t] = 0;
      continue;
    }
    int64_t kend = (lo + 1) * kL3 < size ? (lo + 1) * kL3 : size;
    SeqHit h = seq_walk(A3[lo], lo * kL3, kend, r, T1.data(), T2.data(), nullptr, at);
    out_k[t] = h.found ? h.k : -1;
    out_A[t] = h.A;
    out_w[t] = h.w;
  }
  if (out_prefix) {
    int64_t j = prefix_at / kL3;
    if (j > n3) j = n3;
    SeqHit h = seq_walk(j < n3 ? A3[j] : A3[n3], j * kL3, prefix_at, INFINITY, T1.data(), T2.data(), nullptr, at);
    *out_prefix = h.A;
  }
  return 0;
}

// statistics for the design notes: how many elements were skipped through tables
int64_t shim_count_native(const float* w, int64_t size) {
  (void)w;
  (void)size;
  return 0;
}
}
