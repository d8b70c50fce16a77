// seqsum.hip -- kernels that build the exact sequential-f64-sum index over a weight ring.
//
// Restates the accumulation of rela/prioritized_replay.h:266-308 (sample_ scan) and :85-95
// (blockPop diff) for a GPU: see seqsum_core.h for the arithmetic and the proof sketch.
//
// Pipeline (all stream-ordered, no host sync), N = live weights:
//   seq_sums    grid N/1024  reads 4N B   plain f64 sums of every level-1 (64) and level-2 (1024) node
//   seq_tables  grid N/1024  reads 4N B   (L2 / MALL hits) binade guesses from those sums, level-1 / level-2
//                                         transfer tables, and one RECORD per crossing node: the level-1 nodes
//                                         whose guess is invalid, with a speculative split at the crossing
//   seq_chain   1 workgroup  -            the sequential part, reduced to ~4 verified table applications per
//                                         binade crossing (~25 on a 2^20 ring), then the exact accumulator
//                                         before every level-3 and level-2 node (A3, A2) in parallel; optionally
//                                         the stratified targets of a sample ride along
// Algorithmic bytes: 4N (the reference's linear scan reads each weight once).
//
// Structure of seq_chain (exactness never depends on a guess; every table application is verified with the
// exact incoming accumulator, a failed check falls back to a slower exact path):
//   0  level-3 tables (16 level-2 tables each) into LDS
//   1  crossing records sorted by node
//   2  one composed table per GAP between two crossing nodes (wave-parallel: the gap is cut into aligned
//      level-1 / level-2 / level-3 pieces, loaded one per lane and composed with a shuffle scan)
//   3  one wavefront walks gaps and crossings in order: apply(gap) -> apply(B) -> native add of the crossing
//      weight -> apply(C); a failed record falls back to the 64 native adds of its node, a failed gap to the
//      legacy walk below
//   4  exact values at the level-2 / level-3 boundaries from the gap starts (one thread per gap; whole level-3
//      nodes are jumped), then the level-2 boundaries inside the jumped nodes (one thread per node); the
//      level-2 tables sit in LDS, so none of these walks waits on memory
//   legacy  (record overflow / failed gap): one wavefront walks level-3 nodes with on-demand descent, as the
//      round-1 kernel did
#include "common.h"
#include "prof.h"
#include "seqsum_dev.h"

namespace rela_amd {

namespace {

constexpr int kBlock = 256;  // 4 wavefronts; one workgroup per level-2 node (1024 weights)
constexpr int kChainThreads = 1024;
constexpr int kT2Lds = 2048;  // level-2 tables are staged in LDS up to this many (ring <= 2,097,152)

__device__ __forceinline__ float ring_load(const float* w, int64_t ring, int64_t head, int64_t size,
                                           int64_t k) {
  if (k >= size) return 0.f;
  int64_t p = head + k;
  if (p >= ring) p -= ring;
  return w[p];
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(kBlock) void seq_sums(const float* __restrict__ w, int64_t ring, int64_t head,
                                                   int64_t size, double* __restrict__ s1, double* __restrict__ s2,
                                                   int32_t* __restrict__ ctl) {
  __shared__ double part[kFan];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kL2;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = wave * 4 + c;
    const double s = wave_sum((double)ring_load(w, ring, head, size, base + (int64_t)j * kL1 + lane));
    if (lane == 0) {
      part[j] = s;
      s1[(int64_t)blockIdx.x * kFan + j] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
#pragma unroll
    for (int j = 0; j < kFan; ++j) t += part[j];
    s2[blockIdx.x] = t;
    if (blockIdx.x == 0) ctl[0] = 0;  // the record counter of this build
  }
}

__device__ __forceinline__ SeqElem shfl_down_elem(const SeqElem& e, int off) {
  SeqElem o;
  o.D0 = __shfl_down((long long)e.D0, off, 64);
  o.D1 = __shfl_down((long long)e.D1, off, 64);
  o.bad = false;
  return o;
}

// ordered reduction of 64 per-lane elements (lane order = element order); lane 0 gets the total
__device__ __forceinline__ SeqElem wave_reduce_elem(SeqElem el) {
  const int lane = threadIdx.x & 63;
  const bool bad = __ballot(el.bad) != 0ull;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const SeqElem o = shfl_down_elem(el, off);
    if ((lane & (2 * off - 1)) == 0) el = seq_compose_int(el, o);
  }
  el.bad = bad;
  return el;
}

// perturb: 0 = guesses from the plain sums; 1 = every 3rd valid guess moved by one binade (failed gap -> legacy
// walk); 2 = every guess invalid (record overflow -> legacy walk, all native); 3 = every other crossing record
// split one element late (failed record checks -> native node).  Test hook: results must not change.
__global__ __launch_bounds__(kBlock) void seq_tables(const float* __restrict__ w, int64_t ring, int64_t head,
                                                     int64_t size, const double* __restrict__ s1g,
                                                     const double* __restrict__ s2g, SeqTab* __restrict__ T1,
                                                     SeqTab* __restrict__ T2, SeqRec* __restrict__ rec,
                                                     int32_t* __restrict__ ctl, int perturb) {
  __shared__ double s1[kFan];
  __shared__ double red[kBlock / 64];
  __shared__ SeqTab tab[kFan];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kL2;
  // plain prefix of the level-2 sums before this node (a guess only: any summation order will do)
  double acc = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += kBlock) acc += s2g[i];
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  if (threadIdx.x < kFan) s1[threadIdx.x] = s1g[(int64_t)blockIdx.x * kFan + threadIdx.x];
  __syncthreads();
  double pre = (red[0] + red[1]) + (red[2] + red[3]);
  for (int i = 0; i < wave * 4; ++i) pre += s1[i];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = wave * 4 + c;
    const float x = ring_load(w, ring, head, size, base + (int64_t)j * kL1 + lane);
    const double end = pre + s1[j];
    int e = seq_guess(pre, end);
    const int64_t node = (int64_t)blockIdx.x * kFan + j;
    if (perturb == 1 && e != kTabInvalid && (node % 3) == 1) e += (node & 4) ? 1 : -1;
    if (perturb == 2) e = kTabInvalid;
    const bool nonzero = (fbits(x) << 1) != 0;
    const bool allzero = __ballot(nonzero) == 0ull;
    SeqTab t;
    if (allzero) {
      t = seq_tab_any();
    } else if (e == kTabInvalid) {
      t = seq_tab_invalid();
      // crossing record: speculative split at the element where the PLAIN prefix leaves the binade of `pre`
      double incl = (double)x;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      const double P = pre + incl;
      int kind = 1, m = 0;
      SeqTab B = seq_tab_invalid(), Cc = seq_tab_invalid();
      float wm = 0.f;
      if (pre > 0 && perturb != 2) {
        const int e0 = dexp(pre);
        const unsigned long long left = __ballot(dexp(P) != e0);
        if (left != 0ull && e0 > 52) {
          m = __ffsll((long long)left) - 1;
          if (perturb == 3 && m < 63 && (node & 1)) m += 1;  // wrong split: the record's checks must catch it
          const int e1 = dexp(__shfl(P, m, 64));
          const int e63 = dexp(__shfl(P, 63, 64));
          if (e1 == e63 && e1 < 0x7ff) {  // exactly one crossing inside the node
            SeqElem eb, ec;
            eb.D0 = eb.D1 = 0, eb.bad = false;
            ec = eb;
            if (lane < m) eb = seq_classify(x, e0);
            if (lane > m) ec = seq_classify(x, e1);
            eb = wave_reduce_elem(eb);
            ec = wave_reduce_elem(ec);
            const bool anyb = __ballot(lane < m && nonzero) != 0ull, anyc = __ballot(lane > m && nonzero) != 0ull;
            B = anyb ? seq_make_tab(eb, e0) : seq_tab_any();
            Cc = anyc ? seq_make_tab(ec, e1) : seq_tab_any();
            wm = __shfl(x, m, 64);
            kind = 0;
          }
        }
      }
      if (lane == 0) {
        const int slot = atomicAdd(&ctl[0], 1);
        if (slot < kMaxRec) {
          SeqRec r;
          r.node = (int32_t)node;
          r.kind = kind;
          r.m = m;
          r.wm = wm;
          r.B = B;
          r.C = Cc;
          rec[slot] = r;
        }
      }
    } else {
      SeqElem el = seq_classify(x, e);
      el = wave_reduce_elem(el);
      t = seq_make_tab(el, e);
    }
    pre = end;
    if (lane == 0) {
      T1[node] = t;
      tab[j] = t;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    SeqTab t = tab[0];
    for (int i = 1; i < kFan; ++i) t = seq_compose(t, tab[i]);
    T2[blockIdx.x] = t;
  }
}

// ---- chain ------------------------------------------------------------------------------------------
__device__ __forceinline__ SeqTab shfl_up_tab(const SeqTab& t, int off) {
  SeqTab o;
  o.d[0] = __shfl_up(t.d[0], off, 64);
  o.d[1] = __shfl_up(t.d[1], off, 64);
  o.e = __shfl_up(t.e, off, 64);
  o.par = __shfl_up(t.par, off, 64);
  return o;
}

// The pieces of a gap [a, b) of level-1 nodes, in order: level-1 tables up to a level-2 boundary, level-2
// tables up to a level-3 boundary, level-3 tables, then level-2 and level-1 tables of the tail.
struct GapCut {
  int64_t a, b;           // level-1 node range
  int64_t h1_end;         // level-1 head: [a, h1_end)
  int64_t p, h2_end;      // level-2 head: [p, h2_end) in level-2 units
  int64_t m3, m3_end;     // level-3 middle
  int64_t t2, t2_end;     // level-2 tail
  int64_t t1, t1_end;     // level-1 tail: [t1, b)
  int64_t total;
};
__device__ __forceinline__ GapCut gap_cut(int64_t a, int64_t b) {
  GapCut g;
  g.a = a, g.b = b;
  const int64_t A16 = (a + kFan - 1) / kFan * kFan;
  if (A16 >= b) {  // no whole level-2 node inside (or exactly aligned and empty)
    g.h1_end = b;
    g.p = g.h2_end = g.m3 = g.m3_end = g.t2 = g.t2_end = 0;
    g.t1 = g.t1_end = b;
    g.total = b - a;
    return g;
  }
  g.h1_end = A16;
  const int64_t B16 = b / kFan * kFan;
  g.t1 = B16, g.t1_end = b;
  const int64_t p = A16 / kFan, q = B16 / kFan;  // level-2 range [p, q)
  g.p = p;
  const int64_t P16 = (p + kFan - 1) / kFan * kFan;
  if (P16 >= q) {
    g.h2_end = q;
    g.m3 = g.m3_end = 0;
    g.t2 = g.t2_end = q;
  } else {
    g.h2_end = P16;
    const int64_t Q16 = q / kFan * kFan;
    g.m3 = P16 / kFan, g.m3_end = Q16 / kFan;
    g.t2 = Q16, g.t2_end = q;
  }
  g.total = (g.h1_end - g.a) + (g.h2_end - g.p) + (g.m3_end - g.m3) + (g.t2_end - g.t2) + (g.t1_end - g.t1);
  return g;
}
__device__ __forceinline__ SeqTab gap_piece(const GapCut& g, int64_t q, const SeqTab* __restrict__ T1,
                                            const SeqTab* __restrict__ T2, const SeqTab* sT3) {
  int64_t n = g.h1_end - g.a;
  if (q < n) return T1[g.a + q];
  q -= n;
  n = g.h2_end - g.p;
  if (q < n) return T2[g.p + q];
  q -= n;
  n = g.m3_end - g.m3;
  if (q < n) return sT3[g.m3 + q];
  q -= n;
  n = g.t2_end - g.t2;
  if (q < n) return T2[g.t2 + q];
  q -= n;
  return T1[g.t1 + q];
}

struct ChainArgs {
  const float* w;
  int64_t ring, head, size;
  SeqTab* T1;
  const SeqTab* T2;
  SeqTab* T3;
  double *A3, *A2;
  const SeqRec* rec;
  int32_t* ctl;
  int n2, n3;
  SeqTargetsJob tj;
};

// legacy exact walk inside one level-3 node from the exact accumulator (wave-cooperative, uniform)
__device__ inline double legacy_walk_l3(const ChainArgs& a, double A, int64_t j) {
  const int lane = threadIdx.x & 63;
  const SeqRingAt at{a.w, a.ring, a.head, a.size};
  const int64_t n1 = (int64_t)a.n2 * kFan;
  SeqTab my2 = seq_tab_any();
  {
    const int64_t i2 = j * kFan + lane;
    if (lane < kFan && i2 < a.n2) my2 = a.T2[i2];
  }
  for (int c2 = 0; c2 < kFan; ++c2) {
    double n;
    const SeqTab t2 = seq_tab_bcast(my2, c2);
    if (seq_apply(t2, A, &n)) {
      A = n;
      continue;
    }
    const int64_t u0 = (j * kFan + c2) * kFan;
    SeqTab my1 = seq_tab_any();
    if (lane < kFan && u0 + lane < n1) my1 = a.T1[u0 + lane];
    for (int c1 = 0; c1 < kFan; ++c1) {
      const SeqTab t1 = seq_tab_bcast(my1, c1);
      if (seq_apply(t1, A, &n)) {
        A = n;
        continue;
      }
      const int64_t k0 = (u0 + c1) * kL1;
      const float wl = at(k0 + lane);
      A = seq_native_wave(wl, A, k0, kL1, (double)INFINITY).A;
    }
  }
  return A;
}

__global__ __launch_bounds__(kChainThreads) void seq_chain(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ SeqRec sRec[kMaxRec];        // sorted by node
  __shared__ int32_t sNode[kMaxRec + 1];  // sorted crossing nodes
  __shared__ SeqTab sGap[kMaxRec + 1];    // composed table of gap g
  __shared__ SeqTab sHead[kMaxRec + 1];   // ... of its level-1 head (up to the first level-2 boundary inside)
  __shared__ double sGapA[kMaxRec + 2];   // exact accumulator at the start of gap g (= after crossing g-1)
  __shared__ double sCrossA[kMaxRec];     // ... entering crossing node g
  __shared__ int sFail, sNrec;
  const int n2 = a.n2, n3 = a.n3;
  const bool t2_lds = n2 <= kT2Lds;
  SeqTab* sT3 = reinterpret_cast<SeqTab*>(smem);
  SeqTab* sT2 = sT3 + n3;
  double* sRun = reinterpret_cast<double*>(smem + ((((size_t)n3 + (t2_lds ? n2 : 0)) * sizeof(SeqTab) + 15) & ~(size_t)15));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n1 = (int64_t)n2 * kFan;  // level-1 tables that exist (zero padded to whole level-2 nodes)
  const SeqRingAt at{a.w, a.ring, a.head, a.size};
  const SeqTab* T2 = a.T2;
  if (t2_lds) {  // 24 B per level-2 node: 30 KB at ring 1,310,720
    const int words = n2 * (int)(sizeof(SeqTab) / 4);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(a.T2);
    uint32_t* dst = reinterpret_cast<uint32_t*>(sT2);
    for (int i = tid; i < words; i += kChainThreads) dst[i] = src[i];
    T2 = sT2;
  }
  if (tid == 0) {
    const int nr = a.ctl[0];
    sNrec = nr <= kMaxRec ? nr : 0;
    sFail = nr > kMaxRec ? 1 : 0;
  }
  __syncthreads();
  // ---- phase 0: level-3 tables
  for (int j = tid; j < n3; j += kChainThreads) {
    SeqTab t = seq_tab_any();
    for (int c = 0; c < kFan; ++c) {
      const int i = j * kFan + c;
      if (i < n2) t = seq_compose(t, T2[i]);
    }
    sT3[j] = t;
    a.T3[j] = t;
  }
  const int nrec = sNrec;
  // ---- phase 1: records sorted by node (rank by counting; nodes are distinct)
  if (!sFail && tid < nrec) {
    const int32_t mine = a.rec[tid].node;
    int rank = 0;
    for (int i = 0; i < nrec; ++i) rank += a.rec[i].node < mine;
    sRec[rank] = a.rec[tid];
    sNode[rank] = mine;
  }
  __syncthreads();
  // ---- phase 2: one composed table per gap (gap g = level-1 nodes strictly between crossing g-1 and g), and
  // the table of its level-1 head, which leads to the first level-2 boundary inside the gap
  if (!sFail) {
    for (int g = wave; g <= nrec; g += kChainThreads / 64) {
      const int64_t ga = g == 0 ? 0 : (int64_t)sNode[g - 1] + 1;
      const int64_t gb = g == nrec ? n1 : (int64_t)sNode[g];
      SeqTab carry = seq_tab_any(), head = seq_tab_any();
      if (gb > ga) {
        const GapCut cut = gap_cut(ga, gb);
        const int nh1 = (int)(cut.h1_end - cut.a);  // < 16, or the whole (short) gap
        for (int64_t base = 0; base < cut.total; base += 64) {
          SeqTab t = seq_tab_any();
          if (base + lane < cut.total) t = gap_piece(cut, base + lane, a.T1, T2, sT3);
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const SeqTab o = shfl_up_tab(t, off);
            if (lane >= off) t = seq_compose(o, t);
          }
          if (base == 0 && nh1 > 0 && nh1 <= 64) head = seq_tab_bcast(t, nh1 - 1);
          carry = seq_compose(carry, seq_tab_bcast(t, 63));
        }
      }
      if (lane == 0) {
        sGap[g] = carry;
        sHead[g] = head;
      }
    }
  }
  __syncthreads();
  // ---- phase 3: the sequential part, one wavefront, uniform control flow
  if (wave == 0 && !sFail) {
    double A = 0;
    bool fail = false;
    for (int g = 0; g <= nrec; ++g) {
      if (lane == 0) sGapA[g] = A;
      double n;
      if (!seq_apply(sGap[g], A, &n)) {
        fail = true;
        break;
      }
      A = n;
      if (g == nrec) break;
      if (lane == 0) sCrossA[g] = A;
      const SeqRec& r = sRec[g];
      bool done = false;
      if (r.kind == 0) {
        double b, c;
        if (seq_apply(r.B, A, &b)) {
          const double mid = b + (double)r.wm;  // the crossing add itself: one native rounding
          if (seq_apply(r.C, mid, &c)) {
            A = c;
            done = true;
          }
        }
      }
      if (!done) {  // exact fallback: the 64 native adds of the node
        const int64_t k0 = (int64_t)r.node * kL1;
        const float wl = at(k0 + lane);
        A = seq_native_wave(wl, A, k0, kL1, (double)INFINITY).A;
      }
    }
    if (lane == 0) {
      sGapA[nrec + 1] = A;
      if (fail) sFail = 1;
    }
  }
  __syncthreads();
  // ---- phase 4a: exact values at the level-2 / level-3 boundaries inside each gap, one thread per gap.
  // Whole level-3 nodes are jumped with their table; their level-2 boundaries are filled in phase 4b.
  if (!sFail) {
    for (int g = tid; g <= nrec; g += kChainThreads) {
      const int64_t ga = g == 0 ? 0 : (int64_t)sNode[g - 1] + 1;
      const int64_t gb = g == nrec ? n1 : (int64_t)sNode[g];
      double A = sGapA[g];
      bool bad = false;
      int64_t p = (ga + kFan - 1) / kFan * kFan;  // first level-2 boundary at or after the gap start
      if (p <= gb) {
        double n;
        if (p > ga) {  // through the level-1 head
          bad = !seq_apply(sHead[g], A, &n);
          if (!bad) A = n;
        }
        while (!bad) {
          a.A2[p / kFan] = A;
          if (p % (kFan * kFan) == 0) a.A3[p / (kFan * kFan)] = A;
          if (p + kFan > gb) break;  // no further whole level-2 node inside the gap
          if (p % (kFan * kFan) == 0 && p + kFan * kFan <= gb) {
            bad = !seq_apply(sT3[p / (kFan * kFan)], A, &n);
            p += kFan * kFan;
          } else {
            bad = !seq_apply(T2[p / kFan], A, &n);
            p += kFan;
          }
          if (!bad) A = n;
        }
      }
      if (bad) sFail = 1;  // cannot happen when the gap table applied; kept as a guard
    }
    // Crossing nodes get an EXACT table (entry value -> exit value): the per-stratum walks of the search and
    // the prefix of blockPop then pass them with one comparison instead of 64 native adds.
    for (int g = tid; g < nrec; g += kChainThreads) {
      SeqTab t;
      t.d[0] = sCrossA[g];
      t.d[1] = sGapA[g + 1];
      t.e = kTabExact;
      t.par = 0;
      a.T1[sNode[g]] = t;
    }
    if (tid == 0) {  // padding boundaries past the last level-2 node: the total
      const double total = sGapA[nrec + 1];
      for (int64_t i = n2; i <= (int64_t)n3 * kFan; ++i) a.A2[i] = total;
      a.A3[n3] = total;
    }
  }
  __syncthreads();
  // ---- legacy path: record overflow or a failed gap -- one wavefront walks the level-3 nodes
  const bool legacy = sFail != 0;
  if (legacy && wave == 0) {
    double A = 0;
    for (int j = 0; j < n3; ++j) {
      if (lane == 0) a.A3[j] = A;
      double n;
      if (seq_apply(sT3[j], A, &n)) A = n;
      else A = legacy_walk_l3(a, A, j);
    }
    if (lane == 0) {
      a.A3[n3] = A;
      atomicAdd(&a.ctl[1], 1);
    }
  }
  __syncthreads();
  // ---- phase 4b: level-2 boundaries inside the level-3 nodes that were taken as a whole (their table is
  // valid, so every child applies); in the legacy path every node, with a generic exact walk
  for (int j = tid; j < n3; j += kChainThreads) {
    if (!legacy && sT3[j].e == kTabInvalid) continue;  // its boundaries lie in gap heads / tails: done in 4a
    double A = a.A3[j];
    for (int c = 0; c < kFan; ++c) {
      const int64_t i = (int64_t)j * kFan + c;
      a.A2[i] = A;
      double n;
      if (i >= n2) continue;
      if (seq_apply(T2[i], A, &n)) {
        A = n;
        continue;
      }
      for (int d = 0; d < kFan; ++d) {  // legacy only
        const int64_t u = i * kFan + d;
        if (seq_apply(a.T1[u], A, &n)) {
          A = n;
          continue;
        }
        for (int e = 0; e < kL1; ++e) A += (double)at(u * kL1 + e);
      }
    }
  }
  if (legacy && tid == 0) a.A2[(int64_t)n3 * kFan] = a.A3[n3];

  // ---- second job: stratified targets of a sample (prioritized_replay.h:261-280)
  // libstdc++'s uniform_real_distribution<float>(0, segment) is canonical*(segment-0)+0 with canonical =
  // float(u32)/2^32 clamped below 1 (oracle/mt19937.c restates it).  The reference scans once for all targets
  // and never moves backwards, so the effective target of sample i is max(rand_0..rand_i); a non-positive
  // target means "first acc > 0".
  const int batch = a.tj.batch;
  if (batch > 0) {
    const float sum = (float)a.tj.state->sum;
    const float segment = sum / (float)batch;
    const float cap = sum - 0.2f;
    for (int i = tid; i < batch; i += kChainThreads) {
      float c = (float)a.tj.draws[i] * 2.3283064365386963e-10f;  // exact scaling by 2^-32
      if (c >= 1.0f) c = 0.99999994f;
      const float u = c * segment + 0.0f;
      const float off = (float)i * segment;
      float r = u + off;
      r = (r < cap) ? r : cap;  // std::min(sum - 0.2f, rand)
      a.tj.targets[i] = r;
      sRun[i] = fmax((double)r, 4.9406564584124654e-324);  // denorm_min: acc >= it  <=>  acc > 0
    }
    __syncthreads();
    for (int off = 1; off < batch; off <<= 1) {  // inclusive prefix maximum (Hillis-Steele; batch <= 4096)
      double v[4];
      int c = 0;
      for (int i = tid; i < batch; i += kChainThreads, ++c) v[c] = (i >= off) ? fmax(sRun[i], sRun[i - off]) : sRun[i];
      __syncthreads();
      c = 0;
      for (int i = tid; i < batch; i += kChainThreads, ++c) sRun[i] = v[c];
      __syncthreads();
    }
    for (int i = tid; i < batch; i += kChainThreads) a.tj.eff[i] = sRun[i];
    if (tid == 0) a.tj.state->sum_f = sum;
  }
}

// one wavefront per target
__global__ __launch_bounds__(64) void seq_search_kernel(SeqView v, const double* __restrict__ targets, int nt,
                                                        int64_t* __restrict__ out_k, double* __restrict__ out_A,
                                                        float* __restrict__ out_w) {
  const int i = blockIdx.x;
  if (i >= nt) return;
  const SeqHit h = seq_find_wave(v, targets[i]);
  if ((threadIdx.x & 63) == 0) {
    out_k[i] = h.found ? h.k : -1;
    out_A[i] = h.A;
    out_w[i] = h.w;
  }
}

int g_perturb = 0;  // test hook, see seq_tables

}  // namespace

int seq_index_alloc(SeqIndex* ix, int64_t max_elems) {
  const int n2 = ceil_div(max_elems, kL2) + 1;
  const int n3 = ceil_div(n2, kFan) + 1;
  ix->n2cap = n2;
  ix->n3cap = n3;
  RELA_CHECK(n3 <= 2048, RELA_EINVAL, "seq_index_alloc: ring of %lld weights exceeds the index (2^25)",
             (long long)max_elems);
  RELA_HIP(hipMalloc(&ix->s2, sizeof(double) * n2));
  RELA_HIP(hipMalloc(&ix->s1, sizeof(double) * (size_t)n2 * kFan));
  RELA_HIP(hipMalloc(&ix->T1, sizeof(SeqTab) * (size_t)n2 * kFan));
  RELA_HIP(hipMalloc(&ix->T2, sizeof(SeqTab) * n2));
  RELA_HIP(hipMalloc(&ix->T3, sizeof(SeqTab) * n3));
  RELA_HIP(hipMalloc(&ix->A3, sizeof(double) * (n3 + 1)));
  RELA_HIP(hipMalloc(&ix->A2, sizeof(double) * ((size_t)n3 * kFan + 1)));
  RELA_HIP(hipMalloc(&ix->rec, sizeof(SeqRec) * kMaxRec));
  RELA_HIP(hipMalloc(&ix->ctl, sizeof(int32_t) * 4));
  RELA_HIP(hipMemset(ix->ctl, 0, sizeof(int32_t) * 4));
  static bool attr_set = false;
  if (!attr_set) {
    RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&seq_chain), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (kT2Lds + 256) * (int)sizeof(SeqTab) + 16 + 4096 * (int)sizeof(double)));
    attr_set = true;
  }
  return RELA_OK;
}

void seq_index_free(SeqIndex* ix) {
  (void)hipFree(ix->s2);
  (void)hipFree(ix->s1);
  (void)hipFree(ix->T1);
  (void)hipFree(ix->T2);
  (void)hipFree(ix->T3);
  (void)hipFree(ix->A3);
  (void)hipFree(ix->A2);
  (void)hipFree(ix->rec);
  (void)hipFree(ix->ctl);
  *ix = SeqIndex();
}

int seq_index_build(const SeqIndex& ix, const float* ring_dev, int64_t ring, int64_t head, int64_t size,
                    hipStream_t stream, SeqView* view, const SeqTargetsJob* targets) {
  RELA_CHECK(size >= 0 && size <= ring && head >= 0 && head < (ring > 0 ? ring : 1), RELA_EINVAL,
             "seq_index_build: bad range head=%lld size=%lld ring=%lld", (long long)head,
             (long long)size, (long long)ring);
  const int n2 = ceil_div(size, kL2);
  const int n3 = ceil_div(size, kL3);
  RELA_CHECK(n2 <= ix.n2cap && n3 <= ix.n3cap, RELA_EINVAL, "seq_index_build: index too small");
  RELA_CHECK(!targets || targets->batch <= 4096, RELA_EINVAL, "seq_index_build: batch too large");
  if (n2 > 0) {
    {
      ProfScope prof("seq_sums", stream);
      hipLaunchKernelGGL(seq_sums, dim3(n2), dim3(kBlock), 0, stream, ring_dev, ring, head, size, ix.s1, ix.s2, ix.ctl);
    }
    {
      ProfScope prof("seq_tables", stream);
      hipLaunchKernelGGL(seq_tables, dim3(n2), dim3(kBlock), 0, stream, ring_dev, ring, head, size,
                         (const double*)ix.s1, (const double*)ix.s2, ix.T1, ix.T2, ix.rec, ix.ctl, g_perturb);
    }
  } else {
    RELA_HIP(hipMemsetAsync(ix.ctl, 0, sizeof(int32_t), stream));
  }
  {
    ChainArgs a;
    a.w = ring_dev, a.ring = ring, a.head = head, a.size = size;
    a.T1 = ix.T1, a.T2 = ix.T2, a.T3 = ix.T3, a.A3 = ix.A3, a.A2 = ix.A2;
    a.rec = ix.rec, a.ctl = ix.ctl, a.n2 = n2, a.n3 = n3;
    if (targets) a.tj = *targets;
    const size_t ntab = (size_t)n3 + (n2 <= kT2Lds ? (size_t)n2 : 0);
    const size_t lds = ((ntab * sizeof(SeqTab) + 15) & ~(size_t)15) + sizeof(double) * (size_t)a.tj.batch + 16;
    ProfScope prof("seq_chain", stream);
    hipLaunchKernelGGL(seq_chain, dim3(1), dim3(kChainThreads), lds, stream, a);
  }
  RELA_LAUNCH_CHECK();
  view->w = ring_dev;
  view->ring = ring;
  view->head = head;
  view->size = size;
  view->T1 = ix.T1;
  view->A2 = ix.A2;
  view->A3 = ix.A3;
  view->n3 = n3;
  return RELA_OK;
}

}  // namespace rela_amd

using namespace rela_amd;

// test hook: 0 = normal, 1 = perturbed binade guesses, 2 = every guess invalid.  Exactness must not depend on it.
extern "C" int rela_seqscan_debug_perturb(int mode) {
  g_perturb = mode;
  return RELA_OK;
}

extern "C" int rela_seqscan_search(const float* ring_dev, int64_t ring, int64_t head, int64_t size,
                                   const double* targets_host, int nt, int64_t* out_index,
                                   double* out_acc, float* out_w, double* out_total, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  RELA_CHECK(ring_dev && ring > 0 && nt >= 0, RELA_EINVAL, "rela_seqscan_search: bad arguments");
  SeqIndex ix;
  int rc = seq_index_alloc(&ix, size > 0 ? size : 1);
  if (rc != RELA_OK) return rc;
  SeqView v;
  rc = seq_index_build(ix, ring_dev, ring, head, size, stream, &v);
  if (rc != RELA_OK) {
    seq_index_free(&ix);
    return rc;
  }
  double* d_t = nullptr;
  int64_t* d_k = nullptr;
  double* d_A = nullptr;
  float* d_w = nullptr;
  const int n = nt > 0 ? nt : 1;
  RELA_HIP(hipMalloc(&d_t, sizeof(double) * n));
  RELA_HIP(hipMalloc(&d_k, sizeof(int64_t) * n));
  RELA_HIP(hipMalloc(&d_A, sizeof(double) * n));
  RELA_HIP(hipMalloc(&d_w, sizeof(float) * n));
  if (nt > 0) {
    RELA_HIP(hipMemcpyAsync(d_t, targets_host, sizeof(double) * nt, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(seq_search_kernel, dim3(nt), dim3(64), 0, stream, v, d_t, nt, d_k, d_A, d_w);
    RELA_LAUNCH_CHECK();
    RELA_HIP(hipMemcpyAsync(out_index, d_k, sizeof(int64_t) * nt, hipMemcpyDeviceToHost, stream));
    RELA_HIP(hipMemcpyAsync(out_acc, d_A, sizeof(double) * nt, hipMemcpyDeviceToHost, stream));
    RELA_HIP(hipMemcpyAsync(out_w, d_w, sizeof(float) * nt, hipMemcpyDeviceToHost, stream));
  }
  if (out_total) RELA_HIP(hipMemcpyAsync(out_total, ix.A3 + v.n3, sizeof(double), hipMemcpyDeviceToHost, stream));
  RELA_HIP(hipStreamSynchronize(stream));
  (void)hipFree(d_t);
  (void)hipFree(d_k);
  (void)hipFree(d_A);
  (void)hipFree(d_w);
  seq_index_free(&ix);
  return RELA_OK;
}
