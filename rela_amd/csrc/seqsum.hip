// seqsum.hip -- kernels that build the exact sequential-f64-sum index over a weight ring.
//
// Restates the accumulation of rela/prioritized_replay.h:266-308 (sample_ scan) and :85-95
// (blockPop diff) for a GPU: see seqsum_core.h for the arithmetic and the proof sketch.
//
// Pipeline (all stream-ordered, no host sync), N = live weights, HBM/L2 bound:
//   seq_l2_sums     grid N/1024   reads 4N B      plain f64 sums per level-2 node (guess only)
//   seq_l2_scan     1 block       -               exclusive prefix of those sums
//   seq_tables      grid N/1024   reads 4N B      level-1 (64) and level-2 (1024) transfer tables
//   seq_l3_tables   grid N/16384  -               level-3 tables from level-2
//   seq_chain       1 wave        -               exact accumulator before every level-3 node
// Algorithmic bytes: 4N (the reference's linear scan reads each weight once);
// this pipeline reads them twice, the second time from L2 / Infinity Cache.
#include "common.h"
#include "prof.h"
#include "seqsum_dev.h"

namespace rela_amd {

namespace {

constexpr int kBlock = 256;  // 4 wavefronts; one workgroup per level-2 node (1024 weights)

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

__global__ __launch_bounds__(kBlock) void seq_l2_sums(const float* __restrict__ w, int64_t ring,
                                                      int64_t head, int64_t size,
                                                      double* __restrict__ bsum2) {
  __shared__ double part[kBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kL2;
  double s = 0;
#pragma unroll
  for (int c = 0; c < kL2 / kBlock; ++c) s += (double)ring_load(w, ring, head, size, base + c * kBlock + threadIdx.x);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bsum2[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(kBlock) void seq_l2_scan(const double* __restrict__ bsum2, int n2,
                                                      double* __restrict__ S0) {
  __shared__ double chunk[kBlock];
  const int per = (n2 + kBlock - 1) / kBlock;
  const int lo = threadIdx.x * per;
  const int hi = min(lo + per, n2);
  double s = 0;
  for (int i = lo; i < hi; ++i) s += bsum2[i];
  chunk[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double acc = 0;
    for (int i = 0; i < kBlock; ++i) {
      const double t = chunk[i];
      chunk[i] = acc;
      acc += t;
    }
    S0[n2] = acc;
  }
  __syncthreads();
  double acc = chunk[threadIdx.x];
  for (int i = lo; i < hi; ++i) {
    S0[i] = acc;
    acc += bsum2[i];
  }
}

__device__ __forceinline__ SeqElem shfl_down_elem(const SeqElem& e, int off) {
  SeqElem o;
  o.D0 = __shfl_down((long long)e.D0, off, 64);
  o.D1 = __shfl_down((long long)e.D1, off, 64);
  o.bad = false;
  return o;
}

__global__ __launch_bounds__(kBlock) void seq_tables(const float* __restrict__ w, int64_t ring,
                                                     int64_t head, int64_t size,
                                                     const double* __restrict__ S0,
                                                     SeqTab* __restrict__ T1, SeqTab* __restrict__ T2) {
  __shared__ double s1[kFan];
  __shared__ SeqTab tab[kFan];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kL2;
  float x[4];
  // phase A: plain sums of my four level-1 nodes -> binade guesses
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = wave * 4 + c;
    x[c] = ring_load(w, ring, head, size, base + (int64_t)j * kL1 + lane);
    const double s = wave_sum((double)x[c]);
    if (lane == 0) s1[j] = s;
  }
  __syncthreads();
  double pre = S0[blockIdx.x];
  for (int i = 0; i < wave * 4; ++i) pre += s1[i];
  // phase B: transfer tables
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = wave * 4 + c;
    const double end = pre + s1[j];
    const int e = seq_guess(pre, end);
    pre = end;
    const bool nonzero = (fbits(x[c]) << 1) != 0;
    const bool allzero = __ballot(nonzero) == 0ull;
    SeqTab t;
    if (allzero) {
      t = seq_tab_any();
    } else if (e == kTabInvalid) {
      t = seq_tab_invalid();
    } else {
      SeqElem el = seq_classify(x[c], e);
      const bool bad = __ballot(el.bad) != 0ull;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const SeqElem o = shfl_down_elem(el, off);
        if ((lane & (2 * off - 1)) == 0) el = seq_compose_int(el, o);
      }
      el.bad = bad;
      t = seq_make_tab(el, e);
    }
    if (lane == 0) {
      T1[(int64_t)blockIdx.x * kFan + j] = t;
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

__global__ void seq_l3_tables(const SeqTab* __restrict__ T2, int n2, SeqTab* __restrict__ T3, int n3) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n3) return;
  SeqTab t = seq_tab_any();
  for (int c = 0; c < kFan; ++c) {
    const int i = j * kFan + c;
    if (i < n2) t = seq_compose(t, T2[i]);
  }
  T3[j] = t;
}

// One WAVEFRONT walks the level-3 nodes in order carrying the exact accumulator: 64 level-3
// tables are fetched per load and tried through lane broadcasts; a node whose table cannot be
// applied (a binade crossing) is walked cooperatively at the finer levels.
__global__ __launch_bounds__(64) void seq_chain(const float* __restrict__ w, int64_t ring, int64_t head,
                                                int64_t size, const SeqTab* __restrict__ T1,
                                                const SeqTab* __restrict__ T2, const SeqTab* __restrict__ T3,
                                                int n3, double* __restrict__ A3) {
  const int lane = threadIdx.x & 63;
  SeqView v;
  v.w = w;
  v.ring = ring;
  v.head = head;
  v.size = size;
  v.T1 = T1;
  v.T2 = T2;
  v.A3 = A3;
  v.n3 = n3;
  double A = 0;
  for (int base = 0; base < n3; base += 64) {
    SeqTab mine = seq_tab_invalid();
    if (base + lane < n3) mine = T3[base + lane];
    const int cnt = (n3 - base < 64) ? n3 - base : 64;
    for (int c = 0; c < cnt; ++c) {
      const int j = base + c;
      if (lane == 0) A3[j] = A;
      const SeqTab t3 = seq_tab_bcast(mine, c);
      double n;
      if (seq_apply(t3, A, &n)) {  // level-3 tables are zero padded: always safe to apply
        A = n;
        continue;
      }
      int64_t kend = (int64_t)(j + 1) * kL3;
      if (kend > size) kend = size;
      A = seq_walk_wave(v, A, (int64_t)j * kL3, kend, (double)INFINITY, true).A;
    }
  }
  if (lane == 0) A3[n3] = A;
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

}  // namespace

int seq_index_alloc(SeqIndex* ix, int64_t max_elems) {
  const int n2 = ceil_div(max_elems, kL2) + 1;
  const int n3 = ceil_div(n2, kFan) + 1;
  ix->n2cap = n2;
  ix->n3cap = n3;
  RELA_HIP(hipMalloc(&ix->bsum2, sizeof(double) * n2));
  RELA_HIP(hipMalloc(&ix->S0, sizeof(double) * (n2 + 1)));
  RELA_HIP(hipMalloc(&ix->T1, sizeof(SeqTab) * (size_t)n2 * kFan));
  RELA_HIP(hipMalloc(&ix->T2, sizeof(SeqTab) * n2));
  RELA_HIP(hipMalloc(&ix->T3, sizeof(SeqTab) * n3));
  RELA_HIP(hipMalloc(&ix->A3, sizeof(double) * (n3 + 1)));
  return RELA_OK;
}

void seq_index_free(SeqIndex* ix) {
  (void)hipFree(ix->bsum2);
  (void)hipFree(ix->S0);
  (void)hipFree(ix->T1);
  (void)hipFree(ix->T2);
  (void)hipFree(ix->T3);
  (void)hipFree(ix->A3);
  *ix = SeqIndex();
}

int seq_index_build(const SeqIndex& ix, const float* ring_dev, int64_t ring, int64_t head,
                    int64_t size, hipStream_t stream, SeqView* view) {
  RELA_CHECK(size >= 0 && size <= ring && head >= 0 && head < (ring > 0 ? ring : 1), RELA_EINVAL,
             "seq_index_build: bad range head=%lld size=%lld ring=%lld", (long long)head,
             (long long)size, (long long)ring);
  const int n2 = ceil_div(size, kL2);
  const int n3 = ceil_div(size, kL3);
  RELA_CHECK(n2 <= ix.n2cap && n3 <= ix.n3cap, RELA_EINVAL, "seq_index_build: index too small");
  if (n2 > 0) {
    {
      ProfScope prof("seq_l2_sums", stream);
      hipLaunchKernelGGL(seq_l2_sums, dim3(n2), dim3(kBlock), 0, stream, ring_dev, ring, head, size, ix.bsum2);
    }
    {
      ProfScope prof("seq_l2_scan", stream);
      hipLaunchKernelGGL(seq_l2_scan, dim3(1), dim3(kBlock), 0, stream, ix.bsum2, n2, ix.S0);
    }
    {
      ProfScope prof("seq_tables", stream);
      hipLaunchKernelGGL(seq_tables, dim3(n2), dim3(kBlock), 0, stream, ring_dev, ring, head, size, ix.S0, ix.T1,
                         ix.T2);
    }
    {
      ProfScope prof("seq_l3_tables", stream);
      hipLaunchKernelGGL(seq_l3_tables, dim3(ceil_div(n3, 64)), dim3(64), 0, stream, ix.T2, n2, ix.T3, n3);
    }
  }
  {
    ProfScope prof("seq_chain", stream);
    hipLaunchKernelGGL(seq_chain, dim3(1), dim3(64), 0, stream, ring_dev, ring, head, size, ix.T1, ix.T2, ix.T3, n3,
                       ix.A3);
  }
  RELA_LAUNCH_CHECK();
  view->w = ring_dev;
  view->ring = ring;
  view->head = head;
  view->size = size;
  view->T1 = ix.T1;
  view->T2 = ix.T2;
  view->A3 = ix.A3;
  view->n3 = n3;
  return RELA_OK;
}

}  // namespace rela_amd

using namespace rela_amd;

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
