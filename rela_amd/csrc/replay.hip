// replay.hip -- device-resident prioritized replay behind the C ABI of include/rela_amd.h.
//
// Restates rela/prioritized_replay.h: ConcurrentQueue (:14-171) + PrioritizedReplay (:173-348).
//
// Split of responsibilities
//   host  : the integer bookkeeping of ConcurrentQueue (head_/tail_/size_, back-pressure on
//           cvSize_ :47, the one-outstanding-batch protocol :203-206) and the std::mt19937
//           that draws the stratified samples (:267,279,346).  The host never reads weights.
//   device: the f32 weight ring, the evicted flags, the f64 running sum_ and every float /
//           double operation on them, in the reference's order:
//             add      w = pow(p, alpha) :188; float block sum added to the f64 sum_ :58-66,73
//             sample_  sum narrowed to float :30-36; targets u*segment + i*segment clamped to
//                      sum - 0.2f :264-280; sequential f64 scan :282-308 (seqsum_core.h);
//                      eviction diff :85-95; IS weights :320-322; batch gather (types.cc:8-46)
//             update   diff += (new - old) in float, accumulated in f64 :105-119
//   All device work of one replay is serialised on its own stream in the order the host
//   committed it, which is exactly the in-slot-order commit the reference enforces with
//   cvTail_ (:69-74); producers/consumers are tied in with events, never with host syncs.
//
// HBM layout: weights f32[ring], evicted u8[ring], field f rows at base_f + slot*row_bytes_f.
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <random>
#include <vector>

#include "common.h"
#include "prof.h"
#include "seqsum_dev.h"
#include "sleef_powf_core.h"
#include "vmm_field.h"

namespace rela_amd {

static thread_local char g_err[512] = "";
void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

constexpr int kMaxBatch = 4096;
constexpr int kThreads = 256;

// torch::pow(tensor, exponent) as ATen's CPU kernel evaluates it for a float tensor of n elements and a
// float exponent held in a double Scalar (rela/prioritized_replay.h:188,239,321): the vectorised loop takes
// 2 * Vec::size() = 32 floats per iteration through SLEEF's powf (u10) with the exponent as float; the
// remaining n % 32 elements go through the scalar lambda, std::pow(float, double) -> double pow, rounded to
// float.  Exponent 1 is a copy.  pos = index inside that tensor, n_ref = its length.
__device__ __forceinline__ float pow_aten(float p, float ex, int pos, int n_ref) {
  if (ex == 1.0f) return p;
  // exponents ATen evaluates without pow (pow_tensor_scalar_optimized_kernel); beta = 1 gives the reciprocal.
  // (+-0.5 go through ATen's vectorised sqrt, which is not the IEEE square root in the last bit on every input:
  // the device uses the IEEE one; no BASELINE config uses those exponents and they are not pinned.)
  if (ex == -1.0f) return __fdiv_rn(1.0f, p);
  if (ex == 0.5f) return __fsqrt_rn(p);
  if (ex == 2.0f) return __fmul_rn(p, p);
  if (ex == 3.0f) return __fmul_rn(__fmul_rn(p, p), p);
  if (ex == -0.5f) return __fdiv_rn(1.0f, __fsqrt_rn(p));
  if (ex == -2.0f) return __fdiv_rn(1.0f, __fmul_rn(p, p));
  if (pos < n_ref - (n_ref & 31)) return sleef::powf_u10(p, ex);
  return (float)pow((double)p, (double)ex);
}

// ---- add ------------------------------------------------------------------------------
// blockAppend :57-66,73.  One workgroup: weights in parallel, then lane 0 accumulates the
// block sum in FLOAT in slot order (exactly `sum += weightAcc[i]`) and adds it to sum_.
// `group` > 0: the n slots are n/group consecutive reference blocks (one per batched actor thread);
// each block's float sum is added to sum_ separately, in order.
__global__ __launch_bounds__(kThreads) void replay_append_weights(const float* __restrict__ prio, int n,
                                                                  float alpha, float* __restrict__ w,
                                                                  int ring, int start, int group,
                                                                  ReplayDevState* __restrict__ st) {
  // One reference block = `group` rows: float sum in row order, then sum_ += (double)sum (:64-73).
  // Blocks are independent until the double accumulation, so one thread sums one block and
  // thread 0 folds the block sums in order; a block longer than the LDS chunk stays serial.
  constexpr int kCap = 4096;
  __shared__ float chunk[kCap];
  __shared__ float gsum[kCap];
  double dsum = 0.0;  // thread 0 only
  if (threadIdx.x == 0) dsum = st->sum;
  const int g = group > 0 ? group : n;
  if (g <= kCap) {
    const int cs = (kCap / g) * g;
    for (int base = 0; base < n; base += cs) {
      const int m = min(cs, n - base);
      for (int i = threadIdx.x; i < m; i += kThreads) {
        const int gi = (base + i) / g;                       // reference block of this row
        const int glen = min(g, n - gi * g);                 // its length (the last one may be partial)
        const float v = pow_aten(prio[base + i], alpha, (base + i) - gi * g, glen);
        chunk[i] = v;
        w[(int)(((int64_t)start + base + i) % ring)] = v;
      }
      __syncthreads();
      const int ng = (m + g - 1) / g;  // a trailing partial block counts as a block
      for (int j = threadIdx.x; j < ng; j += kThreads) {
        const int lo = j * g, hi = min(lo + g, m);
        float f = 0.f;
        for (int i = lo; i < hi; ++i) f += chunk[i];
        gsum[j] = f;
      }
      __syncthreads();
      if (threadIdx.x == 0)
        for (int j = 0; j < ng; ++j) dsum += (double)gsum[j];
      __syncthreads();
    }
  } else {
    float fsum = 0.f;
    for (int base = 0; base < n; base += kCap) {
      const int m = min(kCap, n - base);
      for (int i = threadIdx.x; i < m; i += kThreads) {
        const int gi = (base + i) / g;
        const int glen = min(g, n - gi * g);
        const float v = pow_aten(prio[base + i], alpha, (base + i) - gi * g, glen);
        chunk[i] = v;
        w[(int)(((int64_t)start + base + i) % ring)] = v;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        int i = 0;
        while (i < m) {
          const int to_boundary = g - (base + i) % g;  // rows left in the current block
          const int end = min(m, i + to_boundary);
#pragma unroll 8
          for (; i < end; ++i) fsum += chunk[i];
          if ((base + i) % g == 0) {
            dsum += (double)fsum;
            fsum = 0.f;
          }
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0 && n % g != 0) dsum += (double)fsum;
  }
  if (threadIdx.x == 0) st->sum = dsum;
}

// The same append in two launches for large blocks (a batched actor shard inserts thousands of rows per tick): the
// ATen-exact pow is the expensive part and is embarrassingly parallel, only the sums are ordered.
// (1) any number of workgroups: w = pow(prio) into the ring and into a contiguous scratch copy;
// (2) one workgroup: per reference block the FLOAT sum in row order (one thread per block), then thread 0 folds the
//     block sums into the f64 sum_ in order -- exactly the arithmetic of replay_append_weights.
__global__ void replay_append_pow(const float* __restrict__ prio, int n, float alpha, float* __restrict__ w, int ring,
                                  int start, int group, float* __restrict__ tmp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = group > 0 ? group : n;
  const int gi = i / g;
  const int glen = min(g, n - gi * g);
  const float v = pow_aten(prio[i], alpha, i - gi * g, glen);
  tmp[i] = v;
  w[(int)(((int64_t)start + i) % ring)] = v;
}
__global__ __launch_bounds__(kThreads) void replay_append_sums(const float* __restrict__ tmp, int n, int group,
                                                               ReplayDevState* __restrict__ st) {
  __shared__ float gsum[kThreads];
  const int g = group > 0 ? group : n;
  const int ng = (n + g - 1) / g;
  double dsum = 0.0;
  if (threadIdx.x == 0) dsum = st->sum;
  for (int base = 0; base < ng; base += kThreads) {
    const int j = base + threadIdx.x;
    if (j < ng) {
      const int lo = j * g, hi = min(lo + g, n);
      float f = 0.f;
      for (int i = lo; i < hi; ++i) f += tmp[i];
      gsum[threadIdx.x] = f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int m = min(kThreads, ng - base);
      for (int q = 0; q < m; ++q) dsum += (double)gsum[q];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) st->sum = dsum;
}

// copies n rows of one field into ring slots start.. (mod ring); 16-byte lanes when possible
__global__ __launch_bounds__(kThreads) void replay_scatter_rows(const uint8_t* __restrict__ src,
                                                                uint8_t* __restrict__ dst, int64_t row_bytes,
                                                                int n, int ring, int start, int vec16) {
  for (int row = blockIdx.y; row < n; row += gridDim.y) {
    const int slot = (int)(((int64_t)start + row) % ring);
    const uint8_t* s = src + (int64_t)row * row_bytes;
    uint8_t* d = dst + (int64_t)slot * row_bytes;
    if (vec16) {
      const int64_t nv = row_bytes >> 4;
      const uint4* s4 = reinterpret_cast<const uint4*>(s);
      uint4* d4 = reinterpret_cast<uint4*>(d);
      for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kThreads)
        d4[i] = s4[i];
    } else {
      for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < row_bytes; i += (int64_t)gridDim.x * kThreads)
        d[i] = s[i];
    }
  }
}

// Fields whose rows are a few bytes (eps, legal_move, a, reward, terminal, bootstrap: 4..72 B) are moved
// by ONE launch for all of them: a thread owns one row and walks the small fields (1 MB in total per
// 6,400-row block; what mattered was eight launches per insert / per sample, not the bytes).
constexpr int kSmallRowBytes = 256;
constexpr int kMaxSmallFields = 12;
struct SmallFields {
  const uint8_t* src[kMaxSmallFields];
  uint8_t* dst[kMaxSmallFields];
  int32_t row_bytes[kMaxSmallFields];
  int32_t n;
};
__device__ __forceinline__ void copy_small(uint8_t* __restrict__ d, const uint8_t* __restrict__ s, int nbytes) {
  if (((nbytes | (int)(uintptr_t)d | (int)(uintptr_t)s) & 3) == 0) {
    for (int i = 0; i < nbytes; i += 4) *reinterpret_cast<uint32_t*>(d + i) = *reinterpret_cast<const uint32_t*>(s + i);
  } else {
    for (int i = 0; i < nbytes; ++i) d[i] = s[i];
  }
}
__global__ void replay_scatter_small(SmallFields t, int n, int ring, int start) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const int64_t slot = ((int64_t)start + row) % ring;
  for (int f = 0; f < t.n; ++f)
    copy_small(t.dst[f] + slot * t.row_bytes[f], t.src[f] + (int64_t)row * t.row_bytes[f], t.row_bytes[f]);
}
// indexed form: row r comes from source row src_idx[r] and goes to slot start + dst_off[r]
__global__ __launch_bounds__(kThreads) void replay_scatter_rows_indexed(const uint8_t* __restrict__ src,
                                                                        const int32_t* __restrict__ src_idx,
                                                                        const int32_t* __restrict__ dst_off,
                                                                        uint8_t* __restrict__ dst, int64_t row_bytes,
                                                                        int n, int ring, int start, int vec16) {
  for (int row = blockIdx.y; row < n; row += gridDim.y) {
    const int slot = (int)(((int64_t)start + dst_off[row]) % ring);
    const uint8_t* s = src + (int64_t)src_idx[row] * row_bytes;
    uint8_t* d = dst + (int64_t)slot * row_bytes;
    if (vec16) {
      const int64_t nv = row_bytes >> 4;
      const uint4* s4 = reinterpret_cast<const uint4*>(s);
      uint4* d4 = reinterpret_cast<uint4*>(d);
      for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kThreads)
        d4[i] = s4[i];
    } else {
      for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < row_bytes; i += (int64_t)gridDim.x * kThreads)
        d[i] = s[i];
    }
  }
}

// what a failed scan leaves in host memory (see replay_search)
struct ScanFailure {
  int32_t code;
  int32_t size;
  float sum_f;
  double target;
};

// ---- sample ---------------------------------------------------------------------------
// The stratified targets (:261-280) are computed by the chain kernel of the scan index (seqsum.hip).
// Blocks [0, batch): one wavefront per stratum (seq_find_wave: three coalesced loads over the exact prefix
// arrays + at most 64 native adds).  Blocks >= batch: blockPop :84-103 -- the evicted range is the first n_pop
// logical slots of the range the scan just indexed, and `diff -= w` from zero is the negated sequential
// prefix (RNE is symmetric); the first of them evaluates that prefix exactly.
__global__ __launch_bounds__(64) void replay_search(SeqView v, const double* __restrict__ eff, int batch,
                                                    int32_t* __restrict__ ids, float* __restrict__ raw_w,
                                                    uint8_t* __restrict__ evicted, ReplayDevState* __restrict__ st,
                                                    int n_pop, ScanFailure* __restrict__ fail_host) {
  const int i = blockIdx.x;
  if (i >= batch) {
    const int pb = i - batch, npb = gridDim.x - batch;
    for (int k = pb * 64 + (threadIdx.x & 63); k < n_pop; k += npb * 64) evicted[seq_phys(v, k)] = 1;
    if (pb == 0) {
      const double diff = -seq_prefix_wave(v, n_pop);
      if ((threadIdx.x & 63) == 0) {
        st->last_pop = diff;
        st->sum += diff;
      }
    }
    return;
  }
  const SeqHit h = seq_find_wave(v, eff[i]);
  if ((threadIdx.x & 63) != 0) return;
  int64_t k = h.k;
  if (!h.found) {  // :297-302 (the reference aborts here)
    st->err = RELA_ESCAN;
    k = v.size > 0 ? v.size - 1 : 0;
    // tell the host without a synchronisation: a page-locked record the next replay call reads (what the
    // reference prints before its assert(false): the target and where the scan ended)
    fail_host->target = eff[i];
    fail_host->size = (int32_t)v.size;
    fail_host->sum_f = st->sum_f;
    __hip_atomic_store(&fail_host->code, (int32_t)RELA_ESCAN, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int64_t p = seq_phys(v, k);
  ids[i] = (int32_t)p;
  raw_w[i] = h.found ? h.w : 0.f;
  // getElementAndMark :124-128 clears the flag; a slot that is popped right after (:311-315) ends up evicted,
  // and the pop blocks of this launch set that flag
  if (k >= n_pop) evicted[p] = 0;
}

// Block 0: IS weights :320-322 (w/sum -> pow(size*w, -beta) -> /= max).  Block f + 1: gather of small field f
// (eps, legal_move, a, reward, terminal, bootstrap ...) of the batch, out[b] = field[ids[b]] -- four
// independent (row, word) items in flight per thread.  One launch for all of it.
__global__ __launch_bounds__(1024) void replay_finish(const float* __restrict__ raw_w, int batch, float size_f,
                                                      float beta, const ReplayDevState* __restrict__ st,
                                                      float* __restrict__ out, SmallFields t,
                                                      const int32_t* __restrict__ ids) {
  if (blockIdx.x > 0) {
    const int f = blockIdx.x - 1;
    const int rb = t.row_bytes[f];
    if ((rb & 3) == 0 && (((uintptr_t)t.dst[f] | (uintptr_t)t.src[f]) & 3) == 0) {
      const int words = rb >> 2, total = batch * words;
      const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(t.src[f]);
      uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(t.dst[f]);
      for (int base = threadIdx.x; base < total; base += 4 * blockDim.x) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * blockDim.x;
          if (idx < total) {
            const int row = idx / words;
            v[u] = src[(int64_t)ids[row] * words + (idx - row * words)];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * blockDim.x;
          if (idx < total) dst[idx] = v[u];
        }
      }
    } else {
      const uint8_t* __restrict__ src = t.src[f];
      uint8_t* __restrict__ dst = t.dst[f];
      for (int idx = threadIdx.x; idx < batch * rb; idx += blockDim.x) {
        const int row = idx / rb;
        dst[idx] = src[(int64_t)ids[row] * rb + (idx - row * rb)];
      }
    }
    return;
  }
  __shared__ float red[1024];
  const float sum = st->sum_f;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < batch; i += blockDim.x) {
    const float q = raw_w[i] / sum;
    const float s = size_f * q;
    const float p = pow_aten(s, -beta, i, batch);  // torch::pow(size * weights, -beta_) :321
    out[i] = p;
    mx = fmaxf(mx, p);
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + off]);
    __syncthreads();
  }
  mx = red[0];
  for (int i = threadIdx.x; i < batch; i += blockDim.x) out[i] = out[i] / mx;
}

// gathers batch rows of every LARGE field in one launch (blockIdx.z = field): out[i] = field[ids[i]]
// (makeBatch, types.cc:8-46); `steps` > 1: the row is a sequence of `steps` sub-rows and the output is
// time-major, out[t][b] = slot_b[t]  (RNNTransition::makeBatch, types.cc:140-182)
// De-duplicated stack field: out[b] = the `ups` units refs[ids[b]][0..ups) point to (the stack
// GameState::computeFeature built on the way in, atari/game_state.h:53-82, is rebuilt on the way out).
__global__ __launch_bounds__(kThreads) void replay_gather_dedup(const int32_t* __restrict__ refs,
                                                               const int32_t* __restrict__ ids,
                                                               const uint8_t* __restrict__ units, int64_t unit_bytes,
                                                               int ups, uint8_t* __restrict__ out, int batch) {
  for (int y = blockIdx.y; y < batch * ups; y += gridDim.y) {
    const int b = y / ups, k = y - b * ups;
    const int64_t u = refs[(int64_t)ids[b] * ups + k];
    const uint4* s4 = reinterpret_cast<const uint4*>(units + u * unit_bytes);
    uint4* d4 = reinterpret_cast<uint4*>(out + ((int64_t)b * ups + k) * unit_bytes);
    const int64_t nv = unit_bytes >> 4;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kThreads) d4[i] = s4[i];
  }
}

// copies `count` units from a pitched source into the unit ring at sequence first_seq..
__global__ __launch_bounds__(kThreads) void replay_units_write(const uint8_t* __restrict__ src, int64_t pitch,
                                                              uint8_t* __restrict__ units, int64_t unit_bytes,
                                                              int64_t cap, int64_t first_seq, int count) {
  for (int row = blockIdx.y; row < count; row += gridDim.y) {
    const uint4* s4 = reinterpret_cast<const uint4*>(src + (int64_t)row * pitch);
    uint4* d4 = reinterpret_cast<uint4*>(units + ((first_seq + row) % cap) * unit_bytes);
    const int64_t nv = unit_bytes >> 4;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kThreads) d4[i] = s4[i];
  }
}

constexpr int kMaxBigFields = 12;
struct BigFields {
  const uint8_t* field[kMaxBigFields];
  uint8_t* out[kMaxBigFields];
  int64_t slot_bytes[kMaxBigFields];
  int32_t steps[kMaxBigFields];
  int32_t vec16[kMaxBigFields];
  // the batch may be a SLICE of a larger output: rows [out_off, out_off + batch) of tensors with out_batch rows per
  // (time) step -- one partition's share of a batch drawn over several (0 = the output holds exactly this batch)
  int32_t out_batch, out_off;
};
__global__ __launch_bounds__(kThreads) void replay_gather_big(BigFields t, const int32_t* __restrict__ ids, int batch) {
  const int f = blockIdx.z;
  const int steps = t.steps[f];
  const int64_t row_bytes = t.slot_bytes[f] / steps;
  // grid.y is capped (65,535 limit; batch * steps = 62,976 already at B = 512, T = 123): stride over (t, b)
  for (int y = blockIdx.y; y < batch * steps; y += gridDim.y) {
    const int b = y % batch, ts = y / batch;
    const uint8_t* s = t.field[f] + (int64_t)ids[b] * t.slot_bytes[f] + (int64_t)ts * row_bytes;
    const int ob = t.out_batch > 0 ? t.out_batch : batch;
    uint8_t* d = t.out[f] + ((int64_t)ts * ob + t.out_off + b) * row_bytes;
    if (t.vec16[f]) {
      const int64_t nv = row_bytes >> 4;
      const uint4* s4 = reinterpret_cast<const uint4*>(s);
      uint4* d4 = reinterpret_cast<uint4*>(d);
      for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kThreads)
        d4[i] = s4[i];
    } else {
      for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < row_bytes; i += (int64_t)gridDim.x * kThreads)
        d[i] = s[i];
    }
  }
}

// ---- update ---------------------------------------------------------------------------
// update :105-119 for the outstanding batch.  Sequential semantics with duplicate ids: the
// i-th occurrence sees the weight written by the previous occurrence.  One workgroup:
// pow + old-weight gather + duplicate resolution in parallel, the f64 diff chain on lane 0.
__global__ __launch_bounds__(1024) void replay_update(const float* __restrict__ prio, int n, float alpha,
                                                      const int32_t* __restrict__ ids,
                                                      const uint8_t* __restrict__ evicted,
                                                      float* __restrict__ w, ReplayDevState* __restrict__ st) {
  __shared__ float neww[kMaxBatch];
  __shared__ float dlt[kMaxBatch];
  __shared__ __attribute__((aligned(16))) int32_t sid[kMaxBatch];
  __shared__ uint8_t is_last[kMaxBatch];  // separate from sid[]: other lanes are still scanning sid[]
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    neww[i] = pow_aten(prio[i], alpha, i, n);  // torch::pow(priority, alpha_) over the batch :239
    sid[i] = ids[i];
  }
  const int n4 = (n + 3) & ~3;
  for (int i = n + threadIdx.x; i < n4; i += blockDim.x) sid[i] = -1;  // padding of the last int4
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int id = sid[i];
    float d = 0.f;
    bool last = true;
    if (!evicted[id]) {
      // no early exits: every lane reads the same four ids per LDS access (a broadcast), the loop pipelines
      int prev = -1;
      const int4* s4 = reinterpret_cast<const int4*>(sid);
      for (int j4 = 0; j4 < n4 / 4; ++j4) {
        const int4 q = s4[j4];
        const int j = j4 * 4;
        const int qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool same = qq[u] == id;
          prev = (same && j + u < i) ? j + u : prev;
          last = (same && j + u > i) ? false : last;
        }
      }
      const float old = prev >= 0 ? neww[prev] : w[id];
      d = neww[i] - old;  // float - float :113
    } else {
      last = false;
    }
    dlt[i] = d;
    is_last[i] = last ? 1 : 0;
  }
  // all reads of w[] above must finish before anyone writes: writes happen after the barrier
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    if (is_last[i]) w[sid[i]] = neww[i];
  // diff += (double)dlt[i] in order (:113-116): a dependent f64 chain.  One wavefront keeps 64 values per
  // register and feeds the chain through lane broadcasts instead of one LDS round trip per element.
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    double diff = 0;
    for (int base = 0; base < n; base += 64) {
      const float mine = base + lane < n ? dlt[base + lane] : 0.f;  // evicted ids contribute +0.0f (a no-op)
      const int cnt = n - base < 64 ? n - base : 64;
      for (int e = 0; e < cnt; ++e) diff += (double)rl_f(mine, e);
    }
    if (lane == 0) st->sum += diff;
  }
}

// ---- remote partitions (rela_replay_remote_gather): the small fields, raw weights and float sum of a sample whose
// ids, field arrays and state live in ANOTHER process's memory (IPC mappings: peer reads over xGMI between GPUs)
__global__ __launch_bounds__(256) void replay_gather_small_remote(SmallFields t, const int32_t* __restrict__ ids, int batch,
                                                                   const float* __restrict__ raw_w,
                                                                   const ReplayDevState* __restrict__ st,
                                                                   float* __restrict__ raw_w_out, float* __restrict__ sum_out,
                                                                   int out_off) {
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < batch; i += blockDim.x)
      if (raw_w_out) raw_w_out[i] = raw_w[i];
    if (threadIdx.x == 0 && sum_out) sum_out[0] = st->sum_f;
    return;
  }
  const int f = blockIdx.x - 1;
  const int rb = t.row_bytes[f];
  const uint8_t* __restrict__ src = t.src[f];
  uint8_t* __restrict__ dst = t.dst[f];
  for (int idx = threadIdx.x; idx < batch * rb; idx += blockDim.x) {
    const int row = idx / rb;
    dst[(int64_t)out_off * rb + idx] = src[(int64_t)ids[row] * rb + (idx - row * rb)];
  }
}

}  // namespace
}  // namespace rela_amd

using namespace rela_amd;

struct rela_replay {
  int device = 0;
  int capacity = 0, ring = 0, prefetch = 0;
  float alpha = 0, beta = 0;
  std::mt19937 rng;
  mutable std::mutex m;
  std::condition_variable cv_size;
  int head = 0, tail = 0, size = 0;
  bool shut = false;                 // rela_replay_shutdown: producers no longer block
  int safe_tail = 0, safe_size = 0;  // committed prefix, ConcurrentQueue::safeTail_/safeSize_
  std::condition_variable cv_tail;
  std::atomic<int64_t> num_add{0};
  int n_sampled = 0;
  int last_full_size = 0;  // size_ as re-read by the last sample_ (:312), the N of its IS weights
  hipStream_t stream = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_wait = nullptr;
  // r4: the row copies of an insert (56 KB per transition -- all but a few bytes of its traffic) run on a second
  // stream instead of queueing behind the sample path's latency-bound chain on `stream`: a block's slots belong to
  // its producer alone between reserve and commit (prioritized_replay.h:58-66 copies them with the mutex released),
  // so only the COMMIT -- weights and sum_, in slot order -- has to take its turn among sample / update.  The
  // priorities travel through a small ring of staging buffers, (see above)
  // page-locked record a scan that ran off the end of the ring writes (prioritized_replay.h:297-302: the reference
  // prints it and aborts); read at the next sample / update_priority, which then fail with RELA_ESCAN
  ScanFailure* scan_failure = nullptr;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_cin = nullptr, ev_cout = nullptr;
  static constexpr int kPrioStages = 8;
  float* d_pstage[kPrioStages] = {};
  int pstage_cap[kPrioStages] = {};
  hipEvent_t ev_stage[kPrioStages] = {}, ev_done[kPrioStages] = {};
  bool pstage_used[kPrioStages] = {};
  int pstage_next = 0;
  // slots evicted by a sample whose gathers may still be reading them (a sampled row can be evicted by the very call
  // that drew it, :288-315): a row copy into such a slot waits for that sample's event
  struct Eviction {
    int start, count;
    hipEvent_t done;
  };
  std::deque<Eviction> evictions;
  // slots popped by a sample that gathered NOTHING itself (out_rows_dev = NULL: the owner's half of the native
  // exchange; the learner's rela_replay_remote_gather reads the rows later, from another process, where no event of
  // this one can order it): they stay reserved -- begin_add counts them as occupied -- until the update_priority that
  // ends the batch, which the protocol puts behind the learner's read (producers blocked on a full ring
  // used to rewrite exactly the slots such a sample had just evicted and possibly drawn)
  int held = 0;
  std::vector<hipEvent_t> ev_pool;
  // rela_replay_set_decoupled_insert: false (default) = an insert runs wholly on `stream`, in order with sample /
  // update; true = its row copies and priority staging run on `copy_stream` (see above)
  bool legacy_insert = true;
  hipStream_t copy_stream_own = nullptr;
  bool deferred_wait = false;  // rela_replay_set_deferred_wait: sample / update_priority do not stall the caller's stream
  float* d_w = nullptr;
  uint8_t* d_evicted = nullptr;
  ReplayDevState* d_state = nullptr;
  float* d_tmpw = nullptr;  // contiguous copy of the weights of the block being committed (two-launch append)
  int tmpw_cap = 0;
  int32_t* d_ids = nullptr;
  float* d_raw_w = nullptr;
  float* d_targets = nullptr;
  double* d_eff = nullptr;
  uint32_t* d_draws = nullptr;
  float* d_prio = nullptr;  // staging for host-side priorities
  std::vector<int64_t> row_bytes;
  std::vector<int32_t> steps;  // sub-rows per slot (1 = plain field)
  std::vector<uint8_t*> d_fields;
  // chunk_bytes > 0: every field array is chunks of physical memory of at most that size behind one virtual range
  // (vmm_field.h), so that a partition of any size can be exported to another process; 0: one hipMalloc per field, vmm[f] null
  int64_t chunk_bytes = 0;
  std::vector<VmmRange*> vmm;
  SeqIndex ix;
  HostStage stage;  // pinned staging of the RNG draws / host-side priorities (guarded by m)
  // frame-stack de-duplication (rela_replay_set_schema_dedup): the two stack fields hold int32 references
  // into a ring of UNITS (one 84x84 plane, or one 4-plane stack) addressed by a monotone sequence number
  int dd_ups = 0;                 // units per stack (0 = de-duplication off)
  int64_t dd_unit_bytes = 0, dd_cap = 0;
  uint8_t* d_units = nullptr;     // [dd_cap][dd_unit_bytes]
  VmmRange* vmm_units = nullptr;  // chunked mode: d_units = vmm_units->base
  int dd_field[2] = {-1, -1};
  int64_t dd_next_seq = 0;        // sequence number of the next unit
  std::vector<int64_t> dd_slot_min;  // [ring] smallest unit sequence a slot refers to (host; guarded by m)
};

static std::atomic<int64_t> g_default_chunk_bytes{0};
extern "C" int rela_runtime_set_replay_chunk_bytes(int64_t bytes) {
  RELA_CHECK(bytes >= 0, RELA_EINVAL, "rela_runtime_set_replay_chunk_bytes: %lld", (long long)bytes);
  g_default_chunk_bytes.store(bytes);
  return RELA_OK;
}

extern "C" const char* rela_last_error(void) { return g_err; }
extern "C" int rela_abi_version(void) { return 1; }

extern "C" int rela_replay_create(rela_replay** out, int capacity, int seed, float alpha, float beta,
                                  int prefetch, int device) {
  RELA_CHECK(out && capacity > 0, RELA_EINVAL, "rela_replay_create: bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_last_error("rela_replay_create: HIP device %d not available (%d visible); there is no CPU path", device,
                   ndev);
    return RELA_ENODEV;
  }
  DeviceGuard g(device);
  auto* r = new rela_replay();
  r->device = device;
  r->capacity = capacity;
  r->ring = (int)(1.25 * capacity);  // :181
  r->alpha = alpha;
  r->beta = beta;
  r->prefetch = prefetch;
  r->rng.seed(seed);  // :183
  r->chunk_bytes = g_default_chunk_bytes.load();
  if (const char* e = getenv("RELA_REPLAY_CHUNK_GB")) r->chunk_bytes = (int64_t)(atof(e) * (double)((int64_t)1 << 30));
  {  // replay operations are short and sit on the critical path of actors AND learner: highest priority
    int least = 0, greatest = 0;
    RELA_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    RELA_HIP(hipStreamCreateWithPriority(&r->stream, hipStreamNonBlocking, greatest));
  }
  RELA_HIP(hipEventCreateWithFlags(&r->ev_in, hipEventDisableTiming));
  RELA_HIP(hipEventCreateWithFlags(&r->ev_out, hipEventDisableTiming));
  RELA_HIP(hipEventCreateWithFlags(&r->ev_wait, hipEventDisableTiming));
  RELA_HIP(hipHostMalloc(reinterpret_cast<void**>(&r->scan_failure), sizeof(ScanFailure), hipHostMallocDefault));
  memset(r->scan_failure, 0, sizeof(ScanFailure));
  // (normal priority: at the replay stream's priority the copies starve the learner's kernels -- measured in the
  // bench: 4.58 M env-steps/s at normal, 3.86 M at the highest priority)
  RELA_HIP(hipStreamCreateWithFlags(&r->copy_stream_own, hipStreamNonBlocking));
  r->copy_stream = r->stream;  // until rela_replay_set_decoupled_insert(r, 1)
  RELA_HIP(hipEventCreateWithFlags(&r->ev_cin, hipEventDisableTiming));
  RELA_HIP(hipEventCreateWithFlags(&r->ev_cout, hipEventDisableTiming));
  for (int k = 0; k < rela_replay::kPrioStages; ++k) {
    RELA_HIP(hipEventCreateWithFlags(&r->ev_stage[k], hipEventDisableTiming));
    RELA_HIP(hipEventCreateWithFlags(&r->ev_done[k], hipEventDisableTiming));
  }
  RELA_HIP(hipMalloc(&r->d_w, sizeof(float) * (size_t)r->ring));
  RELA_HIP(hipMalloc(&r->d_evicted, (size_t)r->ring));
  RELA_HIP(hipMalloc(&r->d_state, sizeof(ReplayDevState)));
  RELA_HIP(hipMalloc(&r->d_ids, sizeof(int32_t) * kMaxBatch));
  RELA_HIP(hipMalloc(&r->d_raw_w, sizeof(float) * kMaxBatch));
  RELA_HIP(hipMalloc(&r->d_targets, sizeof(float) * kMaxBatch));
  RELA_HIP(hipMalloc(&r->d_eff, sizeof(double) * kMaxBatch));
  RELA_HIP(hipMalloc(&r->d_draws, sizeof(uint32_t) * kMaxBatch));
  RELA_HIP(hipMalloc(&r->d_prio, sizeof(float) * kMaxBatch));
  RELA_HIP(hipMemsetAsync(r->d_w, 0, sizeof(float) * (size_t)r->ring, r->stream));
  RELA_HIP(hipMemsetAsync(r->d_evicted, 0, (size_t)r->ring, r->stream));
  RELA_HIP(hipMemsetAsync(r->d_state, 0, sizeof(ReplayDevState), r->stream));
  int rc = seq_index_alloc(&r->ix, r->ring);
  if (rc != RELA_OK) return rc;
  rc = r->stage.init(sizeof(uint32_t) * kMaxBatch);
  RELA_CHECK(rc == RELA_OK, rc, "rela_replay_create: pinned staging buffer");
  RELA_HIP(hipStreamSynchronize(r->stream));
  *out = r;
  return RELA_OK;
}

extern "C" void rela_replay_destroy(rela_replay* r) {
  if (!r) return;
  DeviceGuard g(r->device);
  (void)hipStreamSynchronize(r->copy_stream_own);
  (void)hipStreamSynchronize(r->stream);
  for (int k = 0; k < rela_replay::kPrioStages; ++k) {
    (void)hipFree(r->d_pstage[k]);
    (void)hipEventDestroy(r->ev_stage[k]);
    (void)hipEventDestroy(r->ev_done[k]);
  }
  for (auto& e : r->evictions) (void)hipEventDestroy(e.done);
  for (auto e : r->ev_pool) (void)hipEventDestroy(e);
  (void)hipHostFree(r->scan_failure);
  (void)hipEventDestroy(r->ev_cin);
  (void)hipEventDestroy(r->ev_cout);
  (void)hipStreamDestroy(r->copy_stream_own);
  for (size_t f = 0; f < r->d_fields.size(); ++f) {
    if (f < r->vmm.size() && r->vmm[f]) {
      r->vmm[f]->destroy();
      delete r->vmm[f];
    } else {
      (void)hipFree(r->d_fields[f]);
    }
  }
  seq_index_free(&r->ix);
  r->stage.destroy();
  if (r->vmm_units) {
    r->vmm_units->destroy();
    delete r->vmm_units;
  } else {
    (void)hipFree(r->d_units);
  }
  (void)hipFree(r->d_w);
  (void)hipFree(r->d_evicted);
  (void)hipFree(r->d_state);
  (void)hipFree(r->d_tmpw);
  (void)hipFree(r->d_ids);
  (void)hipFree(r->d_raw_w);
  (void)hipFree(r->d_targets);
  (void)hipFree(r->d_eff);
  (void)hipFree(r->d_draws);
  (void)hipFree(r->d_prio);
  (void)hipEventDestroy(r->ev_in);
  (void)hipEventDestroy(r->ev_out);
  (void)hipEventDestroy(r->ev_wait);
  (void)hipStreamDestroy(r->stream);
  delete r;
}

extern "C" int rela_replay_set_schema_seq(rela_replay* r, int nfields, const int64_t* row_bytes,
                                          const int32_t* steps);

extern "C" int rela_replay_set_chunk_bytes(rela_replay* r, int64_t bytes) {
  RELA_CHECK(r && bytes >= 0, RELA_EINVAL, "rela_replay_set_chunk_bytes: bad arguments");
  std::lock_guard<std::mutex> lk(r->m);
  RELA_CHECK(r->d_fields.empty(), RELA_ESTATE, "rela_replay_set_chunk_bytes: the schema is already set");
  r->chunk_bytes = bytes;
  return RELA_OK;
}

extern "C" int rela_replay_set_schema(rela_replay* r, int nfields, const int64_t* row_bytes) {
  return rela_replay_set_schema_seq(r, nfields, row_bytes, nullptr);
}

extern "C" int rela_replay_set_schema_seq(rela_replay* r, int nfields, const int64_t* row_bytes,
                                          const int32_t* steps) {
  RELA_CHECK(r && nfields >= 0 && (nfields == 0 || row_bytes), RELA_EINVAL, "rela_replay_set_schema: bad arguments");
  std::lock_guard<std::mutex> lk(r->m);
  RELA_CHECK(r->d_fields.empty() && r->num_add.load() == 0, RELA_ESTATE,
             "rela_replay_set_schema: schema already set or replay not empty");
  DeviceGuard g(r->device);
  for (int f = 0; f < nfields; ++f) {
    RELA_CHECK(row_bytes[f] > 0, RELA_EINVAL, "rela_replay_set_schema: field %d has %lld bytes", f,
               (long long)row_bytes[f]);
    const int st = steps ? steps[f] : 1;
    RELA_CHECK(st >= 1 && row_bytes[f] % st == 0, RELA_EINVAL, "rela_replay_set_schema: field %d: %lld bytes not divisible into %d steps",
               f, (long long)row_bytes[f], st);
    uint8_t* p = nullptr;
    VmmRange* v = nullptr;
    const size_t bytes = (size_t)row_bytes[f] * (size_t)r->ring;
    // chunked mode takes EVERY field, also those of one chunk: mixing the two mappings in the importer is what failed -- after
    // the 37 GB frame-stack field had been mapped from its chunks, hipIpcOpenMemHandle of the partition's 4.03 GB sequence
    // field (one hipMalloc) did not return within 90 s (profiles/r05_vmm_mixed_import_hang.log)
    if (r->chunk_bytes > 0) {
      v = new VmmRange();
      hipError_t e = v->create(bytes, (size_t)r->chunk_bytes, r->device);
      if (e != hipSuccess) {
        set_last_error("rela_replay_set_schema: field %d: %.1f GB in chunks of %.1f GB: %s", f, (double)bytes / 1e9,
                       (double)r->chunk_bytes / 1e9, hipGetErrorString(e));
        v->destroy();
        delete v;
        return e == hipErrorOutOfMemory ? RELA_ENOMEM : RELA_ENODEV;
      }
      p = v->base;
    } else {
      RELA_HIP(hipMalloc(&p, bytes));
    }
    r->d_fields.push_back(p);
    r->vmm.push_back(v);
    r->row_bytes.push_back(row_bytes[f]);
    r->steps.push_back(st);
  }
  return RELA_OK;
}

// ---- frame-stack de-duplication ------------------------------------------------------
extern "C" int rela_replay_set_schema_dedup(rela_replay* r, int nfields, const int64_t* row_bytes, int field_a,
                                            int field_b, int64_t unit_bytes, int units_per_stack,
                                            int64_t guard_units) {
  RELA_CHECK(r && nfields >= 2 && row_bytes && field_a >= 0 && field_b >= 0 && field_a < nfields && field_b < nfields &&
                 field_a != field_b && unit_bytes > 0 && unit_bytes % 16 == 0 && units_per_stack >= 1 &&
                 units_per_stack <= 16 && guard_units >= 0,
             RELA_EINVAL, "rela_replay_set_schema_dedup: bad arguments");
  RELA_CHECK(row_bytes[field_a] == unit_bytes * units_per_stack && row_bytes[field_b] == row_bytes[field_a], RELA_EINVAL,
             "rela_replay_set_schema_dedup: the stack fields must be units_per_stack * unit_bytes long");
  std::vector<int64_t> rb(row_bytes, row_bytes + nfields);
  rb[field_a] = rb[field_b] = (int64_t)sizeof(int32_t) * units_per_stack;  // references instead of frames
  int rc = rela_replay_set_schema_seq(r, nfields, rb.data(), nullptr);
  if (rc != RELA_OK) return rc;
  DeviceGuard g(r->device);
  std::lock_guard<std::mutex> lk(r->m);
  r->dd_ups = units_per_stack;
  r->dd_unit_bytes = unit_bytes;
  r->dd_cap = (int64_t)r->ring + guard_units;
  RELA_CHECK(r->dd_cap < ((int64_t)1 << 31), RELA_EINVAL, "rela_replay_set_schema_dedup: unit ring too large");
  r->dd_field[0] = field_a;
  r->dd_field[1] = field_b;
  const size_t unit_ring_bytes = (size_t)r->dd_cap * (size_t)unit_bytes;
  if (r->chunk_bytes > 0) {  // exported like the field arrays (vmm_field.h)
    r->vmm_units = new VmmRange();
    hipError_t e = r->vmm_units->create(unit_ring_bytes, (size_t)r->chunk_bytes, r->device);
    if (e != hipSuccess) {
      set_last_error("rela_replay_set_schema_dedup: unit ring of %.1f GB in chunks of %.1f GB: %s", (double)unit_ring_bytes / 1e9,
                     (double)r->chunk_bytes / 1e9, hipGetErrorString(e));
      r->vmm_units->destroy();
      delete r->vmm_units;
      r->vmm_units = nullptr;
      return e == hipErrorOutOfMemory ? RELA_ENOMEM : RELA_ENODEV;
    }
    r->d_units = r->vmm_units->base;
  } else {
    RELA_HIP(hipMalloc(&r->d_units, unit_ring_bytes));
  }
  r->dd_slot_min.assign((size_t)r->ring, 0);
  return RELA_OK;
}

extern "C" int rela_replay_units_reserve(rela_replay* r, int count, int nonblocking, int64_t* first_seq,
                                         int32_t* first_index) {
  RELA_CHECK(r && r->dd_ups > 0 && count > 0 && first_seq, RELA_EINVAL, "rela_replay_units_reserve: bad arguments");
  RELA_CHECK(count <= r->dd_cap, RELA_EINVAL, "rela_replay_units_reserve: %d units exceed the unit ring", count);
  std::unique_lock<std::mutex> lk(r->m);
  // a unit may be overwritten only when no live (or reserved) slot refers to it: FIFO order makes the slot at
  // `head` the one with the smallest reference
  auto fits = [&] {
    const int64_t oldest = r->size + r->held > 0 ? r->dd_slot_min[(size_t)((r->head - r->held + r->ring) % r->ring)] : r->dd_next_seq;
    return r->dd_next_seq + count - oldest <= r->dd_cap;
  };
  if (r->shut) return RELA_EWOULDBLOCK;
  if (!fits()) {
    if (nonblocking) return RELA_EWOULDBLOCK;
    r->cv_size.wait(lk, [&] { return r->shut || fits(); });
    if (r->shut) return RELA_EWOULDBLOCK;
  }
  *first_seq = r->dd_next_seq;
  if (first_index) *first_index = (int32_t)(r->dd_next_seq % r->dd_cap);
  r->dd_next_seq += count;
  return RELA_OK;
}

extern "C" int rela_replay_units_write(rela_replay* r, int64_t first_seq, int count, const void* src_dev,
                                       int64_t src_pitch, void* stream_) {
  RELA_CHECK(r && r->dd_ups > 0 && count > 0 && src_dev && src_pitch >= r->dd_unit_bytes && src_pitch % 16 == 0 &&
                 first_seq >= 0 && first_seq + count <= r->dd_next_seq && ((uintptr_t)src_dev & 15) == 0,
             RELA_EINVAL, "rela_replay_units_write: bad arguments");
  hipStream_t producer = (hipStream_t)stream_;
  DeviceGuard g(r->device);
  std::lock_guard<std::mutex> lk(r->m);
  RELA_HIP(hipEventRecord(r->ev_in, producer));
  RELA_HIP(hipStreamWaitEvent(r->stream, r->ev_in, 0));
  {
    const int64_t nv = r->dd_unit_bytes >> 4;
    const int gx = (int)std::min<int64_t>(std::max<int64_t>(1, (nv + kThreads - 1) / kThreads), 64);
    ProfScope prof("replay_scatter_rows", r->stream);
    hipLaunchKernelGGL(replay_units_write, dim3(gx, std::min(count, 32768)), dim3(kThreads), 0, r->stream,
                       (const uint8_t*)src_dev, src_pitch, r->d_units, r->dd_unit_bytes, r->dd_cap, first_seq, count);
  }
  RELA_LAUNCH_CHECK();
  RELA_HIP(hipEventRecord(r->ev_out, r->stream));
  RELA_HIP(hipStreamWaitEvent(producer, r->ev_out, 0));
  return RELA_OK;
}

extern "C" int rela_replay_set_block_min_unit(rela_replay* r, int first_slot, int n, int64_t min_seq) {
  RELA_CHECK(r && r->dd_ups > 0 && n > 0 && first_slot >= 0 && first_slot < r->ring && n <= r->ring, RELA_EINVAL,
             "rela_replay_set_block_min_unit: bad arguments");
  std::lock_guard<std::mutex> lk(r->m);
  // units_reserve protects the units the live slots declare (dd_slot_min); units a producer has stored AHEAD of the
  // transitions that will refer to them rest on the guard window (guard_units) bounding the skew between producers.
  // A producer that was stalled past that window would commit references to units that have been overwritten since:
  // refuse the block loudly instead (the actor's thread then stops its Context with this error).
  RELA_CHECK(min_seq >= r->dd_next_seq - r->dd_cap, RELA_ESTATE,
             "rela_replay_set_block_min_unit: block refers to unit %lld but units below %lld were already overwritten "
             "(a producer fell more than the guard window of %lld units behind: raise guard_units)",
             (long long)min_seq, (long long)(r->dd_next_seq - r->dd_cap), (long long)(r->dd_cap - r->ring));
  for (int i = 0; i < n; ++i) r->dd_slot_min[(size_t)((first_slot + i) % r->ring)] = min_seq;
  return RELA_OK;
}

extern "C" int rela_replay_dedup_info(const rela_replay* r, int* units_per_stack, int64_t* unit_bytes,
                                      int64_t* unit_capacity) {
  RELA_CHECK(r, RELA_EINVAL, "rela_replay_dedup_info: bad arguments");
  if (units_per_stack) *units_per_stack = r->dd_ups;
  if (unit_bytes) *unit_bytes = r->dd_unit_bytes;
  if (unit_capacity) *unit_capacity = r->dd_cap;
  return RELA_OK;
}

static inline int vec16_ok(const void* a, const void* b, int64_t row_bytes) {
  return ((row_bytes & 15) == 0) && (((uintptr_t)a & 15) == 0) && (((uintptr_t)b & 15) == 0);
}

// (caller holds r->m) Opens a row copy of `count` slots from `start` on the copy stream: ordered after the producer's
// queued work and after the gathers of every sample that evicted one of these slots and may still be reading it.
static int copy_begin(rela_replay* r, hipStream_t producer, int start, int count) {
  RELA_HIP(hipEventRecord(r->ev_cin, producer));
  RELA_HIP(hipStreamWaitEvent(r->copy_stream, r->ev_cin, 0));
  while (!r->evictions.empty() && hipEventQuery(r->evictions.front().done) == hipSuccess) {
    r->ev_pool.push_back(r->evictions.front().done);
    r->evictions.pop_front();
  }
  auto overlaps = [&](int a0, int an, int b0, int bn) {  // ring intervals [a0, a0 + an) and [b0, b0 + bn)
    const int d = ((b0 - a0) % r->ring + r->ring) % r->ring;  // b0 relative to a0
    return d < an || d + bn > r->ring;
  };
  for (auto& e : r->evictions)
    if (overlaps(start, count, e.start, e.count)) RELA_HIP(hipStreamWaitEvent(r->copy_stream, e.done, 0));
  return RELA_OK;
}
// ... and closes it: the producer may reuse its source rows once the copies ran
static int copy_end(rela_replay* r, hipStream_t producer) {
  RELA_HIP(hipEventRecord(r->ev_cout, r->copy_stream));
  RELA_HIP(hipStreamWaitEvent(producer, r->ev_cout, 0));
  return RELA_OK;
}

extern "C" int rela_replay_begin_add(rela_replay* r, int n, int nonblocking, int* first_slot) {
  RELA_CHECK(r && n > 0 && first_slot, RELA_EINVAL, "rela_replay_begin_add: bad arguments");
  RELA_CHECK(n <= r->ring, RELA_EINVAL, "rela_replay_begin_add: block of %d exceeds the ring (%d)", n, r->ring);
  std::unique_lock<std::mutex> lk(r->m);
  if (r->shut) return RELA_EWOULDBLOCK;
  if (r->size + r->held + n > r->ring) {  // cvSize_.wait :47 (+ the slots a remote reader may still be reading)
    if (nonblocking) return RELA_EWOULDBLOCK;
    r->cv_size.wait(lk, [&] { return r->shut || r->size + r->held + n <= r->ring; });
    if (r->shut) return RELA_EWOULDBLOCK;
  }
  *first_slot = r->tail;
  if (r->dd_ups > 0) {  // until the producer declares it (set_block_min_unit): nothing older than the guard window
    const int64_t lo = r->dd_next_seq - (r->dd_cap - r->ring);
    for (int i = 0; i < n; ++i) r->dd_slot_min[(size_t)((r->tail + i) % r->ring)] = lo;
  }
  r->tail = (r->tail + n) % r->ring;
  r->size += n;
  return RELA_OK;
}

extern "C" int rela_replay_write_rows(rela_replay* r, int first_slot, int offset, int count,
                                      const void* const* rows_dev, void* stream_) {
  RELA_CHECK(r && count > 0 && offset >= 0 && first_slot >= 0 && first_slot < r->ring, RELA_EINVAL,
             "rela_replay_write_rows: bad arguments");
  RELA_CHECK(r->d_fields.empty() || rows_dev, RELA_EINVAL, "rela_replay_write_rows: rows_dev is NULL");
  hipStream_t producer = (hipStream_t)stream_;
  DeviceGuard g(r->device);
  std::lock_guard<std::mutex> lk(r->m);
  const int start = (int)(((int64_t)first_slot + offset) % r->ring);
  // order after the producer's queued work, run on the COPY stream (the block's slots are the producer's alone until
  // commit), then let the producer continue only after its rows were consumed
  {
    const int rc = copy_begin(r, producer, start, count);
    if (rc != RELA_OK) return rc;
  }
  SmallFields small{};
  for (size_t f = 0; f < r->d_fields.size(); ++f) {
    if (!rows_dev[f]) continue;
    const int64_t rb = r->row_bytes[f];
    if (rb <= kSmallRowBytes && small.n < kMaxSmallFields) {
      small.src[small.n] = (const uint8_t*)rows_dev[f];
      small.dst[small.n] = r->d_fields[f];
      small.row_bytes[small.n] = (int32_t)rb;
      small.n += 1;
      continue;
    }
    const int v16 = vec16_ok(rows_dev[f], r->d_fields[f], rb);
    const int64_t units = v16 ? (rb >> 4) : rb;
    int gx = (int)std::min<int64_t>(std::max<int64_t>(1, (units + kThreads - 1) / kThreads), 64);
    {
      ProfScope prof("replay_scatter_rows", r->copy_stream);
      hipLaunchKernelGGL(replay_scatter_rows, dim3(gx, std::min(count, 32768)), dim3(kThreads), 0, r->copy_stream,
                         (const uint8_t*)rows_dev[f], r->d_fields[f], rb, count, r->ring, start, v16);
    }
  }
  if (small.n > 0) {
    ProfScope prof("replay_scatter_small", r->copy_stream);
    hipLaunchKernelGGL(replay_scatter_small, dim3(ceil_div(count, 256)), dim3(256), 0, r->copy_stream, small, count, r->ring,
                       start);
  }
  RELA_LAUNCH_CHECK();
  return copy_end(r, producer);
}

extern "C" int rela_replay_write_rows_gather(rela_replay* r, int first_slot, int count,
                                             const int32_t* dst_offset_dev, const void* const* bases_dev,
                                             const int32_t* const* src_index_dev, void* stream_) {
  RELA_CHECK(r && count > 0 && first_slot >= 0 && first_slot < r->ring && dst_offset_dev && bases_dev && src_index_dev,
             RELA_EINVAL, "rela_replay_write_rows_gather: bad arguments");
  hipStream_t producer = (hipStream_t)stream_;
  DeviceGuard g(r->device);
  std::lock_guard<std::mutex> lk(r->m);
  {
    const int rc = copy_begin(r, producer, first_slot, count);
    if (rc != RELA_OK) return rc;
  }
  for (size_t f = 0; f < r->d_fields.size(); ++f) {
    if (!bases_dev[f]) continue;
    RELA_CHECK(src_index_dev[f], RELA_EINVAL, "rela_replay_write_rows_gather: field %d has no source index", (int)f);
    const int64_t rb = r->row_bytes[f];
    const int v16 = vec16_ok(bases_dev[f], r->d_fields[f], rb);
    const int64_t units = v16 ? (rb >> 4) : rb;
    int gx = (int)std::min<int64_t>(std::max<int64_t>(1, (units + kThreads - 1) / kThreads), 64);
    {
      ProfScope prof("replay_scatter_rows", r->copy_stream);
      hipLaunchKernelGGL(replay_scatter_rows_indexed, dim3(gx, std::min(count, 32768)), dim3(kThreads), 0, r->copy_stream,
                         (const uint8_t*)bases_dev[f], src_index_dev[f], dst_offset_dev, r->d_fields[f], rb, count,
                         r->ring, first_slot, v16);
    }
  }
  RELA_LAUNCH_CHECK();
  return copy_end(r, producer);
}

extern "C" int rela_replay_commit_add(rela_replay* r, int first_slot, int n, const float* priority_dev,
                                      void* stream_) {
  return rela_replay_commit_add_grouped(r, first_slot, n, 0, priority_dev, stream_);
}

extern "C" int rela_replay_commit_add_grouped(rela_replay* r, int first_slot, int n, int group_rows,
                                              const float* priority_dev, void* stream_) {
  RELA_CHECK(r && n > 0 && priority_dev && group_rows >= 0, RELA_EINVAL, "rela_replay_commit_add: bad arguments");
  hipStream_t producer = (hipStream_t)stream_;
  DeviceGuard g(r->device);
  std::unique_lock<std::mutex> lk(r->m);
  r->cv_tail.wait(lk, [&] { return r->shut || r->safe_tail == first_slot; });  // in-order commit :69
  if (r->safe_tail != first_slot) return RELA_EWOULDBLOCK;  // shut down while an earlier block never committed
  // The priorities go through a staging buffer filled on the copy stream -- behind the block's row copies, so its
  // event also says "the rows are written" -- and the weights / sum_ update takes its turn on the replay stream;
  // the producer waits for the staging copy only, never for the replay stream.
  if (r->legacy_insert) {  // in-order form: the append reads the producer's buffer, the producer waits for it
    RELA_HIP(hipEventRecord(r->ev_in, producer));
    RELA_HIP(hipStreamWaitEvent(r->stream, r->ev_in, 0));
    if (group_rows > 0 && n >= 1024) {
      if (r->tmpw_cap < n) {
        RELA_HIP(hipStreamSynchronize(r->stream));
        (void)hipFree(r->d_tmpw);
        r->d_tmpw = nullptr;
        r->tmpw_cap = 0;
        RELA_HIP(hipMalloc(&r->d_tmpw, sizeof(float) * (size_t)n));
        r->tmpw_cap = n;
      }
      ProfScope prof("replay_append_weights", r->stream);
      hipLaunchKernelGGL(replay_append_pow, dim3((n + 255) / 256), dim3(256), 0, r->stream, priority_dev, n, r->alpha,
                         r->d_w, r->ring, first_slot, group_rows, r->d_tmpw);
      hipLaunchKernelGGL(replay_append_sums, dim3(1), dim3(kThreads), 0, r->stream, (const float*)r->d_tmpw, n,
                         group_rows, r->d_state);
    } else {
      ProfScope prof("replay_append_weights", r->stream);
      hipLaunchKernelGGL(replay_append_weights, dim3(1), dim3(kThreads), 0, r->stream, priority_dev, n, r->alpha,
                         r->d_w, r->ring, first_slot, group_rows, r->d_state);
    }
    RELA_LAUNCH_CHECK();
    RELA_HIP(hipEventRecord(r->ev_out, r->stream));
    RELA_HIP(hipStreamWaitEvent(producer, r->ev_out, 0));
    r->safe_tail = (first_slot + n) % r->ring;
    r->safe_size += n;
    r->num_add += n;
    lk.unlock();
    r->cv_tail.notify_all();
    return RELA_OK;
  }
  const int k = r->pstage_next;
  r->pstage_next = (k + 1) % rela_replay::kPrioStages;
  if (r->pstage_used[k]) RELA_HIP(hipStreamWaitEvent(r->copy_stream, r->ev_done[k], 0));  // its last reader finished
  if (r->pstage_cap[k] < n) {
    RELA_HIP(hipStreamSynchronize(r->copy_stream));
    RELA_HIP(hipStreamSynchronize(r->stream));
    (void)hipFree(r->d_pstage[k]);
    r->d_pstage[k] = nullptr;
    r->pstage_cap[k] = 0;
    RELA_HIP(hipMalloc(&r->d_pstage[k], sizeof(float) * (size_t)n));
    r->pstage_cap[k] = n;
  }
  RELA_HIP(hipEventRecord(r->ev_cin, producer));
  RELA_HIP(hipStreamWaitEvent(r->copy_stream, r->ev_cin, 0));
  // (a copy KERNEL: the producer and the replay stream both wait for this copy, and what follows one of the runtime's blit
  // kernels starts ~25 us late -- profiles/r05_trace_gaps.txt, "after replay_scatter_small before replay_append_pow")
  RELA_HIP(dev_copy2(r->d_pstage[k], priority_dev, sizeof(float) * (size_t)n, nullptr, nullptr, 0, r->copy_stream));
  RELA_HIP(hipEventRecord(r->ev_stage[k], r->copy_stream));
  RELA_HIP(hipStreamWaitEvent(producer, r->ev_stage[k], 0));
  RELA_HIP(hipStreamWaitEvent(r->stream, r->ev_stage[k], 0));
  const float* staged = r->d_pstage[k];
  // large grouped blocks (a batched shard's tick): parallel pow + ordered sums; the single-workgroup kernel otherwise
  // (an ungrouped block is ONE float sum over all its rows: nothing to parallelise but the pow, and small blocks
  // gain nothing from a second launch)
  if (group_rows > 0 && n >= 1024) {
    if (r->tmpw_cap < n) {
      RELA_HIP(hipStreamSynchronize(r->stream));
      (void)hipFree(r->d_tmpw);
      r->d_tmpw = nullptr;
      r->tmpw_cap = 0;
      RELA_HIP(hipMalloc(&r->d_tmpw, sizeof(float) * (size_t)n));
      r->tmpw_cap = n;
    }
    ProfScope prof("replay_append_weights", r->stream);
    hipLaunchKernelGGL(replay_append_pow, dim3((n + 255) / 256), dim3(256), 0, r->stream, staged, n, r->alpha,
                       r->d_w, r->ring, first_slot, group_rows, r->d_tmpw);
    hipLaunchKernelGGL(replay_append_sums, dim3(1), dim3(kThreads), 0, r->stream, (const float*)r->d_tmpw, n,
                       group_rows, r->d_state);
  } else {
    ProfScope prof("replay_append_weights", r->stream);
    hipLaunchKernelGGL(replay_append_weights, dim3(1), dim3(kThreads), 0, r->stream, staged, n, r->alpha,
                       r->d_w, r->ring, first_slot, group_rows, r->d_state);
  }
  RELA_LAUNCH_CHECK();
  RELA_HIP(hipEventRecord(r->ev_done[k], r->stream));
  r->pstage_used[k] = true;
  r->safe_tail = (first_slot + n) % r->ring;  // :70-73
  r->safe_size += n;
  r->num_add += n;  // :190
  lk.unlock();
  r->cv_tail.notify_all();
  return RELA_OK;
}

// Releases a reservation whose producer failed between begin_add and commit (no reference counterpart:
// there an exception on an actor thread ends the process).  The block is committed in order with ZERO
// weights: safe_tail advances so later blocks can commit, sum_ is unchanged, and a zero-weight slot can
// never be the first index whose running sum reaches a positive target (:286), i.e. it is never sampled.
extern "C" int rela_replay_abort_add(rela_replay* r, int first_slot, int n) {
  RELA_CHECK(r && n > 0 && first_slot >= 0 && first_slot < r->ring, RELA_EINVAL, "rela_replay_abort_add: bad arguments");
  DeviceGuard g(r->device);
  std::unique_lock<std::mutex> lk(r->m);
  r->cv_tail.wait(lk, [&] { return r->shut || r->safe_tail == first_slot; });
  if (r->safe_tail != first_slot) return RELA_EWOULDBLOCK;
  const int first_part = std::min(n, r->ring - first_slot);
  RELA_HIP(hipMemsetAsync(r->d_w + first_slot, 0, sizeof(float) * (size_t)first_part, r->stream));
  if (n > first_part) RELA_HIP(hipMemsetAsync(r->d_w, 0, sizeof(float) * (size_t)(n - first_part), r->stream));
  r->safe_tail = (first_slot + n) % r->ring;
  r->safe_size += n;
  lk.unlock();
  r->cv_tail.notify_all();
  return RELA_OK;
}

extern "C" int rela_replay_add(rela_replay* r, int n, const void* const* rows_dev, const float* priority_dev,
                               int nonblocking, void* stream_) {
  RELA_CHECK(r && n > 0 && priority_dev, RELA_EINVAL, "rela_replay_add: bad arguments");
  RELA_CHECK(r->d_fields.empty() || rows_dev, RELA_EINVAL, "rela_replay_add: rows_dev is NULL");
  int slot = 0;
  int rc = rela_replay_begin_add(r, n, nonblocking, &slot);
  if (rc != RELA_OK) return rc;
  if (!r->d_fields.empty()) {
    rc = rela_replay_write_rows(r, slot, 0, n, rows_dev, stream_);
    if (rc != RELA_OK) {
      (void)rela_replay_abort_add(r, slot, n);
      return rc;
    }
  }
  return rela_replay_commit_add(r, slot, n, priority_dev, stream_);
}

// (caller holds r->m) A previous sample's scan ran off the end of the ring: the reference prints the state and aborts
// (prioritized_replay.h:297-302).  It happens when sum_ -- block sums accumulated in FLOAT on append, exact
// differences on pop / update (:58-66,85-95) -- has drifted more than the 0.2 guard of :280 above the sum of the stored
// weights, e.g. with identical priorities, whose block sums all round the same way.  Sticky.
static int scan_failed(rela_replay* r, const char* where) {
  if (!r->scan_failure || __atomic_load_n(&r->scan_failure->code, __ATOMIC_ACQUIRE) == 0) return RELA_OK;
  set_last_error("%s: a stratified target of an earlier sample ran off the end of the ring -- nextIdx: %d/%d, sum: %.10g, "
                 "rand: %.10g (the reference aborts here, rela/prioritized_replay.h:297-302): sum_ has drifted above the "
                 "sum of the stored weights", where, r->scan_failure->size, r->scan_failure->size,
                 (double)r->scan_failure->sum_f, r->scan_failure->target);
  return RELA_ESCAN;
}

extern "C" int rela_replay_sample(rela_replay* r, int batch, void* const* out_rows_dev, float* out_weight_dev,
                                  void* stream_) {
  RELA_CHECK(r && batch > 0 && batch <= kMaxBatch && out_weight_dev, RELA_EINVAL,
             "rela_replay_sample: bad arguments (batch must be 1..%d)", kMaxBatch);
  hipStream_t consumer = (hipStream_t)stream_;
  DeviceGuard g(r->device);
  std::unique_lock<std::mutex> lk(r->m);
  RELA_CHECK(r->n_sampled == 0, RELA_ESTATE,
             "Error: previous samples' priority has not been updated.");  // :203-206
  if (int rc = scan_failed(r, "rela_replay_sample")) return rc;
  RELA_CHECK(r->safe_size > 0, RELA_ESTATE, "rela_replay_sample: replay is empty");
  // raw 32-bit draws, one per sample (generate_canonical<float,24> consumes exactly one)
  std::vector<uint32_t> draws(batch);
  for (int i = 0; i < batch; ++i) draws[i] = (uint32_t)r->rng();
  // outputs belong to the consumer: do not overwrite them before its queued work is done
  RELA_HIP(hipEventRecord(r->ev_in, consumer));
  RELA_HIP(hipStreamWaitEvent(r->stream, r->ev_in, 0));
  r->stage.begin();
  const hipError_t up = r->stage.h2d(r->d_draws, draws.data(), sizeof(uint32_t) * batch, r->stream);
  r->stage.end(r->stream);
  RELA_HIP(up);
  const int size = r->safe_size;  // storage_ [0, safeSize) is static during the scan :261-263
  SeqView v;
  SeqTargetsJob tj;
  tj.draws = r->d_draws, tj.batch = batch, tj.state = r->d_state, tj.targets = r->d_targets, tj.eff = r->d_eff;
  int rc = seq_index_build(r->ix, r->d_w, r->ring, r->head, size, r->stream, &v, &tj);
  if (rc != RELA_OK) return rc;
  // pop storage if full :311-315: `size` is re-read as size_ (reserved blocks included), and the
  // IS weights below use that value (:312,321).  Only committed slots can be evicted.
  const int full_size = r->size;
  r->last_full_size = full_size;
  const int n_pop = full_size > r->capacity ? std::min(full_size - r->capacity, size) : 0;
  {
    const int pop_blocks = n_pop > 0 ? std::min(ceil_div(n_pop, 256), 128) : 0;
    ProfScope prof("replay_search", r->stream);
    hipLaunchKernelGGL(replay_search, dim3(batch + pop_blocks), dim3(64), 0, r->stream, v, r->d_eff, batch,
                       r->d_ids, r->d_raw_w, r->d_evicted, r->d_state, n_pop, r->scan_failure);
  }
  const int evict_start = r->head;
  if (n_pop > 0) {
    r->head = (r->head + n_pop) % r->ring;
    r->size -= n_pop;
    r->safe_size -= n_pop;
  }
  SmallFields small{};
  BigFields big{};
  int nbig = 0, max_y = 1;
  int64_t max_units = 1;
  if (out_rows_dev) {
    for (size_t f = 0; f < r->d_fields.size(); ++f) {
      if (!out_rows_dev[f]) continue;
      const int64_t rb = r->row_bytes[f];
      const int st = r->steps[f];
      if (r->dd_ups > 0 && ((int)f == r->dd_field[0] || (int)f == r->dd_field[1])) continue;  // rebuilt below
      if (st == 1 && rb <= kSmallRowBytes && small.n < kMaxSmallFields) {
        small.src[small.n] = r->d_fields[f];
        small.dst[small.n] = (uint8_t*)out_rows_dev[f];
        small.row_bytes[small.n] = (int32_t)rb;
        small.n += 1;
        continue;
      }
      RELA_CHECK(nbig < kMaxBigFields, RELA_EINVAL, "rela_replay_sample: more than %d large fields", kMaxBigFields);
      const int64_t sub = rb / st;
      const int v16 = vec16_ok(out_rows_dev[f], r->d_fields[f], sub) && (rb % 16 == 0);
      big.field[nbig] = r->d_fields[f];
      big.out[nbig] = (uint8_t*)out_rows_dev[f];
      big.slot_bytes[nbig] = rb;
      big.steps[nbig] = st;
      big.vec16[nbig] = v16;
      max_units = std::max<int64_t>(max_units, v16 ? (sub >> 4) : sub);
      max_y = std::max(max_y, batch * st);
      nbig += 1;
    }
  }
  {
    ProfScope prof("replay_finish", r->stream);
    hipLaunchKernelGGL(replay_finish, dim3(1 + small.n), dim3(1024), 0, r->stream, (const float*)r->d_raw_w, batch,
                       (float)full_size, r->beta, (const ReplayDevState*)r->d_state, out_weight_dev, small,
                       (const int32_t*)r->d_ids);
  }
  if (out_rows_dev && r->dd_ups > 0) {
    for (int q = 0; q < 2; ++q) {
      const int f = r->dd_field[q];
      if (!out_rows_dev[f]) continue;
      RELA_CHECK(((uintptr_t)out_rows_dev[f] & 15) == 0, RELA_EINVAL, "rela_replay_sample: unaligned stack output");
      const int64_t nv = r->dd_unit_bytes >> 4;
      const int gx = (int)std::min<int64_t>(std::max<int64_t>(1, (nv + kThreads - 1) / kThreads), 64);
      ProfScope prof("replay_gather_rows", r->stream);
      hipLaunchKernelGGL(replay_gather_dedup, dim3(gx, std::min(batch * r->dd_ups, 32768)), dim3(kThreads), 0, r->stream,
                         (const int32_t*)r->d_fields[f], (const int32_t*)r->d_ids, (const uint8_t*)r->d_units,
                         r->dd_unit_bytes, r->dd_ups, (uint8_t*)out_rows_dev[f], batch);
    }
  }
  if (nbig > 0) {
    const int gx = (int)std::min<int64_t>(std::max<int64_t>(1, (max_units + kThreads - 1) / kThreads), 64);
    ProfScope prof("replay_gather_rows", r->stream);
    hipLaunchKernelGGL(replay_gather_big, dim3(gx, std::min(max_y, 32768), nbig), dim3(kThreads), 0, r->stream, big,
                       (const int32_t*)r->d_ids, batch);
  }
  RELA_LAUNCH_CHECK();
  if (n_pop > 0) {  // the slots just evicted may be read by the gathers above until this point of the stream
    hipEvent_t ev = nullptr;
    if (!r->ev_pool.empty()) {
      ev = r->ev_pool.back();
      r->ev_pool.pop_back();
    } else {
      RELA_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    RELA_HIP(hipEventRecord(ev, r->stream));
    r->evictions.push_back({evict_start, n_pop, ev});
  }
  if (!r->deferred_wait) {
    RELA_HIP(hipEventRecord(r->ev_out, r->stream));
    RELA_HIP(hipStreamWaitEvent(consumer, r->ev_out, 0));
  }
  r->n_sampled = batch;
  const bool hold = n_pop > 0 && !out_rows_dev;  // (see rela_replay::held)
  if (hold) r->held += n_pop;
  lk.unlock();
  if (n_pop > 0 && !hold) r->cv_size.notify_all();
  return RELA_OK;
}

extern "C" int rela_replay_update_priority(rela_replay* r, int n, const float* priority, int on_device,
                                           void* stream_) {
  RELA_CHECK(r && priority, RELA_EINVAL, "rela_replay_update_priority: bad arguments");
  hipStream_t producer = (hipStream_t)stream_;
  DeviceGuard g(r->device);
  std::unique_lock<std::mutex> lk(r->m);
  RELA_CHECK(n == r->n_sampled && n > 0, RELA_ESTATE,
             "rela_replay_update_priority: %d priorities for an outstanding batch of %d", n, r->n_sampled);  // :237
  if (int rc = scan_failed(r, "rela_replay_update_priority")) return rc;
  const float* p = priority;
  if (on_device) {
    RELA_HIP(hipEventRecord(r->ev_in, producer));
    RELA_HIP(hipStreamWaitEvent(r->stream, r->ev_in, 0));
  } else {
    r->stage.begin();
    const hipError_t up = r->stage.h2d(r->d_prio, priority, sizeof(float) * n, r->stream);
    r->stage.end(r->stream);
    RELA_HIP(up);
    p = r->d_prio;
  }
  {
    ProfScope prof("replay_update", r->stream);
    hipLaunchKernelGGL(replay_update, dim3(1), dim3(1024), 0, r->stream, p, n, r->alpha, r->d_ids, r->d_evicted,
                       r->d_w, r->d_state);
  }
  RELA_LAUNCH_CHECK();
  if (on_device && !r->deferred_wait) {
    RELA_HIP(hipEventRecord(r->ev_out, r->stream));
    RELA_HIP(hipStreamWaitEvent(producer, r->ev_out, 0));
  }
  r->n_sampled = 0;  // sampledIds_.clear() :244
  const bool release = r->held > 0;  // the batch is over: whoever read its rows remotely has done so
  r->held = 0;
  lk.unlock();
  if (release) r->cv_size.notify_all();
  return RELA_OK;
}

// Where the bulk of an insert runs.  off (default): on the replay's stream, in commit order with sample / update --
// best when ONE host thread drives actors and learner and has already pipelined the sample path (bench.py: 5.35 M
// env-steps/s against 4.58 M decoupled, the row copies then contend with the learner's kernels instead of taking their
// turn).  on: row copies and priority staging on a second stream, only the weight / sum_ commit in order -- best when
// an INDEPENDENT sampler keeps the replay's stream busy with its latency-bound chain (the threaded drop-in with an
// unthrottled sampler: 0.9 M -> 1.85 M env-steps/s); the `rela` module turns it on.
extern "C" int rela_replay_set_decoupled_insert(rela_replay* r, int on) {
  RELA_CHECK(r, RELA_EINVAL, "rela_replay_set_decoupled_insert: bad arguments");
  DeviceGuard g(r->device);
  std::unique_lock<std::mutex> lk(r->m);
  RELA_HIP(hipStreamSynchronize(r->copy_stream));
  RELA_HIP(hipStreamSynchronize(r->stream));
  r->legacy_insert = on == 0;
  r->copy_stream = on ? r->copy_stream_own : r->stream;
  return RELA_OK;
}

extern "C" int rela_replay_set_deferred_wait(rela_replay* r, int on) {
  RELA_CHECK(r, RELA_EINVAL, "rela_replay_set_deferred_wait: bad arguments");
  std::unique_lock<std::mutex> lk(r->m);
  r->deferred_wait = on != 0;
  return RELA_OK;
}

extern "C" int rela_replay_wait(rela_replay* r, void* stream_) {
  RELA_CHECK(r, RELA_EINVAL, "rela_replay_wait: bad arguments");
  DeviceGuard g(r->device);
  std::unique_lock<std::mutex> lk(r->m);
  RELA_HIP(hipEventRecord(r->ev_wait, r->stream));
  RELA_HIP(hipStreamWaitEvent((hipStream_t)stream_, r->ev_wait, 0));
  return RELA_OK;
}

extern "C" int rela_replay_last_sample_dev(rela_replay* r, const float** raw_w_dev, const float** sum_f_dev) {
  RELA_CHECK(r, RELA_EINVAL, "rela_replay_last_sample_dev: bad arguments");
  if (raw_w_dev) *raw_w_dev = r->d_raw_w;
  if (sum_f_dev) *sum_f_dev = &r->d_state->sum_f;
  return RELA_OK;
}

extern "C" int rela_replay_last_sample_size(const rela_replay* r) { return r ? r->last_full_size : 0; }

extern "C" int rela_replay_shutdown(rela_replay* r) {
  RELA_CHECK(r, RELA_EINVAL, "rela_replay_shutdown: bad arguments");
  {
    std::lock_guard<std::mutex> lk(r->m);
    r->shut = true;
  }
  r->cv_size.notify_all();
  r->cv_tail.notify_all();
  return RELA_OK;
}

extern "C" int rela_replay_size(const rela_replay* r) {
  if (!r) return 0;
  std::lock_guard<std::mutex> lk(r->m);
  return r->safe_size;  // safeSize(nullptr) :245-247
}

extern "C" int64_t rela_replay_num_add(const rela_replay* r) { return r ? r->num_add.load() : 0; }

extern "C" int rela_replay_limits(const rela_replay* r, int* capacity, int* ring) {
  RELA_CHECK(r, RELA_EINVAL, "rela_replay_limits: bad arguments");
  if (capacity) *capacity = r->capacity;
  if (ring) *ring = r->ring;
  return RELA_OK;
}

extern "C" int rela_replay_debug_state(rela_replay* r, rela_replay_state* out, int32_t* ids_host, float* raw_w_host,
                                       float* targets_host) {
  RELA_CHECK(r && out, RELA_EINVAL, "rela_replay_debug_state: bad arguments");
  DeviceGuard g(r->device);
  std::lock_guard<std::mutex> lk(r->m);
  ReplayDevState st;
  RELA_HIP(hipMemcpyAsync(&st, r->d_state, sizeof(st), hipMemcpyDeviceToHost, r->stream));
  const int n = r->n_sampled;
  if (n > 0 && ids_host) RELA_HIP(hipMemcpyAsync(ids_host, r->d_ids, sizeof(int32_t) * n, hipMemcpyDeviceToHost, r->stream));
  if (n > 0 && raw_w_host)
    RELA_HIP(hipMemcpyAsync(raw_w_host, r->d_raw_w, sizeof(float) * n, hipMemcpyDeviceToHost, r->stream));
  if (n > 0 && targets_host)
    RELA_HIP(hipMemcpyAsync(targets_host, r->d_targets, sizeof(float) * n, hipMemcpyDeviceToHost, r->stream));
  RELA_HIP(hipStreamSynchronize(r->stream));
  out->head = r->head;
  out->tail = r->tail;
  out->size = r->size;
  out->safe_size = r->safe_size;
  out->ring = r->ring;
  out->n_sampled = n;
  out->num_add = r->num_add.load();
  out->sum = st.sum;
  out->dev_error = st.err;
  out->pad = 0;
  return RELA_OK;
}

extern "C" int rela_replay_debug_weights(rela_replay* r, float* weights_host, uint8_t* evicted_host) {
  RELA_CHECK(r, RELA_EINVAL, "rela_replay_debug_weights: bad arguments");
  DeviceGuard g(r->device);
  std::lock_guard<std::mutex> lk(r->m);
  if (weights_host)
    RELA_HIP(hipMemcpyAsync(weights_host, r->d_w, sizeof(float) * (size_t)r->ring, hipMemcpyDeviceToHost, r->stream));
  if (evicted_host)
    RELA_HIP(hipMemcpyAsync(evicted_host, r->d_evicted, (size_t)r->ring, hipMemcpyDeviceToHost, r->stream));
  RELA_HIP(hipStreamSynchronize(r->stream));
  return RELA_OK;
}

extern "C" int rela_replay_debug_read_rows(rela_replay* r, int field, int slot, int count, void* rows_host) {
  RELA_CHECK(r && rows_host && field >= 0 && field < (int)r->d_fields.size() && slot >= 0 && count >= 0 &&
                 slot + count <= r->ring,
             RELA_EINVAL, "rela_replay_debug_read_rows: bad arguments");
  DeviceGuard g(r->device);
  std::lock_guard<std::mutex> lk(r->m);
  const size_t rb = (size_t)r->row_bytes[field];
  RELA_HIP(hipStreamSynchronize(r->copy_stream));
  RELA_HIP(hipMemcpyAsync(rows_host, r->d_fields[field] + (size_t)slot * rb, rb * (size_t)count, hipMemcpyDeviceToHost,
                          r->stream));
  RELA_HIP(hipStreamSynchronize(r->stream));
  return RELA_OK;
}

// Test tap: out[i] = torch::pow(x, exponent)[i] as the replay evaluates it for a tensor of n elements.
namespace rela_amd {
namespace {
__global__ void debug_pow_kernel(const float* __restrict__ x, int n, float ex, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = pow_aten(x[i], ex, i, n);
}
}  // namespace
}  // namespace rela_amd
extern "C" int rela_debug_pow(const float* x_dev, int n, float exponent, float* out_dev, void* stream_) {
  RELA_CHECK(x_dev && out_dev && n >= 0, RELA_EINVAL, "rela_debug_pow: bad arguments");
  if (n > 0)
    hipLaunchKernelGGL(debug_pow_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream_, x_dev, n, exponent,
                       out_dev);
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}


// =====================================================================================================================
// Native partition exchange (rela_amd/parallel.py): the learner process maps a partition's field arrays, the
// ids / raw weights of its last sample and its device state through HIP IPC handles and GATHERS the sampled rows
// itself -- a kernel on the learner's GPU reading the owner's HBM directly (xGMI peer reads between GPUs, plain reads
// when both processes share a GPU) -- instead of the owner packing the rows and a collective moving them (28.9 MB per
// Ape-X step, 222 MB per R2D2 step).  Ordering between the two processes stays with the caller: the owner's sample
// (rela_replay_sample with out_rows_dev = NULL: ids, raw weights, eviction; no gather) must have completed before the
// gather starts, and the owner must not sample or update again before the gather completed.
// =====================================================================================================================
struct rela_replay_remote {
  int device = 0;
  int nfields = 0, ring = 0, max_batch = 0;
  int64_t row_bytes[RELA_IPC_MAX_FIELDS] = {};
  int32_t steps[RELA_IPC_MAX_FIELDS] = {};
  uint8_t* fields[RELA_IPC_MAX_FIELDS] = {};
  VmmRange* vmm[RELA_IPC_MAX_FIELDS] = {};  // fields that arrived as chunks (fields[f] = vmm[f]->base)
  // de-duplicated partition: the two stack fields hold references into the owner's unit ring
  int dd_ups = 0, dd_field[2] = {-1, -1};
  int64_t dd_unit_bytes = 0;
  uint8_t* units = nullptr;
  VmmRange* vmm_units = nullptr;
  int32_t* ids = nullptr;
  float* raw_w = nullptr;
  ReplayDevState* state = nullptr;
};

// `chunks` == nullptr: the plain descriptor (every field one hipIpcMemHandle_t)
static int export_partition(rela_replay* r, rela_replay_ipc_desc* out, rela_replay_chunk_desc* chunks, int* fds_out,
                            int max_fds, const char* who) {
  RELA_CHECK(!r->d_fields.empty() && (int)r->d_fields.size() <= RELA_IPC_MAX_FIELDS, RELA_ESTATE,
             "%s: set the schema first (at most %d fields)", who, RELA_IPC_MAX_FIELDS);
  RELA_CHECK(r->dd_ups == 0 || chunks, RELA_EINVAL,
             "%s: a de-duplicated partition needs the unit ring next to its fields; export it with rela_replay_export_chunks", who);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  DeviceGuard g(r->device);
  std::lock_guard<std::mutex> lk(r->m);
  memset(out, 0, sizeof(*out));
  out->abi = 1;
  out->nfields = (int32_t)r->d_fields.size();
  out->ring = r->ring;
  out->device = r->device;
  out->max_batch = kMaxBatch;
  int nfds = 0;
  auto fail = [&](int rc) {
    for (int k = 0; k < nfds; ++k) (void)::close(fds_out[k]);
    return rc;
  };
  for (size_t f = 0; f < r->d_fields.size(); ++f) {
    out->row_bytes[f] = r->row_bytes[f];
    out->steps[f] = r->steps[f];
    if (VmmRange* v = r->vmm[f]) {
      if (!chunks) {
        set_last_error("%s: field %d is made of %d chunks (rela_replay_set_chunk_bytes); export it with "
                       "rela_replay_export_chunks", who, (int)f, (int)v->handles.size());
        return RELA_EINVAL;
      }
      const int n = (int)v->handles.size();
      if (nfds + n > max_fds) {
        set_last_error("%s: more than %d chunk descriptors; use larger chunks", who, max_fds);
        return fail(RELA_EINVAL);
      }
      hipError_t e = v->export_fds(fds_out + nfds);
      if (e != hipSuccess) {
        set_last_error("%s: hipMemExportToShareableHandle (field %d): %s", who, (int)f, hipGetErrorString(e));
        return fail(RELA_ENODEV);
      }
      nfds += n;
      chunks->field_chunks[f] = n;
      chunks->chunk_bytes[f] = (int64_t)v->chunk;
      chunks->mapped_bytes[f] = (int64_t)v->bytes;
      continue;
    }
    {
      // r4, this pool's boxes: hipIpcOpenMemHandle of a 37 GB allocation (one frame-stack field of a 2^20-row partition)
      // did not return within 200 s in the importing process, 18.5 GB (2^19 rows) maps in under a second: refuse loudly
      hipDeviceptr_t base = nullptr;
      size_t bytes = 0;
      if (hipMemGetAddressRange(&base, &bytes, r->d_fields[f]) == hipSuccess && bytes > ((size_t)24 << 30)) {
        set_last_error("%s: field %d is one allocation of %.1f GB; HIP IPC imports above ~24 GB do not return on this "
                       "platform -- create the partition with rela_replay_set_chunk_bytes (or RELA_REPLAY_CHUNK_GB) and "
                       "export it with rela_replay_export_chunks", who, (int)f, (double)bytes / 1e9);
        return fail(RELA_EINVAL);
      }
    }
    hipError_t e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(out->field_handle[f]), r->d_fields[f]);
    if (e != hipSuccess) {
      set_last_error("%s: hipIpcGetMemHandle (field %d): %s", who, (int)f, hipGetErrorString(e));
      return fail(RELA_ENODEV);
    }
  }
  hipError_t e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(out->ids_handle), r->d_ids);
  if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(out->raw_w_handle), r->d_raw_w);
  if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(out->state_handle), r->d_state);
  if (e != hipSuccess) {
    set_last_error("%s: hipIpcGetMemHandle: %s", who, hipGetErrorString(e));
    return fail(RELA_ENODEV);
  }
  if (chunks && r->dd_ups > 0) {  // the unit ring the two stack fields refer into
    chunks->dd_ups = r->dd_ups;
    chunks->dd_field[0] = r->dd_field[0], chunks->dd_field[1] = r->dd_field[1];
    chunks->dd_unit_bytes = r->dd_unit_bytes;
    chunks->dd_cap = r->dd_cap;
    if (VmmRange* v = r->vmm_units) {
      const int n = (int)v->handles.size();
      if (nfds + n > max_fds) {
        set_last_error("%s: more than %d chunk descriptors; use larger chunks", who, max_fds);
        return fail(RELA_EINVAL);
      }
      e = v->export_fds(fds_out + nfds);
      if (e != hipSuccess) {
        set_last_error("%s: hipMemExportToShareableHandle (unit ring): %s", who, hipGetErrorString(e));
        return fail(RELA_ENODEV);
      }
      nfds += n;
      chunks->units_chunks = n;
      chunks->units_chunk_bytes = (int64_t)v->chunk;
      chunks->units_mapped_bytes = (int64_t)v->bytes;
    } else {
      e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(chunks->units_handle), r->d_units);
      if (e != hipSuccess) {
        set_last_error("%s: hipIpcGetMemHandle (unit ring): %s", who, hipGetErrorString(e));
        return fail(RELA_ENODEV);
      }
    }
  }
  if (chunks) chunks->nfds = nfds;
  return RELA_OK;
}

extern "C" int rela_replay_export_ipc(rela_replay* r, rela_replay_ipc_desc* out) {
  RELA_CHECK(r && out, RELA_EINVAL, "rela_replay_export_ipc: bad arguments");
  return export_partition(r, out, nullptr, nullptr, 0, "rela_replay_export_ipc");
}

extern "C" int rela_replay_export_chunks(rela_replay* r, rela_replay_chunk_desc* out, int* fds_out, int max_fds) {
  RELA_CHECK(r && out && (fds_out || max_fds == 0) && max_fds >= 0, RELA_EINVAL, "rela_replay_export_chunks: bad arguments");
  memset(out, 0, sizeof(*out));
  const int rc = export_partition(r, &out->ipc, out, fds_out, max_fds, "rela_replay_export_chunks");
  if (rc == RELA_OK) out->abi = 2;
  return rc;
}

static int import_partition(rela_replay_remote** out, const rela_replay_ipc_desc* desc, const rela_replay_chunk_desc* chunks,
                            const int* fds, int nfds, int device, const char* who) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_last_error("%s: HIP device %d not available (%d visible); there is no CPU path", who, device, ndev);
    return RELA_ENODEV;
  }
  DeviceGuard g(device);
  auto* rr = new rela_replay_remote();
  rr->device = device;
  rr->nfields = desc->nfields, rr->ring = desc->ring, rr->max_batch = desc->max_batch;
  auto open = [&](const unsigned char* h, void** p) {
    hipIpcMemHandle_t mh;
    memcpy(&mh, h, sizeof(mh));
    return hipIpcOpenMemHandle(p, mh, hipIpcMemLazyEnablePeerAccess);
  };
  hipError_t e = hipSuccess;
  const char* what = "hipIpcOpenMemHandle";
  int fd_at = 0;
  // RELA_IPC_TRACE=1: one line per mapping step on stderr (an import that does not return is this platform's failure mode)
  const bool trace = getenv("RELA_IPC_TRACE") != nullptr;
  for (int f = 0; f < desc->nfields && e == hipSuccess; ++f) {
    rr->row_bytes[f] = desc->row_bytes[f];
    rr->steps[f] = desc->steps[f];
    const int n = chunks ? chunks->field_chunks[f] : 0;
    if (trace)
      fprintf(stderr, "[%s] field %d: %.3f GB as %s\n", who, f, (double)desc->row_bytes[f] * desc->ring / 1e9,
              n > 0 ? "chunks" : "one IPC handle");
    if (n > 0) {
      const int64_t need = desc->row_bytes[f] * (int64_t)desc->ring;
      if (fd_at + n > nfds || chunks->chunk_bytes[f] <= 0 || chunks->mapped_bytes[f] < need ||
          (chunks->mapped_bytes[f] + chunks->chunk_bytes[f] - 1) / chunks->chunk_bytes[f] != n) {
        set_last_error("%s: field %d: %d chunks of %lld bytes do not describe %lld mapped bytes (%d descriptors given)", who, f,
                       n, (long long)chunks->chunk_bytes[f], (long long)chunks->mapped_bytes[f], nfds);
        rela_replay_remote_close(rr);
        return RELA_EINVAL;
      }
      rr->vmm[f] = new VmmRange();
      what = "mapping the chunks (hipMemImportFromShareableHandle / hipMemMap / hipMemSetAccess)";
      e = rr->vmm[f]->import(fds + fd_at, n, (size_t)chunks->chunk_bytes[f], (size_t)chunks->mapped_bytes[f], device);
      rr->fields[f] = rr->vmm[f]->base;
      fd_at += n;
    } else {
      e = open(desc->field_handle[f], reinterpret_cast<void**>(&rr->fields[f]));
    }
  }
  if (e == hipSuccess && chunks && chunks->dd_ups > 0) {
    rr->dd_ups = chunks->dd_ups, rr->dd_unit_bytes = chunks->dd_unit_bytes;
    rr->dd_field[0] = chunks->dd_field[0], rr->dd_field[1] = chunks->dd_field[1];
    const int n = chunks->units_chunks;
    const int64_t need = chunks->dd_cap * chunks->dd_unit_bytes;
    const bool fields_ok = rr->dd_field[0] >= 0 && rr->dd_field[0] < desc->nfields && rr->dd_field[1] >= 0 &&
                           rr->dd_field[1] < desc->nfields && rr->dd_ups <= 16 && rr->dd_unit_bytes > 0 && rr->dd_unit_bytes % 16 == 0 &&
                           desc->row_bytes[rr->dd_field[0]] == 4 * rr->dd_ups && desc->row_bytes[rr->dd_field[1]] == 4 * rr->dd_ups;
    if (!fields_ok || (n > 0 && (fd_at + n > nfds || chunks->units_chunk_bytes <= 0 || chunks->units_mapped_bytes < need ||
                                 (chunks->units_mapped_bytes + chunks->units_chunk_bytes - 1) / chunks->units_chunk_bytes != n))) {
      set_last_error("%s: the unit ring's description is inconsistent (%d chunks, fields %d / %d, %d units per stack)", who, n,
                     rr->dd_field[0], rr->dd_field[1], rr->dd_ups);
      rela_replay_remote_close(rr);
      return RELA_EINVAL;
    }
    if (trace) fprintf(stderr, "[%s] unit ring: %.3f GB as %s\n", who, (double)need / 1e9, n > 0 ? "chunks" : "one IPC handle");
    if (n > 0) {
      rr->vmm_units = new VmmRange();
      what = "mapping the unit ring's chunks";
      e = rr->vmm_units->import(fds + fd_at, n, (size_t)chunks->units_chunk_bytes, (size_t)chunks->units_mapped_bytes, device);
      rr->units = rr->vmm_units->base;
      fd_at += n;
    } else {
      e = open(chunks->units_handle, reinterpret_cast<void**>(&rr->units));
    }
  }
  if (trace) fprintf(stderr, "[%s] fields mapped (%s); ids / weights / state\n", who, hipGetErrorString(e));
  if (e == hipSuccess) what = "hipIpcOpenMemHandle", e = open(desc->ids_handle, reinterpret_cast<void**>(&rr->ids));
  if (e == hipSuccess) e = open(desc->raw_w_handle, reinterpret_cast<void**>(&rr->raw_w));
  if (e == hipSuccess) e = open(desc->state_handle, reinterpret_cast<void**>(&rr->state));
  if (e != hipSuccess) {
    set_last_error("%s: %s failed: %s", who, what, hipGetErrorString(e));
    rela_replay_remote_close(rr);
    return RELA_ENODEV;
  }
  *out = rr;
  return RELA_OK;
}

extern "C" int rela_replay_import_ipc(rela_replay_remote** out, const rela_replay_ipc_desc* desc, int device) {
  RELA_CHECK(out && desc && desc->abi == 1 && desc->nfields >= 1 && desc->nfields <= RELA_IPC_MAX_FIELDS, RELA_EINVAL,
             "rela_replay_import_ipc: bad descriptor");
  return import_partition(out, desc, nullptr, nullptr, 0, device, "rela_replay_import_ipc");
}

extern "C" int rela_replay_import_chunks(rela_replay_remote** out, const rela_replay_chunk_desc* desc, const int* fds, int nfds,
                                         int device) {
  RELA_CHECK(out && desc && desc->abi == 2 && desc->ipc.abi == 1 && desc->ipc.nfields >= 1 &&
                 desc->ipc.nfields <= RELA_IPC_MAX_FIELDS && nfds == desc->nfds && (fds || nfds == 0),
             RELA_EINVAL, "rela_replay_import_chunks: bad descriptor (or %d descriptors for its %d chunks)", nfds,
             desc ? desc->nfds : -1);
  return import_partition(out, &desc->ipc, desc, fds, nfds, device, "rela_replay_import_chunks");
}

extern "C" void rela_replay_remote_close(rela_replay_remote* rr) {
  if (!rr) return;
  DeviceGuard g(rr->device);
  (void)hipDeviceSynchronize();
  for (int f = 0; f < rr->nfields; ++f) {
    if (rr->vmm[f]) {
      rr->vmm[f]->destroy();
      delete rr->vmm[f];
    } else if (rr->fields[f]) {
      (void)hipIpcCloseMemHandle(rr->fields[f]);
    }
  }
  if (rr->vmm_units) {
    rr->vmm_units->destroy();
    delete rr->vmm_units;
  } else if (rr->units) {
    (void)hipIpcCloseMemHandle(rr->units);
  }
  if (rr->ids) (void)hipIpcCloseMemHandle(rr->ids);
  if (rr->raw_w) (void)hipIpcCloseMemHandle(rr->raw_w);
  if (rr->state) (void)hipIpcCloseMemHandle(rr->state);
  delete rr;
}

extern "C" int rela_replay_remote_gather(rela_replay_remote* rr, int batch, void* const* out_rows_dev, float* raw_w_out,
                                         float* sum_f_out, int out_batch, int out_offset, void* stream_) {
  RELA_CHECK(rr && batch > 0 && batch <= rr->max_batch && out_rows_dev, RELA_EINVAL, "rela_replay_remote_gather: bad arguments");
  if (out_batch <= 0) out_batch = batch, out_offset = 0;
  RELA_CHECK(out_offset >= 0 && out_offset + batch <= out_batch, RELA_EINVAL,
             "rela_replay_remote_gather: rows [%d, %d) do not fit an output of %d", out_offset, out_offset + batch, out_batch);
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(rr->device);
  SmallFields small{};
  BigFields big{};
  int nbig = 0, max_y = 1;
  int64_t max_units = 1;
  for (int f = 0; f < rr->nfields; ++f) {
    if (!out_rows_dev[f]) continue;
    const int64_t rb = rr->row_bytes[f];
    const int st = rr->steps[f];
    if (rr->dd_ups > 0 && (f == rr->dd_field[0] || f == rr->dd_field[1])) {
      RELA_CHECK(((uintptr_t)out_rows_dev[f] & 15) == 0, RELA_EINVAL, "rela_replay_remote_gather: unaligned stack output");
      continue;  // rebuilt from the unit ring below
    }
    if (st == 1 && rb <= kSmallRowBytes && small.n < kMaxSmallFields) {
      small.src[small.n] = rr->fields[f];
      small.dst[small.n] = (uint8_t*)out_rows_dev[f];
      small.row_bytes[small.n] = (int32_t)rb;
      small.n += 1;
      continue;
    }
    RELA_CHECK(nbig < kMaxBigFields, RELA_EINVAL, "rela_replay_remote_gather: more than %d large fields", kMaxBigFields);
    const int64_t sub = rb / st;
    const int v16 = vec16_ok(out_rows_dev[f], rr->fields[f], sub) && (rb % 16 == 0);
    big.out_batch = out_batch, big.out_off = out_offset;
    big.field[nbig] = rr->fields[f];
    big.out[nbig] = (uint8_t*)out_rows_dev[f];
    big.slot_bytes[nbig] = rb;
    big.steps[nbig] = st;
    big.vec16[nbig] = v16;
    max_units = std::max<int64_t>(max_units, v16 ? (sub >> 4) : sub);
    max_y = std::max(max_y, batch * st);
    nbig += 1;
  }
  {
    ProfScope prof("replay_gather_remote", s);
    hipLaunchKernelGGL(replay_gather_small_remote, dim3(1 + small.n), dim3(256), 0, s, small, (const int32_t*)rr->ids, batch,
                       (const float*)rr->raw_w, (const ReplayDevState*)rr->state, raw_w_out, sum_f_out, out_offset);
    if (nbig > 0) {
      const int gx = (int)std::min<int64_t>(std::max<int64_t>(1, (max_units + kThreads - 1) / kThreads), 64);
      hipLaunchKernelGGL(replay_gather_big, dim3(gx, std::min(max_y, 32768), nbig), dim3(kThreads), 0, s, big,
                         (const int32_t*)rr->ids, batch);
    }
    for (int q = 0; q < 2 && rr->dd_ups > 0; ++q) {  // stacks out of the owner's unit ring (as rela_replay_sample does locally)
      const int f = rr->dd_field[q];
      if (!out_rows_dev[f]) continue;
      const int64_t nv = rr->dd_unit_bytes >> 4;
      const int gx = (int)std::min<int64_t>(std::max<int64_t>(1, (nv + kThreads - 1) / kThreads), 64);
      uint8_t* out = (uint8_t*)out_rows_dev[f] + (int64_t)out_offset * rr->dd_ups * rr->dd_unit_bytes;
      hipLaunchKernelGGL(replay_gather_dedup, dim3(gx, std::min(batch * rr->dd_ups, 32768)), dim3(kThreads), 0, s,
                         (const int32_t*)rr->fields[f], (const int32_t*)rr->ids, (const uint8_t*)rr->units, rr->dd_unit_bytes,
                         rr->dd_ups, out, batch);
    }
  }
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}


// Generic form for buffers this library allocated (the learners' flat parameter buffers): the weight publish of
// rela_amd/parallel.py maps the learner's buffer into every actor process, which then loads its nets straight from it
// (rela_ffnet_load / rela_lstmnet_load with on_device = 1: a peer read over xGMI) instead of receiving a broadcast.
extern "C" int rela_ipc_export_buffer(const void* dev_ptr, unsigned char handle_out[64]) {
  RELA_CHECK(dev_ptr && handle_out, RELA_EINVAL, "rela_ipc_export_buffer: bad arguments");
  RELA_HIP(hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle_out), const_cast<void*>(dev_ptr)));
  return RELA_OK;
}
extern "C" int rela_ipc_import_buffer(const unsigned char handle[64], void** dev_ptr_out, int device) {
  RELA_CHECK(handle && dev_ptr_out, RELA_EINVAL, "rela_ipc_import_buffer: bad arguments");
  DeviceGuard g(device);
  RELA_CHECK(g.ok, RELA_ENODEV, "rela_ipc_import_buffer: HIP device %d not available; there is no CPU path", device);
  hipIpcMemHandle_t mh;
  memcpy(&mh, handle, sizeof(mh));
  RELA_HIP(hipIpcOpenMemHandle(dev_ptr_out, mh, hipIpcMemLazyEnablePeerAccess));
  return RELA_OK;
}
extern "C" int rela_ipc_close_buffer(void* dev_ptr, int device) {
  if (!dev_ptr) return RELA_OK;
  DeviceGuard g(device);
  RELA_HIP(hipIpcCloseMemHandle(dev_ptr));
  return RELA_OK;
}
