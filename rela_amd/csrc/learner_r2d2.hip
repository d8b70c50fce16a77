// R2D2 learner step on the device: pyrela/main.py:206-251 with R2D2Agent.loss (pyrela/r2d2.py:189-206),
// td_err (:122-187) and AtariLSTMNet.unroll_rnn / forward (pyrela/net.py:127-163), without PyTorch autograd
// (SURVEY 8a G2 / K15, 8f-2):
//
//   forward   per net (target first, then online; one set of trunk buffers):
//               conv trunk over ALL T*B frames of the batch at once (T = burn_in + seq_len + multi_step)
//               GX = a3 x W_ih^T + (b_ih + b_hh) for all T*B rows in ONE GEMM            (K = 3136)
//               T recurrent steps: gates_t = GX_t + h_{t-1} x W_hh^T (split-K GEMM, K = 512) -> LSTM cell
//               (torch gate order i,f,g,o); the burn-in steps carry no gradient (:144-147) and the state
//               after them is zeroed where terminal[burn_in-1] (dummy burn-in at an episode start, :149-154)
//               dueling heads + Q over the seq_len + multi_step training steps
//   td        batch-global q.min() of the online Q (net.py:160), greedy action, Q_online[a], Q_target[greedy],
//             target_i = r_i + bootstrap_i * gamma^n * Q_target[i + n], pad mask i >= seq_len_b - burn_in (:169-183),
//             Huber summed over the sequence (:197-203), eta-mixed priority (:103-120)
//   backward  d(mean(loss * w)) through the heads, BPTT over the seq_len + multi_step training steps
//             (dh_{t-1} = dgates_t x W_hh, split-K), then batched over all training rows:
//             dW_hh = DG^T x H_prev, dW_ih = DG^T x a3, d_a3 = DG x W_ih (ReLU mask) and the shared conv-trunk
//             backward of learner_common.h.  Every contraction is an instance of gemm_lds (f32 MFMA).
//   update    clip_grad_norm_ + Adam (pyrela/main.py:124-126,233-238) on flat buffers in state_dict order.
#include "gemm_bf16s.h"
#include "learner_common.h"

using namespace rela_amd;

namespace rela_amd {
namespace {

constexpr int kHid = 512, kGates = 2048, kFeat = 3136;
constexpr int kRecSplitF = 4;  // split-K of the recurrent forward GEMM (K = 512)
constexpr int kRecSplitB = 8;  // split-K of the recurrent backward GEMM (K = 2048)

using TileRows = TileCfg<128, 64, 4, 2, false>;  // M = many rows, A k-contiguous
using TileRec = TileCfg<64, 64, 2, 4, false>;    // M = batch rows of one time step
using TileWg = TileCfg<128, 64, 4, 2, true>;     // weight gradients, M = 2048 gate rows
using TileW32r = TileCfg<32, 64, 2, 4, true>;    // head weight gradients (M = 32)

// GX[r][g] = bias[g] + sum_k a3[r][k] * wihT[k][g]
struct ProbGateX : ProbBase {
  const float *a3, *wihT, *bias;
  float* gx;
  __device__ float4 loadA(int m, int k) const { return m < M ? ld4(a3 + (size_t)m * kFeat + k) : zero4(); }
  __device__ float4 loadB(int k, int n) const { return ld4(wihT + (size_t)k * kGates + n); }
  __device__ void store(int, int m, int n, float v) const { gx[(size_t)m * kGates + n] = v + bias[n]; }
};
// part[z][b][g] = sum_{k in slice z} h[b][k] * whhT[k][g]
struct ProbRecFwd : ProbBase {
  const float *h, *whhT;
  float* part;
  __device__ float4 loadA(int m, int k) const { return m < M ? ld4(h + (size_t)m * kHid + k) : zero4(); }
  __device__ float4 loadB(int k, int n) const { return ld4(whhT + (size_t)k * kGates + n); }
  __device__ void store(int z, int m, int n, float v) const { part[((size_t)z * M + m) * kGates + n] = v; }
};
// part[z][b][k] = sum_{g in slice z} dg[b][g] * whh[g][k]
struct ProbRecBwd : ProbBase {
  const float *dg, *whh;
  float* part;
  __device__ float4 loadA(int m, int k) const { return m < M ? ld4(dg + (size_t)m * kGates + k) : zero4(); }
  __device__ float4 loadB(int k, int n) const { return ld4(whh + (size_t)k * kHid + n); }
  __device__ void store(int z, int m, int n, float v) const { part[((size_t)z * M + m) * kHid + n] = v; }
};
// dW_hh[g][k] = sum_r dg[r][g] * hprev[r][k]
struct ProbWhh : ProbBase {
  const float *dg, *hprev;
  float* out;
  __device__ float4 loadA(int r, int m) const { return r < K ? ld4(dg + (size_t)r * kGates + m) : zero4(); }
  __device__ float4 loadB(int r, int n) const { return r < K ? ld4(hprev + (size_t)r * kHid + n) : zero4(); }
  __device__ void store(int, int m, int n, float v) const { out[(size_t)m * kHid + n] = v; }
};
// dW_ih[g][c*49+pos] = sum_r dg[r][g] * a3[r][pos*64+c]     (written in state_dict order, net.py:105-106)
struct ProbWih : ProbBase {
  const float *dg, *a3;
  float* out;
  __device__ float4 loadA(int r, int m) const { return r < K ? ld4(dg + (size_t)r * kGates + m) : zero4(); }
  __device__ float4 loadB(int r, int n) const { return r < K ? ld4(a3 + (size_t)r * kFeat + n) : zero4(); }
  __device__ void store(int, int m, int n, float v) const {
    const int pos = n >> 6, c = n & 63;
    out[(size_t)m * kFeat + c * 49 + pos] = v;
  }
};
// d_a3[r][j] = relu'(a3) * sum_g dg[r][g] * wihp[g][j]      j = pos*64 + c
struct ProbIhDgrad : ProbBase {
  const float *dg, *wihp, *a3;
  float* d_a3;
  __device__ float4 loadA(int m, int k) const { return m < M ? ld4(dg + (size_t)m * kGates + k) : zero4(); }
  __device__ float4 loadB(int k, int n) const { return ld4(wihp + (size_t)k * kFeat + n); }
  __device__ void store(int, int m, int n, float v) const {
    const size_t i = (size_t)m * kFeat + n;
    d_a3[i] = a3[i] > 0.f ? v : 0.f;
  }
};
// d_o[r][u] = sum_k d_ha[r][k] * Wh[k][u]      Wh rows: 0..A-1 = fc_a.weight, 31 = fc_v.weight (no ReLU on o)
struct ProbHeadDgradSeq : ProbBase {
  const float *d_ha, *a_w, *v_w;
  float* d_o;
  int A;
  __device__ float4 loadA(int m, int k) const { return m < M ? ld4(d_ha + (size_t)m * 32 + k) : zero4(); }
  __device__ float4 loadB(int k, int n) const {
    if (k < A) return ld4(a_w + (size_t)k * kHid + n);
    if (k == 31) return ld4(v_w + n);
    return zero4();
  }
  __device__ void store(int, int m, int n, float v) const { d_o[(size_t)m * kHid + n] = v; }
};

// ---- weight copies in the orders the GEMM loaders read ------------------------------------------
// wihT[k = pos*64+c][g] and wihp[g][k = pos*64+c]  <-  weight_ih_l0[g][c*49+pos]; whhT[k][g] <- weight_hh_l0[g][k]
// One block per 8 gate rows: the rows are read once, coalesced, into LDS (8 x 12.5 KB) and leave as coalesced rows of
// wihp and as 32-byte pieces of wihT / whhT (the one-thread-per-element form read with a stride of 49 floats and wrote
// wihT with a stride of 8 KB: 93 us per call, 0.13 ms of every learner step).
constexpr int kPackRows = 8;
__global__ __launch_bounds__(256) void pack_lstm_learner(const float* __restrict__ wih, const float* __restrict__ whh,
                                                         const float* __restrict__ bih, const float* __restrict__ bhh,
                                                         float* __restrict__ wihT, float* __restrict__ wihp,
                                                         float* __restrict__ whhT, float* __restrict__ bsum) {
  extern __shared__ __attribute__((aligned(16))) float rows[];  // [kPackRows][kFeat]
  const int g0 = blockIdx.x * kPackRows, tid = threadIdx.x;
  for (int i = tid; i < kPackRows * kFeat / 4; i += 256)
    reinterpret_cast<float4*>(rows)[i] = reinterpret_cast<const float4*>(wih + (size_t)g0 * kFeat)[i];
  __syncthreads();
  if (wihp) {
    for (int i = tid; i < kPackRows * kFeat; i += 256) {
      const int r = i / kFeat, k = i - r * kFeat;
      wihp[(size_t)(g0 + r) * kFeat + k] = rows[r * kFeat + (k & 63) * 49 + (k >> 6)];
    }
  }
  for (int i = tid; i < kPackRows * kFeat; i += 256) {  // (k, r) with r fastest: 8 consecutive gate rows per k
    const int k = i / kPackRows, r = i - k * kPackRows;
    wihT[(size_t)k * kGates + g0 + r] = rows[r * kFeat + (k & 63) * 49 + (k >> 6)];
  }
  __syncthreads();
  for (int i = tid; i < kPackRows * kHid / 4; i += 256)
    reinterpret_cast<float4*>(rows)[i] = reinterpret_cast<const float4*>(whh + (size_t)g0 * kHid)[i];
  __syncthreads();
  for (int i = tid; i < kPackRows * kHid; i += 256) {
    const int k = i / kPackRows, r = i - k * kPackRows;
    whhT[(size_t)k * kGates + g0 + r] = rows[r * kHid + k];
  }
  if (tid < kPackRows) bsum[g0 + tid] = bih[g0 + tid] + bhh[g0 + tid];
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// one LSTM time step for Bn rows: pre-activations = gx (x-part + bias) + sum of the split-K partials of
// h_{t-1} x W_hh^T; torch cell (gate order i,f,g,o).  save != 0 leaves the ACTIVATED gates in gx.
__global__ void lstm_cell_fwd(float* __restrict__ gx, const float* __restrict__ part, int nsplit, int Bn,
                              const float* __restrict__ c_prev, float* __restrict__ c_out, float* __restrict__ h_out,
                              int save) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Bn * kHid) return;
  const int b = idx / kHid, u = idx - b * kHid;
  float pre[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const size_t o = (size_t)b * kGates + q * kHid + u;
    float v = gx[o];
    for (int z = 0; z < nsplit; ++z) v += part[(size_t)z * Bn * kGates + o];
    pre[q] = v;
  }
  const float gi = sigm(pre[0]), gf = sigm(pre[1]), gg = tanhf(pre[2]), go = sigm(pre[3]);
  const float c = gf * c_prev[idx] + gi * gg;
  c_out[idx] = c;
  h_out[idx] = go * tanhf(c);
  if (save) {
    float* row = gx + (size_t)b * kGates + u;
    row[0] = gi, row[kHid] = gf, row[2 * kHid] = gg, row[3 * kHid] = go;
  }
}

// Reads of bytes another CU wrote inside this launch.  With an acquire fence (`buffer_inv sc1` + the wait for it: ~1.7 us
// per barrier) plain loads are fine (SC1 = false: BPTT); WITHOUT it every such load must bypass this CU's L1: a buffer
// load with the sc1 bit (SC1 = true: the forward chains) -- sc1 stores, every storing wave drained, ONE lane of each
// workgroup adding to an agent-scope counter behind a workgroup barrier, the consumer's lane polling it and the other
// waves loading behind a workgroup barrier.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sc1_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <bool SC1>
__device__ __forceinline__ float4 ld4_shared(__amdgpu_buffer_rsrc_t rs, const float* base, size_t index) {
  if constexpr (SC1) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(index * 4), 0, 16);  // aux 16 = sc1
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
  } else {
    return *reinterpret_cast<const float4*>(base + index);
  }
}

// ---- persistent recurrent forward: XCD-local chains (batches of up to 64 rows) ---------------------------------
// The T recurrent steps of BOTH nets in ONE launch (2 x 123 x 3 launches of ~13 us each before r2: launch-bound).
// A CHAIN is one net x one tile of 16 batch rows (2 x 4 chains), run by the 32 blocks whose ids are equal mod 8 -- the
// dispatcher deals blocks round-robin over the XCDs, so a chain shares an L2 -- each block owning 16 hidden units (64
// gate columns: i, f, g, o of each) with its slice of W_hh^T in registers (wave w: k in [64 w, 64 w + 64), one f32 MFMA
// B-fragment register per k-step).  Per step every wave multiplies its k-slice of h_{t-1} into a partial [16 x 16]
// tile on v_mfma_f32_16x16x4_f32, the eight partials are added in wave order onto the x-part in LDS, one thread per
// (row, unit) runs the cell with c_t in a register and stores h_t WRITE-THROUGH (agent-scope atomic stores: visible
// to the other CUs without a release fence).  Steps are separated by a chain barrier: one arrival counter per chain
// and step (zeroed ahead of the launch, 32 arrivals), ONE lane polls it relaxed, then h_{t-1} is read with sc1 buffer
// loads and no acquire fence (every storing wave drained its stores before the workgroup barrier that precedes the
// arrival; 0.79 -> 0.56 ms against plain loads behind an agent-scope acquire).  The protocol is placement-
// INDEPENDENT: co-location only decides whether the h a block reads is still in its XCD's L2, never correctness, and
// the sc1 loads never allocate in the per-CU L1, so a second chain block on the same CU cannot read a stale line
// (the hand-off form and its measured envelope: the CDNA4 guide shipped with the build image, "hand-offs measured
// with sc1 loads"; checked here by the bit-reproducibility-under-load test).  Every spin is bounded: a block that
// gives up sets the timeout word and leaves, and the host reports it (rela_r2d2_learner_check) and falls back to
// per-step launches.  History (r2-r3): a one-block-per-4-units kernel over all eight XCDs (14 us per step, 5-6 of
// them in a 128-arrival grid barrier and its acquire) was replaced by these chains (6.4 us per step) and removed in r4.
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
constexpr int kRecThreads = 512;
constexpr unsigned kRecSpinLimit = 1u << 22;
struct RecNet {
  float* gx;          // [T][Bn][2048] pre-activations of the input half (+ bias); saved steps get the activated gates
  const float* whhT;  // [512][2048]
  float *H, *C;       // [(T + 1)][Bn][512], slot 0 = initial state
  unsigned* bar;      // (unused by the chains: their counters are ChainArgs2::bar)
  int save;
};
struct ChainArgs2 {
  RecNet net[2];
  const uint8_t* term;
  unsigned* tmo;
  unsigned* bar;  // [8 chains][Tpad] arrival counters
  int T, Tpad, Bn, burn;
};
constexpr int kChainBlocks = 32, kChainUnits = kHid / kChainBlocks;  // 16 units per block
static_assert(kChainUnits == 16, "tile shapes below");
template <bool SC1>
__global__ __launch_bounds__(kRecThreads) void lstm_rec_chain(ChainArgs2 a) {
  __shared__ float red[8][16][65];
  __shared__ int alive;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int chain = blockIdx.x & 7, j = blockIdx.x >> 3;  // chain = (net, row tile); j = unit block
  const RecNet nt = a.net[chain >> 2];
  const int row0 = (chain & 3) * 16;
  if (row0 >= a.Bn) return;  // (the whole chain leaves: its counters are its own)
  const int nrow = min(16, a.Bn - row0);
  // B fragments: column tile c = gate c, column li = unit 16 j + li; wave w: k in [64 w, 64 w + 64)
  float bfr[4][16];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) bfr[c][ks] = nt.whhT[(size_t)(wave * 64 + 16 * g + ks) * kGates + c * kHid + 16 * j + li];
  const size_t blk = (size_t)a.Bn * kHid;
  const __amdgpu_buffer_rsrc_t rsH = sc1_rsrc(nt.H, (size_t)(a.T + 1) * blk * 4);
  // the cell's thread: unit u = tid & 15, row r = tid >> 4 (threads 0..255)
  const int cu = tid & 15, cr = tid >> 4;
  const bool cell = tid < 256 && cr < nrow;
  const int crow = row0 + cr;
  float c_reg = 0.f;
  if (cell) c_reg = nt.C[(size_t)crow * kHid + 16 * j + cu];  // slot 0 = the initial state
  float gpre[4];
  auto gx_fetch = [&](int t) {
    const float* grow = nt.gx + ((size_t)t * a.Bn + crow) * kGates + 16 * j + cu;
#pragma unroll
    for (int q = 0; q < 4; ++q) gpre[q] = grow[q * kHid];
  };
  if (cell) gx_fetch(0);
  unsigned* bar = a.bar + (size_t)chain * a.Tpad;
  for (int t = 0; t < a.T; ++t) {
    // h_{t-1} of the chain's rows (rows past the batch re-read its last row; their results are never stored)
    float4 hv[4];
    {
      const int arow = row0 + min(li, nrow - 1);
      const size_t hoff = (size_t)t * blk + (size_t)arow * kHid + wave * 64 + 16 * g;
#pragma unroll
      for (int q = 0; q < 4; ++q) hv[q] = ld4_shared<SC1>(rsH, nt.H, hoff + 4 * q);
    }
    const float av[16] = {hv[0].x, hv[0].y, hv[0].z, hv[0].w, hv[1].x, hv[1].y, hv[1].z, hv[1].w,
                          hv[2].x, hv[2].y, hv[2].z, hv[2].w, hv[3].x, hv[3].y, hv[3].z, hv[3].w};
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bfr[c][ks], acc[c], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][4 * g + r][c * 16 + li] = acc[c][r];
    __syncthreads();
    if (cell) {
      float pre[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pre[q] = gpre[q];
#pragma unroll
        for (int w = 0; w < 8; ++w) pre[q] += red[w][cr][q * 16 + cu];
      }
      // the state that ENTERS the first training step is zeroed where the burn-in was a dummy (r2d2.py:149-154)
      const bool zero = a.burn > 0 && t + 1 == a.burn && a.term[(size_t)(a.burn - 1) * a.Bn + crow] != 0;
      const float gi = sigm(pre[0]), gf = sigm(pre[1]), gg = tanhf(pre[2]), go = sigm(pre[3]);
      float c = gf * c_reg + gi * gg;
      float h = go * tanhf(c);
      if (zero) c = 0.f, h = 0.f;
      c_reg = c;
      const size_t o = (size_t)(t + 1) * blk + (size_t)crow * kHid + 16 * j + cu;
      __hip_atomic_store((gu32*)(nt.H + o), __float_as_uint(h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
      nt.C[o] = c;
      if (nt.save && t >= a.burn) {
        float* grow = nt.gx + ((size_t)t * a.Bn + crow) * kGates + 16 * j + cu;
        grow[0] = gi, grow[kHid] = gf, grow[2 * kHid] = gg, grow[3 * kHid] = go;
      }
    }
    if (t + 1 == a.T) break;
    // grid barrier of the chain: every storing wave drains its stores, then ONE lane signals and polls
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (cell) gx_fetch(t + 1);  // in flight across the wait
    if (tid == 0) {
      gu32* cnt = (gu32*)(bar + t);
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool ok = true;
      for (unsigned spins = 0; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)kChainBlocks;) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kRecSpinLimit || __hip_atomic_load((gu32*)a.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store((gu32*)a.tmo, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = false;
          break;
        }
      }
      if constexpr (!SC1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      alive = ok ? 1 : 0;
    }
    __syncthreads();
    if (!alive) return;
  }
}

// ---- persistent BPTT: XCD-local chains (batches of up to 128 rows) ---------------------------------------------
// The seq_len + n backward steps of the online net in ONE launch (2 launches per step before r2: a split-K GEMM for
// dh = dgates_{t+1} x W_hh and the cell kernel; launch gaps alone cost ~1 ms per learner step).  chain = tile of 16
// batch rows, run by the 32 blocks whose ids are equal mod 8 (one XCD, see lstm_rec_chain); block j of a chain owns
// hidden units 16j .. 16j+15: its [2048 x 16] slice of W_hh stays in registers (wave w: gate columns [256w, 256w +
// 256), one f32 MFMA B-fragment register per k-step), every step the eight waves multiply their slices of
// dgates_{t+1} -- plain loads behind ONE agent-scope acquire per step: sc1 loads are SLOWER here (0.82 -> 0.96-1.01
// ms), the 128 KB of gate gradients a block reads per step then bypass L1 -- into partial [16 x 16] tiles that meet in
// LDS, a thread per (row, four units) runs the cell backward with the recurrent cell-state gradient in registers and
// writes the gate gradients of the block's 64 gate columns WRITE-THROUGH, in place of the activated gates.  Chain
// barrier per step as in lstm_rec_chain (bounded spins, timeout word).  (Requesting the cell's own inputs of step
// t - 1 before the wait at the barrier, as the forward kernel does with its x-part, made this kernel SLOWER, 0.79 ->
// 1.05 ms; the cause is not established.)
constexpr int kBpttBlocks = kHid / 16;

struct BpttArgs {
  float* ga;           // [Tt][Bn][2048] activated gates of the training steps -> gate gradients, in place
  const float* d_o;    // [Tt][Bn][512] gradient from the heads
  const float* whh;    // [2048][512] weight_hh_l0 (gate-major rows)
  const float* C;      // [(T + 1)][Bn][512] cell states, slot 0 = initial
  float* dc_rec;       // [Bn][512] running dL/dc (zeroed before the launch)
  unsigned* bar;       // [chains][Tpad] arrival counters
  unsigned* tmo;
  int Tt, Bn, burn;
};

template <bool SC1>
__global__ __launch_bounds__(kRecThreads) void lstm_bptt_chain(BpttArgs a, int Tpad) {
  __shared__ float red[8][16][17];
  __shared__ int alive;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4, chain = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int row0 = chain * 16;
  if (row0 >= a.Bn) return;  // (the whole chain leaves)
  float bfr[64];
#pragma unroll
  for (int ks = 0; ks < 64; ++ks) bfr[ks] = a.whh[(size_t)(256 * wave + 64 * g + ks) * kHid + 16 * j + li];
  const size_t blk = (size_t)a.Bn * kHid;
  const __amdgpu_buffer_rsrc_t rsG = sc1_rsrc(a.ga, (size_t)a.Tt * a.Bn * kGates * 4);
  // cell backward for (row, units 16 j + 4 q .. + 3): threads 0 .. 63, as in the kernel above (16-byte loads, 8-byte
  // write-through stores: one thread per (row, unit) with 4-byte stores was 20 % slower)
  const int r = tid >> 2, q = tid & 3, row = row0 + r;
  const bool cell = tid < 4 * 16 && row < a.Bn;
  const size_t u0 = (size_t)row * kHid + 16 * j + 4 * q;
  float dcr[4] = {0.f, 0.f, 0.f, 0.f};
  if (cell) {
    const float4 d4 = *reinterpret_cast<const float4*>(a.dc_rec + u0);
    dcr[0] = d4.x, dcr[1] = d4.y, dcr[2] = d4.z, dcr[3] = d4.w;
  }
  unsigned* bar = a.bar + (size_t)chain * Tpad;
  for (int t = a.Tt - 1; t >= 0; --t) {
    const bool rec = t + 1 < a.Tt;  // the newest step has no recurrent term
    float* ga_t = a.ga + (size_t)t * a.Bn * kGates;
    if (rec) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      const int arow = min(row0 + li, a.Bn - 1);  // (rows past the batch repeat the last one, unread)
      const size_t doff = (size_t)(t + 1) * a.Bn * kGates + (size_t)arow * kGates + 256 * wave + 64 * g;
      float4 v[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) v[c] = ld4_shared<SC1>(rsG, a.ga, doff + 4 * c);  // all in flight before the first MFMA
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[c].x, bfr[4 * c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[c].y, bfr[4 * c + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[c].z, bfr[4 * c + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[c].w, bfr[4 * c + 3], acc, 0, 0, 0);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) red[wave][4 * g + rr][li] = acc[rr];
    }
    __syncthreads();
    if (cell) {
      const int gs = a.burn + t;
      const float4 d_o4 = *reinterpret_cast<const float4*>(a.d_o + (size_t)t * blk + u0);
      float dh[4] = {d_o4.x, d_o4.y, d_o4.z, d_o4.w};
      if (rec) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int w = 0; w < 8; ++w) dh[u] += red[w][r][4 * q + u];
      }
      float* grow = ga_t + (size_t)row * kGates + 16 * j + 4 * q;
      const float4 gi4 = *reinterpret_cast<const float4*>(grow), gf4 = *reinterpret_cast<const float4*>(grow + kHid);
      const float4 gg4 = *reinterpret_cast<const float4*>(grow + 2 * kHid), go4 = *reinterpret_cast<const float4*>(grow + 3 * kHid);
      const float4 cn4 = *reinterpret_cast<const float4*>(a.C + (size_t)(gs + 1) * blk + u0);
      const float4 cp4 = *reinterpret_cast<const float4*>(a.C + (size_t)gs * blk + u0);
      const float gi[4] = {gi4.x, gi4.y, gi4.z, gi4.w}, gf[4] = {gf4.x, gf4.y, gf4.z, gf4.w};
      const float gg[4] = {gg4.x, gg4.y, gg4.z, gg4.w}, go[4] = {go4.x, go4.y, go4.z, go4.w};
      const float cn[4] = {cn4.x, cn4.y, cn4.z, cn4.w}, cp[4] = {cp4.x, cp4.y, cp4.z, cp4.w};
      float di[4], df[4], dgg[4], dgo[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float tc = tanhf(cn[u]);
        const float dc = dcr[u] + dh[u] * go[u] * (1.0f - tc * tc);
        di[u] = dc * gg[u] * gi[u] * (1.0f - gi[u]);
        df[u] = dc * cp[u] * gf[u] * (1.0f - gf[u]);
        dgg[u] = dc * gi[u] * (1.0f - gg[u] * gg[u]);
        dgo[u] = dh[u] * tc * go[u] * (1.0f - go[u]);
        dcr[u] = dc * gf[u];
      }
      auto store_wt = [&](float* dst, const float* v) {  // write-through: read by the chain's other blocks in the next step
        gu64* p = (gu64*)dst;
        __hip_atomic_store(p, ((unsigned long long)__float_as_uint(v[1]) << 32) | __float_as_uint(v[0]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, ((unsigned long long)__float_as_uint(v[3]) << 32) | __float_as_uint(v[2]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      store_wt(grow, di);
      store_wt(grow + kHid, df);
      store_wt(grow + 2 * kHid, dgg);
      store_wt(grow + 3 * kHid, dgo);
    }
    if (t == 0) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      gu32* cnt = (gu32*)(bar + t);
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool ok = true;
      for (unsigned spins = 0; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)kBpttBlocks;) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kRecSpinLimit || __hip_atomic_load((gu32*)a.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store((gu32*)a.tmo, (unsigned)(1000 + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = false;
          break;
        }
      }
      if constexpr (!SC1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      alive = ok ? 1 : 0;
    }
    __syncthreads();
    if (!alive) return;
  }
  if (cell) *reinterpret_cast<float4*>(a.dc_rec + u0) = make_float4(dcr[0], dcr[1], dcr[2], dcr[3]);  // (as the kernel above leaves it)
}

// hid *= 1 - terminal[burn_in - 1]   (r2d2.py:149-154)
__global__ void zero_hidden_where_terminal(const uint8_t* __restrict__ term, int Bn, float* __restrict__ h,
                                           float* __restrict__ c) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Bn * kHid) return;
  if (term[idx / kHid]) h[idx] = 0.f, c[idx] = 0.f;
}

// BPTT of one step.  ga: activated gates of the step (overwritten by the pre-activation gradients),
// d_o: gradient from the heads, dh partials: dgates_{t+1} x W_hh (nsplit = 0 at the last step).
__global__ void lstm_cell_bwd(float* __restrict__ ga, const float* __restrict__ d_o, const float* __restrict__ dh_part,
                              int nsplit, int Bn, const float* __restrict__ c_now, const float* __restrict__ c_prev,
                              float* __restrict__ dc_rec) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Bn * kHid) return;
  const int b = idx / kHid, u = idx - b * kHid;
  float dh = d_o[idx];
  for (int z = 0; z < nsplit; ++z) dh += dh_part[(size_t)z * Bn * kHid + idx];
  float* row = ga + (size_t)b * kGates + u;
  const float gi = row[0], gf = row[kHid], gg = row[2 * kHid], go = row[3 * kHid];
  const float tc = tanhf(c_now[idx]);
  const float dc = dc_rec[idx] + dh * go * (1.0f - tc * tc);
  row[0] = dc * gg * gi * (1.0f - gi);
  row[kHid] = dc * c_prev[idx] * gf * (1.0f - gf);
  row[2 * kHid] = dc * gi * (1.0f - gg * gg);
  row[3 * kHid] = dh * tc * go * (1.0f - go);
  dc_rec[idx] = dc * gf;
}

// q.min() over the whole [Tt, B, A] tensor (net.py:160)
__global__ __launch_bounds__(1024) void q_min_all(const float* __restrict__ q, int64_t n, float* __restrict__ out) {
  __shared__ float red[1024];
  float m = INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += 1024) m = fminf(m, q[i]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fminf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// per training row r = (t, b): greedy action of the online net, Q_online[a_r], Q_target[greedy_r]
__global__ void seq_select(const float* __restrict__ q_on, const float* __restrict__ q_tg,
                           const float* __restrict__ legal, const int64_t* __restrict__ act,
                           const float* __restrict__ qmin, int rows, int A, float* __restrict__ qa_on,
                           float* __restrict__ qa_tg) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float mn = qmin[0];
  const float* qo = q_on + (size_t)r * A;
  const float* lg = legal + (size_t)r * A;
  int best = 0;
  float bv = -INFINITY;
  for (int j = 0; j < A; ++j) {
    const float v = (1.0f + qo[j] - mn) * lg[j];
    if (v > bv) bv = v, best = j;  // first maximum, as torch.argmax
  }
  qa_on[r] = qo[(int)act[r]];
  qa_tg[r] = q_tg[(size_t)r * A + best];
}

// per sequence b: err_i = (r_i + boot_i * gamma^n * Qt[i+n] - Qo[i]) * (1 - pad_i), Huber sum, eta-mixed
// priority, and d(mean_b(loss_b * w_b)) / dQo[i]; one thread per sequence (B <= 1024), one block.
__global__ __launch_bounds__(1024) void seq_td_loss(const float* __restrict__ qa_on, const float* __restrict__ qa_tg,
                                                    const float* __restrict__ reward, const float* __restrict__ boot,
                                                    const float* __restrict__ seq_len, const float* __restrict__ w,
                                                    int Bn, int seq, int burn, int nstep, float gamma_n, float eta,
                                                    float one_minus_eta, float* __restrict__ dqa,
                                                    float* __restrict__ prio, float* __restrict__ loss_seq,
                                                    float* __restrict__ loss_out) {
  __shared__ float red[1024];
  const int b = threadIdx.x;
  float lw = 0.f;
  if (b < Bn) {
    const float len = seq_len[b];
    const float inv_b = 1.0f / (float)Bn;
    float lsum = 0.f, psum = 0.f, pmax = 0.f;  // |err| * mask >= 0, so the maximum over the row starts at 0
    for (int i = 0; i < seq + nstep; ++i) {
      const size_t r = (size_t)i * Bn + b;  // training rows are time-major
      float g = 0.f;
      if (i < seq) {
        const float target = reward[r] + boot[r] * (gamma_n * qa_tg[(size_t)(i + nstep) * Bn + b]);
        const bool pad = (float)i >= len - (float)burn;  // should_padding :175
        const float e = pad ? 0.f : target - qa_on[r];
        const float ae = fabsf(e);
        lsum += ae < 1.0f ? 0.5f * e * e : ae - 0.5f;
        const float pm = ((float)i < len) ? ae : 0.f;  // aggregate_priority mask :113-115
        psum += pm;
        pmax = fmaxf(pmax, pm);
        // err = target - Qo:  d mean(loss * w) / d Qo[i] = -w * clamp(err, -1, 1) / B
        g = pad ? 0.f : -(w[b] * fminf(fmaxf(e, -1.0f), 1.0f)) * inv_b;
      }
      dqa[r] = g;
    }
    loss_seq[b] = lsum;
    prio[b] = eta * pmax + one_minus_eta * (psum / (len - (float)burn));
    lw = lsum * w[b];
  }
  red[threadIdx.x] = lw;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = red[0] / (float)Bn;
}

// gradient through the dueling head  q = v + a*legal - mean_A(a*legal)  (net.py:93-99), per training row
__global__ void seq_head_grad(const float* __restrict__ dqa, const int64_t* __restrict__ act,
                              const float* __restrict__ legal, int rows, int A, float* __restrict__ d_ha) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * 32) return;
  const int r = idx >> 5, k = idx & 31;
  const float g = dqa[r];
  float v = 0.f;
  if (k < A) v = legal[(size_t)r * A + k] * (g * ((k == (int)act[r] ? 1.0f : 0.0f) - 1.0f / (float)A));
  if (k == 31) v = g;
  d_ha[idx] = v;
}

}  // namespace
}  // namespace rela_amd

// flat parameter layout: rela_lstmnet_params order, every segment padded to 4 floats
struct rela_r2d2_learner {
  int device = 0;
  int A = 0, Bmax = 0, seq = 0, burn = 0, n = 0, T = 0;
  float gamma_n = 0.f, eta = 0.f, one_minus_eta = 0.f;
  OptimState opt;
  int64_t off[15] = {0};  // segment offsets, off[14] = total
  float *P = nullptr, *PT = nullptr, *G = nullptr, *S1 = nullptr, *S2 = nullptr;
  rela_lstmnet *online = nullptr, *target = nullptr;
  float *w2p = nullptr, *w3p = nullptr;                               // conv dgrad operand copies (online)
  float *wihT[2] = {nullptr, nullptr}, *whhT[2] = {nullptr, nullptr}, *bsum[2] = {nullptr, nullptr};  // [online, target]
  float* wihp = nullptr;                                              // online W_ih in k order (dgrad)
  float *a1 = nullptr, *a2 = nullptr, *a3 = nullptr;                  // trunk activations of T*B frames
  float* gxs[2] = {nullptr, nullptr};  // per net [T*B][2048]: GX; the online one -> activated gates -> dgates
  float*& gx = gxs[0];
  float *Hs[2] = {nullptr, nullptr}, *Cs[2] = {nullptr, nullptr};     // [(T+1)*B][512] per net
  float* rec_part = nullptr;                                          // split-K partials of the recurrent GEMMs
  unsigned* rec_bar = nullptr;                                        // [0] timeout word, [4 ..] per-step arrival counters
  unsigned* rec_chain_bar = nullptr;                                  // lstm_rec_chain: [8 chains][Tpad]
  bool rec_chains_fit = true;                                         // its 256 blocks are resident at once (checked at create)
  bool rec_persist = true;
  // 1: the target net's conv trunk (no gradient, activations never read back) and the three large GEMMs of the LSTM's
  //    input side (gate GEMM of both nets, its data and weight gradients) on split-bf16 MFMA (gemm_bf16s.h)
  int precision = 0;
  // rec64 operands of those GEMMs (hi | lo bf16 records, 4 bytes per element)
  uint8_t* wrec[2] = {nullptr, nullptr};  // W_ih [2048][49 chunks], k = pos * 64 + c       (gate GEMM, per net)
  uint8_t* wTrec = nullptr;               // W_ih^T [3136][32 chunks], online                 (data gradient)
  uint8_t *arec = nullptr, *trec = nullptr;  // activations / gate gradients by rows; transposed operands of dW_ih
  uint64_t wver[2] = {1, 1}, rec_ver[2] = {0, 0};  // weight version (repack) / version the records were made from
  // f32x3 (r5): the trunks on split3 records and the x part of the gates as a three-part GEMM (csrc/gemm_s3.h)
  uint8_t* s3rec = nullptr;          // rowsAll x (a2 + a3 records)
  void* wx3[2] = {nullptr, nullptr};  // W_ih of either net as three-part fragments (gate columns as stored)
  uint64_t x3_ver[2] = {0, 0};
  float *ha = nullptr, *q_on = nullptr, *q_tg = nullptr;              // heads of the training rows
  float *qmin = nullptr, *qa_on = nullptr, *qa_tg = nullptr, *dqa = nullptr, *d_ha = nullptr, *d_o = nullptr;
  float *dc_rec = nullptr;
  float *d_a3 = nullptr, *d_a2 = nullptr, *d_a1 = nullptr, *col = nullptr, *part = nullptr, *cpart = nullptr,
        *s32 = nullptr;
  double* npart = nullptr;
  float *norm = nullptr, *loss = nullptr, *loss_seq = nullptr;
  bool loaded = false;
  // batch of the last rela_r2d2_learner_loss, until rela_r2d2_learner_grad consumes it
  int pend_B = 0;
  const uint8_t* pend_obs = nullptr;
};

namespace {
const int64_t* seg_counts(int A, int64_t cnt[14]) {
  const int64_t c[14] = {32 * 256, 32, 64 * 512, 64, 64 * 576, 64, (int64_t)kGates * kFeat, (int64_t)kGates * kHid,
                         kGates, kGates, 512, 1, (int64_t)A * 512, A};
  for (int i = 0; i < 14; ++i) cnt[i] = c[i];
  return cnt;
}

rela_lstmnet_params lparams_at(const rela_r2d2_learner* l, float* base) {
  rela_lstmnet_params p;
  const float** f = reinterpret_cast<const float**>(&p);
  for (int i = 0; i < 14; ++i) f[i] = base + l->off[i];
  return p;
}

int repack_r2d2(rela_r2d2_learner* l, bool online, bool target, hipStream_t s) {
  static const hipError_t pack_attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_lstm_learner),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         (int)(kPackRows * kFeat * sizeof(float)));
  RELA_HIP(pack_attr);
  if (online) {
    const rela_lstmnet_params p = lparams_at(l, l->P);
    int rc = rela_lstmnet_load(l->online, &p, 1, s);
    if (rc != RELA_OK) return rc;
    hipLaunchKernelGGL(permute_weights, dim3(ceil_div(64 * 512, 256)), dim3(256), 0, s, kPermConv2, p.conv2_w, l->w2p,
                       64 * 512);
    hipLaunchKernelGGL(permute_weights, dim3(ceil_div(64 * 576, 256)), dim3(256), 0, s, kPermConv3, p.conv3_w, l->w3p,
                       64 * 576);
    hipLaunchKernelGGL(pack_lstm_learner, dim3(kGates / kPackRows), dim3(256), kPackRows * kFeat * sizeof(float), s, p.w_ih, p.w_hh, p.b_ih, p.b_hh,
                       l->wihT[0], l->wihp, l->whhT[0], l->bsum[0]);
    RELA_LAUNCH_CHECK();
    l->wver[0] += 1;
  }
  if (target) {
    const rela_lstmnet_params p = lparams_at(l, l->PT);
    int rc = rela_lstmnet_load(l->target, &p, 1, s);
    if (rc != RELA_OK) return rc;
    hipLaunchKernelGGL(pack_lstm_learner, dim3(kGates / kPackRows), dim3(256), kPackRows * kFeat * sizeof(float), s, p.w_ih, p.w_hh, p.b_ih, p.b_hh,
                       l->wihT[1], (float*)nullptr, l->whhT[1], l->bsum[1]);
    RELA_LAUNCH_CHECK();
    l->wver[1] += 1;
  }
  return RELA_OK;
}

// rec64 copies of W_ih for the split-bf16 GEMMs, made from the packed f32 copies when those changed
int refresh_weight_records(rela_r2d2_learner* l, int which, hipStream_t s) {
  if (l->rec_ver[which] == l->wver[which]) return RELA_OK;
  using namespace gemm16;
  ProfScope prof("learner_lstm_split_w", s);
  if (which == 0) {
    // wihp [2048][3136] (k = pos * 64 + c) row-wise; wihT [3136][2048] row-wise = W_ih^T
    hipLaunchKernelGGL(split_rows_rec64, dim3(ceil_div((int64_t)kGates * kFeat / 8, 256)), dim3(256), 0, s,
                       (const float*)l->wihp, (int64_t)kGates, kFeat, l->wrec[0]);
    hipLaunchKernelGGL(split_rows_rec64, dim3(ceil_div((int64_t)kGates * kFeat / 8, 256)), dim3(256), 0, s,
                       (const float*)l->wihT[0], (int64_t)kFeat, kGates, l->wTrec);
  } else {
    // the target net keeps only wihT [3136][2048]: its columns are W_ih's rows
    hipLaunchKernelGGL(split_cols_rec64, dim3(kGates / 64, kFeat / 64), dim3(256), 0, s, (const float*)l->wihT[1],
                       (int64_t)kFeat, kGates, 0, l->wrec[1]);
  }
  RELA_LAUNCH_CHECK();
  l->rec_ver[which] = l->wver[which];
  return RELA_OK;
}

const char* const kTrunkNames[3] = {"learner_fwd_conv1", "learner_fwd_conv2", "learner_fwd_conv3"};

// Forward of both nets over the whole batch, in three stages: (1) per net the conv trunk over all T*B frames and the
// input half of the gates (target first: the online pass leaves its a1 / a2 / a3 for the backward pass); (2) the T
// recurrent steps of BOTH nets in one persistent launch (two independent latency chains side by side); (3) the heads
// over the training steps.  Leaves H / C of every step in Hs[w] / Cs[w], the activated gates of the online net's
// training steps in gxs[0] and the dueling Q of the training rows in q_on / q_tg.
int forward_pre(rela_r2d2_learner* l, int which, int Bn, const uint8_t* obs, const float* h0, const float* c0,
                hipStream_t s) {
  const rela_lstmnet* net = which == 0 ? l->online : l->target;
  const int rowsAll = l->T * Bn;
  // bf16x2 (r3): BOTH trunks on split-bf16 MFMA.  The online pass leaves a1 / a2 / a3 of the training frames for the
  // backward kernels: its conv1 records are copied out (they never leave LDS otherwise), a3's records feed the gate GEMM
  // directly, then the training rows are turned back into f32 in place (below) -- the backward differentiates the
  // forward that really ran, ReLU pattern included.  RELA_R2D2_ONLINE_F32=1 keeps the online trunk in f32 (r2).
  constexpr bool online_f32 = false;  // (r2's f32 online trunk in bf16x2 mode: an A/B switch until r4)
  const bool fast_target = which == 1 && l->precision == 1;
  const bool fast_online = which == 0 && l->precision == 1 && !online_f32 && rowsAll >= 128;
  bool a3_records = false;  // a fast trunk hands a3 over as the split records the gate GEMM reads
  int rc;
  if (fast_online) {
    rc = lstmnet_trunk_records(net, rowsAll, obs, l->a1, l->a2, l->a3, l->burn * Bn, s, kTrunkNames);
    a3_records = true;
  } else {
    // (f32x3: on split3 records; only the online pass leaves a1 / a2 / a3 in f32, for the backward kernels)
    const bool x3 = l->precision == 2 && l->s3rec != nullptr;
    rc = lstmnet_trunk(net, rowsAll, obs, l->a1, l->a2, l->a3, s, kTrunkNames, fast_target,
                       (fast_target || x3) ? &a3_records : nullptr, x3 ? l->s3rec : nullptr, which == 0);
  }
  if (rc != RELA_OK) return rc;
  if (l->precision == 1) {
    using namespace gemm16;
    rc = refresh_weight_records(l, which, s);
    if (rc != RELA_OK) return rc;
    const uint8_t* arec = reinterpret_cast<const uint8_t*>(l->a3);
    if (!a3_records) {
      ProfScope prof("learner_lstm_split_rows", s);
      hipLaunchKernelGGL(split_rows_rec64, dim3(ceil_div((int64_t)rowsAll * kFeat / 8, 256)), dim3(256), 0, s,
                         (const float*)l->a3, (int64_t)rowsAll, kFeat, l->arec);
      arec = l->arec;
    }
    rc = launch_rec64_nt(arec, l->wrec[which], rowsAll, kGates, kFeat / 64,
                         EpiBias{l->gxs[which], l->bsum[which], kGates}, s, "learner_lstm_gates_x");
    if (rc != RELA_OK) return rc;
    if (fast_online) {  // the training frames' activations back into f32 for the backward pass
      const size_t tr0 = (size_t)l->burn * Bn;
      rc = trunk_unsplit_rows(l->a1 + tr0 * kA1, l->a2 + tr0 * kA2, l->a3 + tr0 * kA3, rowsAll - (int)tr0, s);
      if (rc != RELA_OK) return rc;
    }
  } else if (a3_records) {
    // f32x3: W_ih in three bf16 parts (re-packed when the weights changed), the GEMM over a3's records
    if (l->x3_ver[which] != l->wver[which]) {
      ProfScope prof("learner_lstm_split_w", s);
      const rela_lstmnet_params p = lparams_at(l, which == 0 ? l->P : l->PT);
      rc = pack_gate_x3(p.w_ih, l->wx3[which], false, s);
      if (rc != RELA_OK) return rc;
      l->x3_ver[which] = l->wver[which];
    }
    ProfScope prof("learner_lstm_gates_x", s);
    rc = gate_x3_gemm(l->s3rec + (size_t)rowsAll * kRec2Bytes, l->wx3[which], l->bsum[which], l->gxs[which], rowsAll, s);
    if (rc != RELA_OK) return rc;
  } else {
    ProbGateX p{};
    p.M = rowsAll, p.N = kGates, p.K = kFeat;
    p.a3 = l->a3, p.wihT = l->wihT[which], p.bias = l->bsum[which], p.gx = l->gxs[which];
    launch_gemm<TileRows>(p, 1, s, "learner_lstm_gates_x");
  }
  const size_t blk = (size_t)Bn * kHid;
  RELA_HIP(dev_copy2(l->Hs[which], h0, blk * sizeof(float), l->Cs[which], c0, blk * sizeof(float), s));  // (one launch: common.h)
  return RELA_OK;
}

int forward_rec_steps(rela_r2d2_learner* l, int which, int Bn, const uint8_t* term, bool save, hipStream_t s) {
  const int T = l->T, burn = l->burn;
  float *H = l->Hs[which], *Cc = l->Cs[which], *gx = l->gxs[which];
  const size_t blk = (size_t)Bn * kHid;
  const int cell_grid = ceil_div((int64_t)Bn * kHid, 256);
  for (int t = 0; t < T; ++t) {
    if (t == burn && burn > 0) {  // state after the burn-in: zero it where the burn-in was a dummy
      hipLaunchKernelGGL(zero_hidden_where_terminal, dim3(cell_grid), dim3(256), 0, s, term + (size_t)(burn - 1) * Bn,
                         Bn, H + t * blk, Cc + t * blk);
    }
    ProbRecFwd p{};
    p.M = Bn, p.N = kGates, p.K = kHid;
    p.h = H + t * blk, p.whhT = l->whhT[which], p.part = l->rec_part;
    launch_gemm<TileRec>(p, kRecSplitF, s, "learner_lstm_rec_fwd");
    ProfScope prof("learner_lstm_cell", s);
    hipLaunchKernelGGL(lstm_cell_fwd, dim3(cell_grid), dim3(256), 0, s, gx + (size_t)t * Bn * kGates,
                       (const float*)l->rec_part, kRecSplitF, Bn, (const float*)(Cc + t * blk), Cc + (t + 1) * blk,
                       H + (t + 1) * blk, (save && t >= burn) ? 1 : 0);
  }
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

int forward_both(rela_r2d2_learner* l, int Bn, const uint8_t* obs, const float* legal_train, const uint8_t* term,
                 const float* h0, const float* c0, hipStream_t s) {
  const int T = l->T, burn = l->burn, Tt = T - burn;
  int rc = forward_pre(l, 1, Bn, obs, h0, c0, s);
  if (rc != RELA_OK) return rc;
  rc = forward_pre(l, 0, Bn, obs, h0, c0, s);
  if (rc != RELA_OK) return rc;
  if (l->rec_persist && l->rec_chains_fit && Bn <= 64) {
    // arrival counters of this launch: [8 chains][Tpad] (the timeout word rec_bar[0] is sticky until ..._check)
    const int Tpad = (T + 3) / 4 * 4;
    RELA_HIP(hipMemsetAsync(l->rec_chain_bar, 0, sizeof(unsigned) * (size_t)(8 * Tpad), s));
    ChainArgs2 ca{};
    for (int w = 0; w < 2; ++w) {
      ca.net[w].gx = l->gxs[w], ca.net[w].whhT = l->whhT[w], ca.net[w].H = l->Hs[w], ca.net[w].C = l->Cs[w];
      ca.net[w].bar = nullptr, ca.net[w].save = w == 0 ? 1 : 0;
    }
    ca.term = term, ca.tmo = l->rec_bar, ca.bar = l->rec_chain_bar, ca.T = T, ca.Tpad = Tpad, ca.Bn = Bn, ca.burn = burn;
    ProfScope prof("learner_lstm_rec_persist", s);
    hipLaunchKernelGGL(lstm_rec_chain<true>, dim3(8 * kChainBlocks), dim3(kRecThreads), 0, s, ca);
    RELA_LAUNCH_CHECK();
  } else {
    rc = forward_rec_steps(l, 1, Bn, term, false, s);
    if (rc != RELA_OK) return rc;
    rc = forward_rec_steps(l, 0, Bn, term, true, s);
    if (rc != RELA_OK) return rc;
  }
  // heads over the training steps: o_t = H[t+1], t in [burn, T)
  const size_t blk = (size_t)Bn * kHid;
  rc = lstmnet_heads(l->target, Tt * Bn, l->Hs[1] + (size_t)(burn + 1) * blk, legal_train, l->ha, l->q_tg, s,
                     "learner_fwd_heads");
  if (rc != RELA_OK) return rc;
  return lstmnet_heads(l->online, Tt * Bn, l->Hs[0] + (size_t)(burn + 1) * blk, legal_train, l->ha, l->q_on, s,
                       "learner_fwd_heads");
}
}  // namespace

extern "C" int rela_r2d2_learner_create(rela_r2d2_learner** out, int num_action, int max_batch, int multi_step,
                                        float gamma, int seq_len, int burn_in, double eta, int optimizer, float lr,
                                        float eps, float grad_clip, int device) {
  RELA_CHECK(out && num_action >= 1 && num_action <= 31 && max_batch >= 1 && max_batch <= 1024 && multi_step >= 1 &&
                 seq_len >= 1 && burn_in >= 0 && (optimizer == 0 || optimizer == 1),
             RELA_EINVAL, "rela_r2d2_learner_create: bad arguments (A=%d batch=%d n=%d seq=%d burn=%d optimizer=%d)",
             num_action, max_batch, multi_step, seq_len, burn_in, optimizer);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_last_error("rela_r2d2_learner_create: HIP device %d not available (%d visible); there is no CPU path", device,
                   ndev);
    return RELA_ENODEV;
  }
  DeviceGuard g(device);
  auto* l = new rela_r2d2_learner();
  l->device = device;
  l->A = num_action, l->Bmax = max_batch, l->seq = seq_len, l->burn = burn_in, l->n = multi_step;
  l->T = burn_in + seq_len + multi_step;
  l->gamma_n = (float)pow((double)gamma, (double)multi_step);  // r2d2.py:165
  l->eta = (float)eta;
  l->one_minus_eta = (float)(1.0 - eta);  // TorchScript folds (1.0 - self.eta) in double, r2d2.py:119
  l->opt.optimizer = optimizer, l->opt.lr = lr, l->opt.eps = eps, l->opt.clip = grad_clip;
  int64_t cnt[14];
  seg_counts(num_action, cnt);
  for (int i = 0; i < 14; ++i) l->off[i + 1] = l->off[i] + (cnt[i] + 3) / 4 * 4;
  const size_t nb = sizeof(float) * (size_t)l->off[14];
  const size_t B = (size_t)max_batch, A = (size_t)num_action, T = (size_t)l->T, Tt = T - burn_in;
  const size_t rowsAll = T * B, rowsTr = Tt * B;
  auto alloc = [&](float** p, size_t floats, bool zero) -> int {
    RELA_HIP(hipMalloc(p, sizeof(float) * floats));
    if (zero) RELA_HIP(hipMemset(*p, 0, sizeof(float) * floats));
    return RELA_OK;
  };
#define R2_ALLOC(ptr, floats, zero)               \
  do {                                            \
    int _rc = alloc(&(ptr), (floats), (zero));    \
    if (_rc != RELA_OK) return _rc;               \
  } while (0)
  R2_ALLOC(l->P, l->off[14], true);
  R2_ALLOC(l->PT, l->off[14], true);
  R2_ALLOC(l->G, l->off[14], true);
  R2_ALLOC(l->S1, l->off[14], true);
  R2_ALLOC(l->S2, l->off[14], true);
  (void)nb;
  int rc = rela_lstmnet_create(&l->online, num_action, device);
  if (rc != RELA_OK) return rc;
  rc = rela_lstmnet_create(&l->target, num_action, device);
  if (rc != RELA_OK) return rc;
  R2_ALLOC(l->w2p, 64 * 512, false);
  R2_ALLOC(l->w3p, 64 * 576, false);
  for (int w = 0; w < 2; ++w) {
    R2_ALLOC(l->wihT[w], (size_t)kGates * kFeat, false);
    R2_ALLOC(l->whhT[w], (size_t)kGates * kHid, false);
    R2_ALLOC(l->bsum[w], kGates, false);
    R2_ALLOC(l->Hs[w], (T + 1) * B * kHid, true);
    R2_ALLOC(l->Cs[w], (T + 1) * B * kHid, true);
  }
  R2_ALLOC(l->wihp, (size_t)kGates * kFeat, false);
  {
    const size_t chunks = (rowsAll + 63) / 64;
    float* tmp = nullptr;
    for (int w = 0; w < 2; ++w) {
      R2_ALLOC(tmp, (size_t)kGates * kFeat, false);
      l->wrec[w] = reinterpret_cast<uint8_t*>(tmp);
    }
    R2_ALLOC(tmp, (size_t)kGates * kFeat, false);
    l->wTrec = reinterpret_cast<uint8_t*>(tmp);
    R2_ALLOC(tmp, rowsAll * kFeat, false);
    l->arec = reinterpret_cast<uint8_t*>(tmp);
    R2_ALLOC(tmp, (size_t)(kGates + kFeat) * chunks * 64, false);  // [2048][chunks] followed by [3136][chunks]
    l->trec = reinterpret_cast<uint8_t*>(tmp);
  }
  {
    float* tmp = nullptr;
    R2_ALLOC(tmp, (rowsAll * (size_t)(kRec2Bytes + kRec3Bytes) + 3) / 4, false);
    l->s3rec = reinterpret_cast<uint8_t*>(tmp);
    for (int w = 0; w < 2; ++w) {
      R2_ALLOC(tmp, ((size_t)gate_x3_packed_bytes() + 3) / 4, false);
      l->wx3[w] = tmp;
    }
  }
  R2_ALLOC(l->a1, rowsAll * kA1, false);
  R2_ALLOC(l->a2, rowsAll * kA2, false);
  R2_ALLOC(l->a3, rowsAll * kA3, false);
  R2_ALLOC(l->gxs[0], rowsAll * kGates, false);
  R2_ALLOC(l->gxs[1], rowsAll * kGates, false);
  {
    const size_t f = (size_t)kRecSplitF * B * kGates, b = (size_t)kRecSplitB * B * kHid;
    R2_ALLOC(l->rec_part, f > b ? f : b, false);
  }
  RELA_HIP(hipMalloc(&l->rec_bar, sizeof(unsigned) * (size_t)(8 + 2 * ((T + 3) / 4 * 4))));
  RELA_HIP(hipMemset(l->rec_bar, 0, sizeof(unsigned) * (size_t)(8 + 2 * ((T + 3) / 4 * 4))));
  RELA_HIP(hipMalloc(&l->rec_chain_bar, sizeof(unsigned) * (size_t)(8 * ((T + 3) / 4 * 4))));
  l->rec_persist = !(getenv("RELA_R2D2_REC") && strcmp(getenv("RELA_R2D2_REC"), "steps") == 0);
  if (l->rec_persist) {
    // The persistent kernels spin on a grid barrier: every block must be resident at once.  A plain launch checks
    // nothing, so check here -- blocks per CU the occupancy query admits x the CUs this process sees (a CU-masked or
    // partitioned device reports fewer) against the grids, with the query's known over-report of one block per CU
    // taken off -- and fall back to the per-step launches (split-K GEMM + cell kernel) when they would not fit.
    int cus = 0;
    RELA_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    //  the instantiations that are really launched: lstm_rec_chain<true>, lstm_bptt_chain<false>
    int occ_c = 0, occ_d = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_c, lstm_rec_chain<true>, kRecThreads, 0) != hipSuccess) occ_c = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_d, lstm_bptt_chain<false>, kRecThreads, 0) != hipSuccess) occ_d = 0;
    // (the query's known over-report concerns kernels near an SGPR allocation step at several blocks per CU: take one
    // block per CU off from four up; these kernels use < 80 SGPRs and need two blocks per CU at most)
    auto room = [cus](int occ) { return (int64_t)cus * (occ >= 4 ? occ - 1 : occ); };
    if (room(occ_c) < 8 * kChainBlocks || room(occ_d) < 8 * kBpttBlocks) {
      fprintf(stderr, "rela_r2d2_learner_create: %d CUs x (%d, %d) resident blocks cannot hold the persistent recurrent "
                      "grids (%d, %d): using the per-step launches\n", cus, occ_c, occ_d, 8 * kChainBlocks, 8 * kBpttBlocks);
      l->rec_chains_fit = false;
    }
  }
  R2_ALLOC(l->ha, rowsTr * 32, false);
  R2_ALLOC(l->q_on, rowsTr * A, false);
  R2_ALLOC(l->q_tg, rowsTr * A, false);
  R2_ALLOC(l->qmin, 4, false);
  R2_ALLOC(l->qa_on, rowsTr, false);
  R2_ALLOC(l->qa_tg, rowsTr, false);
  R2_ALLOC(l->dqa, rowsTr, false);
  R2_ALLOC(l->d_ha, rowsTr * 32, false);
  R2_ALLOC(l->d_o, rowsTr * kHid, false);
  R2_ALLOC(l->dc_rec, B * kHid, false);
  R2_ALLOC(l->d_a3, rowsTr * kA3, false);
  R2_ALLOC(l->d_a2, rowsTr * kA2, false);
  R2_ALLOC(l->d_a1, rowsTr * kA1, false);
  R2_ALLOC(l->col, trunk_col_floats(rowsTr), false);
  R2_ALLOC(l->part, kTrunkPartFloats, false);
  R2_ALLOC(l->cpart, (size_t)kColsumBlocks * kGates, false);
  R2_ALLOC(l->s32, 32, false);
  R2_ALLOC(l->norm, 2, true);
  R2_ALLOC(l->loss, 1, true);
  R2_ALLOC(l->loss_seq, B, true);
#undef R2_ALLOC
  RELA_HIP(hipMalloc(&l->npart, sizeof(double) * kNormBlocks));
  *out = l;
  return RELA_OK;
}

extern "C" void rela_r2d2_learner_destroy(rela_r2d2_learner* l) {
  if (!l) return;
  DeviceGuard g(l->device);
  (void)hipDeviceSynchronize();
  void* ps[] = {l->P,      l->PT,     l->G,       l->S1,      l->S2,      l->w2p,     l->w3p,   l->wihT[0], l->wihT[1],
                l->whhT[0], l->whhT[1], l->bsum[0], l->bsum[1], l->Hs[0],   l->Hs[1],   l->Cs[0], l->Cs[1],   l->wihp,
                l->a1,    l->a2,      l->a3,      l->gx,      l->rec_part, l->ha,   l->q_on,    l->q_tg,
                l->qmin,   l->qa_on,  l->qa_tg,   l->dqa,     l->d_ha,    l->d_o,     l->dc_rec, l->d_a3,   l->d_a2,
                l->d_a1,   l->col,    l->part,    l->cpart,   l->s32,     l->npart,   l->norm,  l->loss,    l->loss_seq};
  for (void* p : ps) (void)hipFree(p);
  (void)hipFree(l->rec_bar);
  (void)hipFree(l->rec_chain_bar);
  (void)hipFree(l->gxs[1]);
  (void)hipFree(l->wrec[0]);
  (void)hipFree(l->wrec[1]);
  (void)hipFree(l->wTrec);
  (void)hipFree(l->arec);
  (void)hipFree(l->s3rec);
  (void)hipFree(l->wx3[0]);
  (void)hipFree(l->wx3[1]);
  (void)hipFree(l->trec);
  rela_lstmnet_destroy(l->online);
  rela_lstmnet_destroy(l->target);
  delete l;
}

extern "C" int rela_r2d2_learner_load(rela_r2d2_learner* l, const rela_lstmnet_params* online,
                                      const rela_lstmnet_params* target, int on_device, void* stream_) {
  RELA_CHECK(l && online, RELA_EINVAL, "rela_r2d2_learner_load: bad arguments");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(l->device);
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  const float* const* fo = reinterpret_cast<const float* const*>(online);
  const float* const* ft = reinterpret_cast<const float* const*>(target ? target : online);
  int64_t cnt[14];
  seg_counts(l->A, cnt);
  for (int i = 0; i < 14; ++i) {
    RELA_CHECK(fo[i] && ft[i], RELA_EINVAL, "rela_r2d2_learner_load: parameter %d is NULL", i);
    RELA_HIP(hipMemcpyAsync(l->P + l->off[i], fo[i], sizeof(float) * cnt[i], kind, s));
    RELA_HIP(hipMemcpyAsync(l->PT + l->off[i], ft[i], sizeof(float) * cnt[i], kind, s));
  }
  if (!on_device) RELA_HIP(hipStreamSynchronize(s));  // the host buffers may go away
  const size_t nb = sizeof(float) * (size_t)l->off[14];
  RELA_HIP(hipMemsetAsync(l->S1, 0, nb, s));
  RELA_HIP(hipMemsetAsync(l->S2, 0, nb, s));
  l->opt.adam_t = 0;
  int rc = repack_r2d2(l, true, true, s);
  if (rc != RELA_OK) return rc;
  l->loaded = true;
  return RELA_OK;
}

extern "C" int rela_r2d2_learner_sync_target(rela_r2d2_learner* l, void* stream_) {
  RELA_CHECK(l && l->loaded, RELA_ESTATE, "rela_r2d2_learner_sync_target: parameters were never loaded");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(l->device);
  RELA_HIP(hipMemcpyAsync(l->PT, l->P, sizeof(float) * (size_t)l->off[14], hipMemcpyDeviceToDevice, s));
  return repack_r2d2(l, false, true, s);
}

extern "C" int rela_r2d2_learner_params(rela_r2d2_learner* l, rela_lstmnet_params* online_out,
                                        rela_lstmnet_params* target_out) {
  RELA_CHECK(l, RELA_EINVAL, "rela_r2d2_learner_params: bad arguments");
  if (online_out) *online_out = lparams_at(l, l->P);
  if (target_out) *target_out = lparams_at(l, l->PT);
  return RELA_OK;
}

extern "C" int rela_r2d2_learner_grads(rela_r2d2_learner* l, rela_lstmnet_params* grads_out) {
  RELA_CHECK(l && grads_out, RELA_EINVAL, "rela_r2d2_learner_grads: bad arguments");
  *grads_out = lparams_at(l, l->G);
  return RELA_OK;
}

extern "C" int rela_r2d2_learner_flat(rela_r2d2_learner* l, float** params_dev, float** grads_dev, int64_t* count) {
  RELA_CHECK(l, RELA_EINVAL, "rela_r2d2_learner_flat: bad arguments");
  if (params_dev) *params_dev = l->P;
  if (grads_dev) *grads_dev = l->G;
  if (count) *count = l->off[14];
  return RELA_OK;
}

extern "C" const float* rela_r2d2_learner_stats_dev(const rela_r2d2_learner* l) { return l ? l->norm : nullptr; }

extern "C" int rela_r2d2_learner_set_precision(rela_r2d2_learner* l, int mode) {
  RELA_CHECK(l && mode >= 0 && mode <= 2, RELA_EINVAL, "rela_r2d2_learner_set_precision: mode must be 0, 1 or 2");
  l->precision = mode;
  // 2 (f32x3): conv2 / conv3 of both nets' trunks on the three-part bf16 kernels (csrc/gemm_f32emu.h, f32 accuracy);
  // everything else -- gate GEMMs, recurrences, backward -- as in mode 0.  (Mode 1 passes its choices per call.)
  int rc = rela_lstmnet_set_precision(l->online, mode == 2 ? 2 : 0);
  if (rc != RELA_OK) return rc;
  return rela_lstmnet_set_precision(l->target, mode == 2 ? 2 : 0);
}

extern "C" int rela_r2d2_learner_check(rela_r2d2_learner* l, void* stream_) {
  RELA_CHECK(l, RELA_EINVAL, "rela_r2d2_learner_check: null learner");
  DeviceGuard g(l->device);
  RELA_HIP(hipStreamSynchronize((hipStream_t)stream_));
  unsigned tmo = 0;
  RELA_HIP(hipMemcpy(&tmo, l->rec_bar, sizeof(unsigned), hipMemcpyDeviceToHost));
  if (tmo != 0) {
    RELA_HIP(hipMemset(l->rec_bar, 0, sizeof(unsigned)));
    // every apply() since the timeout was skipped on the device (clip_coef saw the word); from here on this learner
    // runs the recurrences as per-step launches, which need no co-residency
    l->rec_persist = false;
    set_last_error("rela_r2d2_learner_check: the grid barrier of a persistent recurrent kernel timed out at step %u: the "
                   "results of that call are invalid, no optimiser update was applied since, and this learner now uses "
                   "the per-step launches", tmo - 1);
    return RELA_ESTATE;
  }
  return RELA_OK;
}

extern "C" int rela_r2d2_learner_backward(rela_r2d2_learner* l, int batch, const void* const* rows_dev,
                                          const float* weight_dev, float* priority_dev, float* loss_dev,
                                          float* loss_seq_dev, void* stream_) {
  int rc = rela_r2d2_learner_loss(l, batch, rows_dev, weight_dev, priority_dev, loss_dev, loss_seq_dev, stream_);
  if (rc != RELA_OK) return rc;
  return rela_r2d2_learner_grad(l, stream_);
}

// The forward half of the step (both unrolls, sequence TD errors, aggregated priorities, loss, head gradient) and,
// below, the backward half (BPTT, weight and data gradients): as rela_apex_learner_loss / _grad -- the priorities are
// final after the first half, so update_priority and the next sample may be queued between the two; the batch's
// frames must stay untouched until the second half's work is done.
extern "C" int rela_r2d2_learner_loss(rela_r2d2_learner* l, int batch, const void* const* rows_dev,
                                      const float* weight_dev, float* priority_dev, float* loss_dev,
                                      float* loss_seq_dev, void* stream_) {
  RELA_CHECK(l && l->loaded, RELA_ESTATE, "rela_r2d2_learner_loss: parameters were never loaded");
  RELA_CHECK(batch >= 1 && batch <= l->Bmax && rows_dev && weight_dev && priority_dev, RELA_EINVAL,
             "rela_r2d2_learner_loss: bad arguments (batch %d, max %d)", batch, l->Bmax);
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(l->device);
  l->pend_B = 0;
  const int Bn = batch, A = l->A, T = l->T, burn = l->burn, Tt = T - burn, rowsTr = Tt * Bn;
  // RNNTransition batch, time-major (types.cc:140-182), in the order of the 10-field sequence schema
  const uint8_t* obs = static_cast<const uint8_t*>(rows_dev[0]);     // [T][B][4][84][84]
  const float* legal = static_cast<const float*>(rows_dev[2]);       // [T][B][A]
  const int64_t* act = static_cast<const int64_t*>(rows_dev[3]);     // [T][B]
  const float* reward = static_cast<const float*>(rows_dev[4]);      // [T][B]
  const uint8_t* term = static_cast<const uint8_t*>(rows_dev[5]);    // [T][B]
  const float* boot = static_cast<const float*>(rows_dev[6]);        // [T][B]
  const float* h0 = static_cast<const float*>(rows_dev[7]);          // [1][B][512]
  const float* c0 = static_cast<const float*>(rows_dev[8]);
  const float* seq_len = static_cast<const float*>(rows_dev[9]);     // [B]
  RELA_CHECK(obs && legal && act && reward && term && boot && h0 && c0 && seq_len, RELA_EINVAL,
             "rela_r2d2_learner_loss: a batch field is NULL");
  const size_t tr0 = (size_t)burn * Bn;  // first training row
  const float* legal_tr = legal + tr0 * A;
  const int64_t* act_tr = act + tr0;
  // target net first (no gradient, :158-159), then the online net, whose activations stay for the backward pass
  int rc = forward_both(l, Bn, obs, legal_tr, term, h0, c0, s);
  if (rc != RELA_OK) return rc;
  {
    ProfScope prof("learner_seq_td", s);
    hipLaunchKernelGGL(q_min_all, dim3(1), dim3(1024), 0, s, (const float*)l->q_on, (int64_t)rowsTr * A, l->qmin);
    hipLaunchKernelGGL(seq_select, dim3(ceil_div(rowsTr, 256)), dim3(256), 0, s, (const float*)l->q_on,
                       (const float*)l->q_tg, legal_tr, act_tr, (const float*)l->qmin, rowsTr, A, l->qa_on, l->qa_tg);
    hipLaunchKernelGGL(seq_td_loss, dim3(1), dim3(1024), 0, s, (const float*)l->qa_on, (const float*)l->qa_tg,
                       reward + tr0, boot + tr0, seq_len, weight_dev, Bn, l->seq, burn, l->n, l->gamma_n, l->eta,
                       l->one_minus_eta, l->dqa, priority_dev, l->loss_seq, l->loss);
    hipLaunchKernelGGL(seq_head_grad, dim3(ceil_div((int64_t)rowsTr * 32, 256)), dim3(256), 0, s, (const float*)l->dqa,
                       act_tr, legal_tr, rowsTr, A, l->d_ha);
  }
  if (loss_dev) RELA_HIP(hipMemcpyAsync(loss_dev, l->loss, sizeof(float), hipMemcpyDeviceToDevice, s));
  if (loss_seq_dev)
    RELA_HIP(hipMemcpyAsync(loss_seq_dev, l->loss_seq, sizeof(float) * Bn, hipMemcpyDeviceToDevice, s));
  RELA_LAUNCH_CHECK();
  l->pend_B = Bn;
  l->pend_obs = obs;
  return RELA_OK;
}

extern "C" int rela_r2d2_learner_grad(rela_r2d2_learner* l, void* stream_) {
  RELA_CHECK(l && l->loaded && l->pend_B > 0, RELA_ESTATE, "rela_r2d2_learner_grad: no rela_r2d2_learner_loss to differentiate");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(l->device);
  const int Bn = l->pend_B, A = l->A, T = l->T, burn = l->burn, Tt = T - burn, rowsTr = Tt * Bn;
  const uint8_t* obs = l->pend_obs;
  const size_t tr0 = (size_t)burn * Bn;  // first training row
  l->pend_B = 0;

  const rela_lstmnet_params P = lparams_at(l, l->P);
  float* Gm[14];  // gradient tensors in rela_lstmnet_params order
  for (int i = 0; i < 14; ++i) Gm[i] = l->G + l->off[i];
  const size_t blk = (size_t)Bn * kHid;
  float *H = l->Hs[0], *Cc = l->Cs[0];
  const float* o_tr = H + (size_t)(burn + 1) * blk;  // LSTM outputs of the training steps
  // heads: d_o, dW, db
  {
    ProbHeadDgradSeq p{};
    p.M = rowsTr, p.N = kHid, p.K = 32;
    p.d_ha = l->d_ha, p.a_w = P.a_w, p.v_w = P.v_w, p.d_o = l->d_o, p.A = A;
    launch_gemm<TileRows>(p, 1, s, "learner_dgrad_heads");
  }
  if (rowsTr >= 1024) {  // split-K over 32 slices: 256 blocks instead of 8 (0.13 ms on 8 CUs at 5,312 rows)
    constexpr int kSplitHeads = 32;
    static_assert((size_t)kSplitHeads * 32 * 512 <= kTrunkPartFloats, "part size");
    ProbHeadWgradPart p{};
    p.M = 32, p.N = kHid, p.K = rowsTr;
    p.d_ha = l->d_ha, p.h = o_tr, p.part = l->part;
    launch_gemm<TileW32r>(p, kSplitHeads, s, "learner_wgrad_heads");
    hipLaunchKernelGGL(head_wgrad_reduce, dim3(ceil_div(32 * 512, 256)), dim3(256), 0, s, (const float*)l->part,
                       kSplitHeads, A, Gm[12], Gm[10]);
  } else {
    ProbHeadWgrad p{};
    p.M = 32, p.N = kHid, p.K = rowsTr;
    p.d_ha = l->d_ha, p.h = o_tr, p.g_a_w = Gm[12], p.g_v_w = Gm[10], p.A = A;
    launch_gemm<TileW32r>(p, 1, s, "learner_wgrad_heads");
  }
  colsum_launch(l->d_ha, rowsTr, 32, l->cpart, l->s32, s);
  hipLaunchKernelGGL(head_bias_grad, dim3(1), dim3(32), 0, s, (const float*)l->s32, A, Gm[13], Gm[11]);
  // BPTT over the training steps, newest first; the activated gates in gx become the gate gradients in place
  RELA_HIP(hipMemsetAsync(l->dc_rec, 0, blk * sizeof(float), s));
  const int cell_grid = ceil_div((int64_t)Bn * kHid, 256);
  float* ga_tr = l->gx + tr0 * kGates;
  if (l->rec_persist && l->rec_chains_fit && Bn <= 128 && Tt <= l->T) {
    const int Tpad_c = (l->T + 3) / 4 * 4;
    RELA_HIP(hipMemsetAsync(l->rec_chain_bar, 0, sizeof(unsigned) * (size_t)(8 * Tpad_c), s));
    BpttArgs ba{};
    ba.ga = ga_tr, ba.d_o = l->d_o, ba.whh = P.w_hh, ba.C = Cc, ba.dc_rec = l->dc_rec;
    ba.bar = l->rec_chain_bar, ba.tmo = l->rec_bar, ba.Tt = Tt, ba.Bn = Bn, ba.burn = burn;
    ProfScope prof("learner_lstm_bptt_persist", s);
    hipLaunchKernelGGL(lstm_bptt_chain<false>, dim3(8 * kBpttBlocks), dim3(kRecThreads), 0, s, ba, Tpad_c);
  } else
  for (int t = Tt - 1; t >= 0; --t) {
    const int gs = burn + t;  // global step
    {
      ProfScope prof("learner_lstm_cell_bwd", s);
      hipLaunchKernelGGL(lstm_cell_bwd, dim3(cell_grid), dim3(256), 0, s, ga_tr + (size_t)t * Bn * kGates,
                         (const float*)(l->d_o + (size_t)t * blk), (const float*)l->rec_part,
                         t == Tt - 1 ? 0 : kRecSplitB, Bn, (const float*)(Cc + (size_t)(gs + 1) * blk),
                         (const float*)(Cc + (size_t)gs * blk), l->dc_rec);
    }
    if (t > 0) {  // the state entering the first training step comes from the burn-in: no gradient (:144-147)
      ProbRecBwd p{};
      p.M = Bn, p.N = kHid, p.K = kGates;
      p.dg = ga_tr + (size_t)t * Bn * kGates, p.whh = P.w_hh, p.part = l->rec_part;
      launch_gemm<TileRec>(p, kRecSplitB, s, "learner_lstm_rec_bwd");
    }
  }
  const float* DG = ga_tr;                          // [rowsTr][2048]
  const float* hprev = H + (size_t)burn * blk;      // h_{t-1} of training step t = H[burn + t]
  const float* a3_tr = l->a3 + tr0 * kA3;
  if (l->precision != 1) {
    ProbWhh p{};
    p.M = kGates, p.N = kHid, p.K = rowsTr;
    p.dg = DG, p.hprev = hprev, p.out = Gm[7];
    launch_gemm<TileWg>(p, 1, s, "learner_wgrad_lstm_hh");
  }
  if (l->precision == 1) {
    // dW_ih[g][c*49+pos] = sum_r dg[r][g] * a3[r][pos*64+c]: both operands transposed into rec64 rows of 64 r each
    // (a3's columns land in weight_ih_l0's own column order, so the result is stored row-major as it is)
    using namespace gemm16;
    const int chunks = ceil_div(rowsTr, 64);
    uint8_t* dgT = l->trec;
    uint8_t* a3T = l->trec + (size_t)kGates * chunks * REC;
    {
      ProfScope prof("learner_lstm_split_cols", s);
      hipLaunchKernelGGL(split_cols_rec64, dim3(kGates / 64, chunks), dim3(256), 0, s, DG, (int64_t)rowsTr, kGates, 0, dgT);
      hipLaunchKernelGGL(split_cols_rec64, dim3(kFeat / 64, chunks), dim3(256), 0, s, a3_tr, (int64_t)rowsTr, kFeat, 1, a3T);
    }
    int rc = launch_rec64_nt(dgT, a3T, kGates, kFeat, chunks, EpiPlain{Gm[6], kFeat}, s, "learner_wgrad_lstm_ih");
    if (rc != RELA_OK) return rc;
    // dW_hh[g][k] = sum_r dg[r][g] * hprev[r][k]: the same transposed gate gradients against hprev^T (which takes
    // the place of a3^T, consumed by the launch above)
    {
      ProfScope prof("learner_lstm_split_cols", s);
      hipLaunchKernelGGL(split_cols_rec64, dim3(kHid / 64, chunks), dim3(256), 0, s, hprev, (int64_t)rowsTr, kHid, 0, a3T);
    }
    rc = launch_rec64_nt(dgT, a3T, kGates, kHid, chunks, EpiPlain{Gm[7], kHid}, s, "learner_wgrad_lstm_hh");
    if (rc != RELA_OK) return rc;
  } else {
    ProbWih p{};
    p.M = kGates, p.N = kFeat, p.K = rowsTr;
    p.dg = DG, p.a3 = a3_tr, p.out = Gm[6];
    launch_gemm<TileWg>(p, 1, s, "learner_wgrad_lstm_ih");
  }
  colsum_launch(DG, rowsTr, kGates, l->cpart, Gm[8], s);  // bias_ih_l0 and bias_hh_l0 see the same gradient
  RELA_HIP(hipMemcpyAsync(Gm[9], Gm[8], sizeof(float) * kGates, hipMemcpyDeviceToDevice, s));
  if (l->precision == 1) {
    using namespace gemm16;
    {
      ProfScope prof("learner_lstm_split_rows", s);
      hipLaunchKernelGGL(split_rows_rec64, dim3(ceil_div((int64_t)rowsTr * kGates / 8, 256)), dim3(256), 0, s, DG,
                         (int64_t)rowsTr, kGates, l->arec);
    }
    int rc = launch_rec64_nt(l->arec, l->wTrec, rowsTr, kFeat, kGates / 64, EpiReluMask{l->d_a3, a3_tr, kFeat}, s,
                             "learner_dgrad_lstm_ih");
    if (rc != RELA_OK) return rc;
  } else {
    ProbIhDgrad p{};
    p.M = rowsTr, p.N = kFeat, p.K = kGates;
    p.dg = DG, p.wihp = l->wihp, p.a3 = a3_tr, p.d_a3 = l->d_a3;
    launch_gemm<TileRows>(p, 1, s, "learner_dgrad_lstm_ih");
  }
  {
    TrunkBwd t{};
    t.Bn = rowsTr, t.obs = obs + tr0 * 28224, t.a1 = l->a1 + tr0 * kA1, t.a2 = l->a2 + tr0 * kA2, t.d_a3 = l->d_a3;
    t.d_a2 = l->d_a2, t.d_a1 = l->d_a1, t.col = l->col, t.part = l->part, t.cpart = l->cpart;
    t.w2p = l->w2p, t.w3p = l->w3p;
    t.g_c1w = Gm[0], t.g_c1b = Gm[1], t.g_c2w = Gm[2], t.g_c2b = Gm[3], t.g_c3w = Gm[4], t.g_c3b = Gm[5];
    t.fast = l->precision == 1;
    trunk_backward(t, s);
  }
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

extern "C" int rela_r2d2_learner_apply(rela_r2d2_learner* l, void* stream_) {
  RELA_CHECK(l && l->loaded, RELA_ESTATE, "rela_r2d2_learner_apply: parameters were never loaded");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(l->device);
  // (a grid-barrier timeout of this step's persistent kernels leaves stale gradients: the update is skipped on the
  // device, rela_r2d2_learner_check reports it and switches to the per-step launches)
  optimizer_apply(l->opt, l->P, l->G, l->S1, l->S2, l->off[14], l->npart, l->norm, s, l->rec_bar);
  RELA_LAUNCH_CHECK();
  return repack_r2d2(l, true, false, s);
}
