// actor_r2d2.hip -- device-resident R2D2 actor shard (C ABI: rela_r2d2_actor_*).
//
// Restates what BasicThreadLoop (rela/thread_loop.h:74-105) drives through R2D2Actor
// (rela/r2d2_actor.h:189-353):
//   act        :221-249  historyHidden_.push_back(hidden_); TorchScript "act" (r2d2.py:58-73 ->
//                        net.py:110-124): conv trunk + one LSTM step, greedy over the raw advantages
//                        shifted by their batch minimum, eps-greedy; hidden_ <- new state
//   post_step  :252-302  push (r,t); zero the hidden state of finished envs (:258-267); n-step pop
//                        (dqn_actor.h:58-106); per-step priority (r2d2.py:76-100: online(s_t,h_t),
//                        online.act(s_t+n,h_t+n), target(s_t+n,h_t+n)); R2D2TransitionBuffer::push
//                        (:29-87); on canPop popTransition (:93-170) + aggregate_priority
//                        (r2d2.py:103-120) + replay add (:301)
// The integer bookkeeping of the sequence windows is r2d2_seq_core.h (host); this file owns the HBM
// side: per env one window of T = burn_in + seq_len + multi_step slots PER FIELD, laid out so a whole
// window is exactly one replay row of that field -- emitting a sequence is a plain row copy.
#include <atomic>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "prof.h"
#include "r2d2_seq_core.h"

using namespace rela_amd;

namespace {

constexpr int64_t kObs = 4 * 84 * 84;
constexpr int kHid = 512;
constexpr int kT = 256;

struct Windows {
  uint8_t* s;     // [R][T][28224]
  float* eps;     // [R][T]
  float* legal;   // [R][T][A]
  int64_t* a;     // [R][T]
  float* reward;  // [R][T]
  uint8_t* term;  // [R][T]
  float* boot;    // [R][T]
  float* prio;    // [R][seq+n]   per-step priorities (batchSeqPriority_)
  float *h0, *c0, *nh0, *nc0;  // [R][512]  batchH0_ / batchNextH0_
  int T, A, seq, burn, n;
};

__device__ __forceinline__ void block_copy16(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int64_t bytes) {
  const uint4* s4 = reinterpret_cast<const uint4*>(src);
  uint4* d4 = reinterpret_cast<uint4*>(dst);
  for (int64_t i = threadIdx.x; i < (bytes >> 4); i += blockDim.x) d4[i] = s4[i];
}
__device__ __forceinline__ void block_zero16(uint8_t* __restrict__ dst, int64_t bytes) {
  uint4* d4 = reinterpret_cast<uint4*>(dst);
  for (int64_t i = threadIdx.x; i < (bytes >> 4); i += blockDim.x) d4[i] = make_uint4(0, 0, 0, 0);
}

// R2D2TransitionBuffer::push :51-71 for every env: write this step's transition into its window slot
__global__ __launch_bounds__(kT) void r2d2_write_step(Windows w, const int32_t* __restrict__ slot,
                                                     const uint8_t* __restrict__ flags,
                                                     const uint8_t* __restrict__ obs, const float* __restrict__ eps,
                                                     const float* __restrict__ legal, const int64_t* __restrict__ act,
                                                     const float* __restrict__ reward,
                                                     const uint8_t* __restrict__ term, const float* __restrict__ boot,
                                                     const float* __restrict__ prio_step,
                                                     const float* __restrict__ hid_h, const float* __restrict__ hid_c) {
  const int i = blockIdx.x;
  const int j = slot[i];
  block_copy16(w.s + ((int64_t)i * w.T + j) * kObs, obs + (int64_t)i * kObs, kObs);
  const int64_t o = (int64_t)i * w.T + j;
  for (int k = threadIdx.x; k < w.A; k += blockDim.x) w.legal[o * w.A + k] = legal[(int64_t)i * w.A + k];
  if (threadIdx.x == 0) {
    w.eps[o] = eps[i];
    w.a[o] = act[i];
    w.reward[o] = reward[i];
    w.term[o] = term[i];
    w.boot[o] = boot[i];
    w.prio[(int64_t)i * (w.seq + w.n) + (j - w.burn)] = prio_step[i];
  }
  const uint8_t f = flags[i];
  if (f & 1)  // batchH0_[i] <- hid :41
    for (int k = threadIdx.x; k < kHid; k += blockDim.x) {
      w.h0[(int64_t)i * kHid + k] = hid_h[(int64_t)i * kHid + k];
      w.c0[(int64_t)i * kHid + k] = hid_c[(int64_t)i * kHid + k];
    }
  if (f & 2)  // batchNextH0_[i] <- hid :65-68
    for (int k = threadIdx.x; k < kHid; k += blockDim.x) {
      w.nh0[(int64_t)i * kHid + k] = hid_h[(int64_t)i * kHid + k];
      w.nc0[(int64_t)i * kHid + k] = hid_c[(int64_t)i * kHid + k];
    }
}

// padLike (types.cc:69-80) over slot ranges: zeros everywhere, terminal = 1, optional priority 0
__global__ __launch_bounds__(kT) void r2d2_pad(Windows w, const int32_t* __restrict__ ranges /*[n][4]*/) {
  const int env = ranges[blockIdx.x * 4 + 0], begin = ranges[blockIdx.x * 4 + 1], end = ranges[blockIdx.x * 4 + 2];
  const int zero_prio = ranges[blockIdx.x * 4 + 3];
  const int j = begin + blockIdx.y;
  if (j >= end) return;
  const int64_t o = (int64_t)env * w.T + j;
  block_zero16(w.s + o * kObs, kObs);
  for (int k = threadIdx.x; k < w.A; k += blockDim.x) w.legal[o * w.A + k] = 0.f;
  if (threadIdx.x == 0) {
    w.eps[o] = 0.f;
    w.a[o] = 0;
    w.reward[o] = 0.f;
    w.term[o] = 1;
    w.boot[o] = 0.f;
    if (zero_prio) w.prio[(int64_t)env * (w.seq + w.n) + (j - w.burn)] = 0.f;
  }
}

// carry-over :119-139: slots [seq, seq+burn+n) -> [0, burn+n) in ascending order (the ranges may
// overlap when burn+n > seq), priorities by the reference's effective rule, h0 <- next_h0
__global__ __launch_bounds__(kT) void r2d2_carry(Windows w, const int32_t* __restrict__ envs) {
  const int env = envs[blockIdx.x];
  const int64_t base = (int64_t)env * w.T;
  for (int j = 0; j < w.burn + w.n; ++j) {
    const int64_t d = base + j, s = base + w.seq + j;
    block_copy16(w.s + d * kObs, w.s + s * kObs, kObs);
    for (int k = threadIdx.x; k < w.A; k += blockDim.x) w.legal[d * w.A + k] = w.legal[s * w.A + k];
    if (threadIdx.x == 0) {
      w.eps[d] = w.eps[s];
      w.a[d] = w.a[s];
      w.reward[d] = w.reward[s];
      w.term[d] = w.term[s];
      w.boot[d] = w.boot[s];
    }
    __syncthreads();
  }
  float* p = w.prio + (int64_t)env * (w.seq + w.n);
  if (threadIdx.x == 0)
    for (int j = w.burn; j < w.n; ++j) p[j] = p[w.seq + j];  // see r2d2_seq_core.h on the stale entries
  for (int k = threadIdx.x; k < kHid; k += blockDim.x) {
    w.h0[(int64_t)env * kHid + k] = w.nh0[(int64_t)env * kHid + k];
    w.c0[(int64_t)env * kHid + k] = w.nc0[(int64_t)env * kHid + k];
  }
}

// priority rows + lengths of every emitted sequence, read from the PRE-carry windows
__global__ void r2d2_collect(Windows w, const int32_t* __restrict__ emits /*[n][3] env,len,second*/, int nseq,
                             float* __restrict__ prow, float* __restrict__ lens) {
  const int q = blockIdx.x;
  if (q >= nseq) return;
  const int env = emits[q * 3], len = emits[q * 3 + 1], second = emits[q * 3 + 2];
  const float* p = w.prio + (int64_t)env * (w.seq + w.n);
  for (int j = threadIdx.x; j < w.seq; j += blockDim.x) {
    float v;
    if (!second) v = p[j];
    else if (j < w.n) v = (j >= w.burn) ? p[w.seq + j] : p[j];
    else v = 0.f;
    prow[(int64_t)q * w.seq + j] = v;
  }
  if (threadIdx.x == 0) lens[q] = (float)len;
}

// R2D2Agent.aggregate_priority  r2d2.py:103-120 (one lane per sequence; seq_len <= a few hundred)
__global__ void r2d2_aggregate(const float* __restrict__ prow, const float* __restrict__ lens, int nseq, int seq,
                               int burn, float eta, float one_minus_eta, float* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nseq) return;
  float sum = 0.f, mx = -INFINITY;
  for (int t = 0; t < seq; ++t) {
    const float m = ((float)t < lens[q]) ? 1.f : 0.f;
    const float v = prow[(int64_t)q * seq + t] * m;
    sum += v;
    mx = fmaxf(mx, v);
  }
  const float mean = sum / (lens[q] - (float)burn);
  out[q] = __fadd_rn(__fmul_rn(eta, mx), __fmul_rn(one_minus_eta, mean));
}

// setRewardAndTerminal :258-267: finished envs restart from the zero state
__global__ void r2d2_reset_hidden(const uint8_t* __restrict__ term, int R, float* __restrict__ h, float* __restrict__ c) {
  const int i = blockIdx.x;
  if (i >= R || !term[i]) return;
  for (int k = threadIdx.x; k < kHid; k += blockDim.x) {
    h[(int64_t)i * kHid + k] = 0.f;
    c[(int64_t)i * kHid + k] = 0.f;
  }
}

}  // namespace

struct rela_r2d2_actor {
  int device = 0;
  int R = 0, K = 0, A = 0, n = 0, seq = 0, burn = 0, T = 0;
  float gamma = 0.f, gamma_n = 0.f, eta = 0.f, one_minus_eta = 0.f;
  rela_replay* replay = nullptr;
  uint64_t seed = 0, act_calls = 0;
  std::atomic<int64_t> num_act{0};
  int head = 0, count = 0, cur = -1;
  // n-step ring (dqn_actor.h:120-123) + hidden history (r2d2_actor.h:345)
  uint8_t* obs = nullptr;
  int64_t* act = nullptr;
  float* rew = nullptr;
  uint8_t* term = nullptr;
  float *hist_h = nullptr, *hist_c = nullptr;  // [n+1][R][512]
  float *hid_h = nullptr, *hid_c = nullptr, *tmp_h = nullptr, *tmp_c = nullptr;  // [R][512]
  float *eps = nullptr, *legal = nullptr;          // current values (uploaded / written by the caller)
  float *eps_hist = nullptr, *legal_hist = nullptr;  // [n+1][R], [n+1][R][A]: snapshots per history slot
  float* q = nullptr;  // [4][R][A]: adv(act), q_online, adv_next, q_target
  float* q_hist = nullptr;  // [n+1][R][A]: act()'s own Q table of every history slot
  // weights and history slot the advantages in q[0] (written by act) belong to
  const rela_lstmnet* q_net = nullptr;
  uint64_t q_version = 0;
  int q_slot = -1;
  int reuse_mode = 1;  // 0: recompute everything, 1: reuse both act() steps, 2: only the one of next_obs
  std::vector<const rela_lstmnet*> qh_net;  // net / weight version act() evaluated every history slot with
  std::vector<uint64_t> qh_version;
  float *out_r = nullptr, *out_b = nullptr, *prio_step = nullptr;
  uint8_t* out_t = nullptr;
  Windows w{};
  float *prow = nullptr, *lens = nullptr, *agg = nullptr;  // [2R][seq], [2R], [2R]
  int32_t *d_slot = nullptr, *d_ranges = nullptr, *d_emits = nullptr, *d_envs = nullptr;
  int32_t* d_gather = nullptr;  // [2][3R]: destination offsets, source envs and emit indices of one batch
  uint8_t* d_flags = nullptr;
  uint8_t* restart = nullptr;
  uint8_t* fresh_planes = nullptr;  // [R][7056] staging of the newest plane of every row (rela_r2d2_actor_plane_stage)  // [R] rela_r2d2_actor_slide_stacks
  void* ws = nullptr;
  int64_t ws_bytes = 0;
  std::vector<uint8_t> h_term;  // [n+1][R] host copy: the bookkeeping needs the flags
  SeqBook* book = nullptr;
  SeqPlan plan;
  HostStage stage;  // pinned staging of the per-tick index plans (one segment per post_step)
};

extern "C" int rela_r2d2_actor_create(rela_r2d2_actor** out, int rows, int group_rows, int num_action, int multi_step,
                                      float gamma, int seq_len, int burn_in, double eta, rela_replay* replay,
                                      uint64_t seed, int device) {
  RELA_CHECK(out && rows >= 1 && group_rows >= 1 && rows % group_rows == 0 && num_action >= 1 && num_action <= 31 &&
                 multi_step >= 1 && seq_len >= 1 && burn_in >= 0,
             RELA_EINVAL, "rela_r2d2_actor_create: bad arguments");
  RELA_CHECK(burn_in <= seq_len && multi_step <= seq_len, RELA_EINVAL,
             "rela_r2d2_actor_create: needs burn_in <= seq_len and multi_step <= seq_len");  // r2d2_actor.h:25-26
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_last_error("rela_r2d2_actor_create: HIP device %d not available (%d visible); there is no CPU path", device,
                   ndev);
    return RELA_ENODEV;
  }
  DeviceGuard g(device);
  auto* a = new rela_r2d2_actor();
  a->device = device;
  a->R = rows;
  a->K = group_rows;
  a->A = num_action;
  a->n = multi_step;
  a->seq = seq_len;
  a->burn = burn_in;
  a->T = burn_in + seq_len + multi_step;
  a->gamma = gamma;
  a->gamma_n = (float)pow((double)gamma, (double)multi_step);
  a->eta = (float)eta;
  a->one_minus_eta = (float)(1.0 - eta);  // TorchScript folds (1.0 - self.eta) in double, r2d2.py:119
  a->replay = replay;
  a->seed = seed;
  const size_t H = (size_t)multi_step + 1, R = (size_t)rows, A = (size_t)num_action, T = (size_t)a->T;
  auto alloc = [&](void** p, size_t bytes) -> int {
    RELA_HIP(hipMalloc(p, bytes));
    RELA_HIP(hipMemset(*p, 0, bytes));
    return RELA_OK;
  };
#define RELA_ALLOC(ptr, bytes)                                  \
  do {                                                          \
    int _rc = alloc(reinterpret_cast<void**>(&(ptr)), (bytes)); \
    if (_rc != RELA_OK) return _rc;                             \
  } while (0)
  RELA_ALLOC(a->obs, H * R * kObs);
  RELA_ALLOC(a->act, H * R * sizeof(int64_t));
  RELA_ALLOC(a->rew, H * R * sizeof(float));
  RELA_ALLOC(a->term, H * R);
  RELA_ALLOC(a->hist_h, H * R * kHid * sizeof(float));
  RELA_ALLOC(a->hist_c, H * R * kHid * sizeof(float));
  RELA_ALLOC(a->hid_h, R * kHid * sizeof(float));
  RELA_ALLOC(a->hid_c, R * kHid * sizeof(float));
  RELA_ALLOC(a->tmp_h, R * kHid * sizeof(float));
  RELA_ALLOC(a->tmp_c, R * kHid * sizeof(float));
  RELA_ALLOC(a->eps, R * sizeof(float));
  RELA_ALLOC(a->legal, R * A * sizeof(float));
  RELA_ALLOC(a->eps_hist, H * R * sizeof(float));
  RELA_ALLOC(a->legal_hist, H * R * A * sizeof(float));
  RELA_ALLOC(a->q, 4 * R * A * sizeof(float));
  RELA_ALLOC(a->q_hist, (size_t)(multi_step + 1) * R * A * sizeof(float));
  a->qh_net.assign((size_t)multi_step + 1, nullptr);
  a->qh_version.assign((size_t)multi_step + 1, 0);
  RELA_ALLOC(a->out_r, R * sizeof(float));
  RELA_ALLOC(a->out_b, R * sizeof(float));
  RELA_ALLOC(a->prio_step, R * sizeof(float));
  RELA_ALLOC(a->out_t, R);
  Windows& w = a->w;
  w.T = a->T;
  w.A = num_action;
  w.seq = seq_len;
  w.burn = burn_in;
  w.n = multi_step;
  RELA_ALLOC(w.s, R * T * kObs);
  RELA_ALLOC(w.eps, R * T * sizeof(float));
  RELA_ALLOC(w.legal, R * T * A * sizeof(float));
  RELA_ALLOC(w.a, R * T * sizeof(int64_t));
  RELA_ALLOC(w.reward, R * T * sizeof(float));
  RELA_ALLOC(w.term, R * T);
  RELA_ALLOC(w.boot, R * T * sizeof(float));
  RELA_ALLOC(w.prio, R * (size_t)(seq_len + multi_step) * sizeof(float));
  RELA_ALLOC(w.h0, R * kHid * sizeof(float));
  RELA_ALLOC(w.c0, R * kHid * sizeof(float));
  RELA_ALLOC(w.nh0, R * kHid * sizeof(float));
  RELA_ALLOC(w.nc0, R * kHid * sizeof(float));
  RELA_ALLOC(a->prow, 2 * R * (size_t)seq_len * sizeof(float));
  RELA_ALLOC(a->lens, 2 * R * sizeof(float));
  RELA_ALLOC(a->agg, 2 * R * sizeof(float));
  RELA_ALLOC(a->d_slot, R * sizeof(int32_t));
  RELA_ALLOC(a->d_flags, R);
  RELA_ALLOC(a->d_ranges, 2 * R * 4 * sizeof(int32_t));
  RELA_ALLOC(a->d_emits, 2 * R * 3 * sizeof(int32_t));
  RELA_ALLOC(a->d_gather, 2 * R * 3 * sizeof(int32_t));
  RELA_ALLOC(a->d_envs, R * sizeof(int32_t));
#undef RELA_ALLOC
  a->ws_bytes = rela_lstmnet_workspace_bytes(nullptr, rows);
  RELA_HIP(hipMalloc(&a->ws, (size_t)a->ws_bytes));
  {
    std::vector<float> ones(R * A, 1.0f);
    RELA_HIP(hipMemcpy(a->legal, ones.data(), R * A * sizeof(float), hipMemcpyHostToDevice));
  }
  a->h_term.assign(H * R, 0);
  {
    const int rc = a->stage.init(R * 320 + 4096);
    RELA_CHECK(rc == RELA_OK, rc, "rela_r2d2_actor_create: pinned staging buffer");
  }
  a->book = new SeqBook(rows, multi_step, seq_len, burn_in);
  *out = a;
  return RELA_OK;
}

extern "C" void rela_r2d2_actor_destroy(rela_r2d2_actor* a) {
  if (!a) return;
  DeviceGuard g(a->device);
  (void)hipDeviceSynchronize();
  void* ps[] = {a->obs,   a->act,    a->rew,    a->term,      a->hist_h, a->hist_c, a->hid_h,  a->hid_c,  a->tmp_h,
                a->tmp_c, a->eps,    a->legal,  a->eps_hist, a->legal_hist, a->q,         a->out_r,  a->out_b,  a->prio_step, a->out_t, a->w.s,
                a->w.eps, a->w.legal, a->w.a,   a->w.reward,  a->w.term, a->w.boot, a->w.prio, a->w.h0,   a->w.c0,
                a->w.nh0, a->w.nc0,  a->prow,   a->lens,      a->agg,    a->d_slot, a->d_flags, a->d_ranges, a->d_emits, a->d_gather,
                a->d_envs, a->ws, a->q_hist, a->restart, a->fresh_planes};
  for (void* p : ps) (void)hipFree(p);
  a->stage.destroy();
  delete a->book;
  delete a;
}

static inline int next_slot(const rela_r2d2_actor* a) { return (a->head + a->count) % (a->n + 1); }

extern "C" void* rela_r2d2_actor_obs_slot(rela_r2d2_actor* a) {
  return a ? a->obs + (size_t)next_slot(a) * a->R * kObs : nullptr;
}
extern "C" void* rela_r2d2_actor_plane_stage(rela_r2d2_actor* a) {
  if (!a) return nullptr;
  if (!a->fresh_planes) {
    DeviceGuard g(a->device);
    if (hipMalloc(&a->fresh_planes, (size_t)a->R * 84 * 84) != hipSuccess) a->fresh_planes = nullptr;
  }
  return a->fresh_planes;
}
extern "C" int rela_r2d2_actor_slide_stacks(rela_r2d2_actor* a, const uint8_t* restart_host, void* stream_) {
  RELA_CHECK(a && restart_host, RELA_EINVAL, "rela_r2d2_actor_slide_stacks: bad arguments");
  RELA_CHECK(a->fresh_planes, RELA_ESTATE, "rela_r2d2_actor_slide_stacks: no plane was staged (rela_r2d2_actor_plane_stage)");
  RELA_CHECK(a->act_calls > 0, RELA_ESTATE, "rela_r2d2_actor_slide_stacks: the first observation must be uploaded whole");
  RELA_CHECK(a->count <= a->n, RELA_ESTATE, "rela_r2d2_actor_slide_stacks: act() twice without post_step()");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(a->device);
  if (!a->restart) RELA_HIP(hipMalloc(&a->restart, (size_t)a->R));
  RELA_HIP(hipMemcpyAsync(a->restart, restart_host, (size_t)a->R, hipMemcpyHostToDevice, s));
  const int H = a->n + 1, slot = next_slot(a), prev = (slot + H - 1) % H;
  return slide_stacks(a->obs + (size_t)slot * a->R * kObs, a->obs + (size_t)prev * a->R * kObs, a->fresh_planes, a->restart,
                      a->R, s);
}
extern "C" int rela_r2d2_actor_set_reuse(rela_r2d2_actor* a, int on) {
  RELA_CHECK(a, RELA_EINVAL, "rela_r2d2_actor_set_reuse: bad arguments");
  RELA_CHECK(on >= 0 && on <= 2, RELA_EINVAL, "rela_r2d2_actor_set_reuse: 0 (off), 1 (on) or 2 (next_obs only)");
  a->reuse_mode = on;
  return RELA_OK;
}
extern "C" int64_t rela_r2d2_actor_num_act(const rela_r2d2_actor* a) { return a ? a->num_act.load() : 0; }
extern "C" const float* rela_r2d2_actor_hidden_dev(const rela_r2d2_actor* a, int which) {
  return a ? (which ? a->hid_c : a->hid_h) : nullptr;
}
extern "C" const float* rela_r2d2_actor_last_priority_dev(const rela_r2d2_actor* a) { return a ? a->prio_step : nullptr; }

extern "C" int rela_r2d2_actor_act(rela_r2d2_actor* a, const rela_lstmnet* online, const uint8_t* obs_host,
                                   const float* eps_host, const float* legal_host, int64_t* action_host,
                                   const int64_t** action_dev_out, void* stream_) {
  RELA_CHECK(a && online, RELA_EINVAL, "rela_r2d2_actor_act: bad arguments");
  RELA_CHECK(rela_lstmnet_num_action(online) == a->A, RELA_EINVAL, "rela_r2d2_actor_act: net has %d actions, actor %d",
             rela_lstmnet_num_action(online), a->A);
  RELA_CHECK(a->count <= a->n, RELA_ESTATE, "rela_r2d2_actor_act: act() twice without post_step()");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(a->device);
  const int slot = next_slot(a);
  const size_t R = (size_t)a->R, HB = R * kHid * sizeof(float);
  uint8_t* obs = a->obs + (size_t)slot * R * kObs;
  if (obs_host) RELA_HIP(hipMemcpyAsync(obs, obs_host, R * kObs, hipMemcpyHostToDevice, s));
  if (eps_host) RELA_HIP(hipMemcpyAsync(a->eps, eps_host, R * sizeof(float), hipMemcpyHostToDevice, s));
  if (legal_host) RELA_HIP(hipMemcpyAsync(a->legal, legal_host, R * a->A * sizeof(float), hipMemcpyHostToDevice, s));
  // historyHidden_.push_back(hidden_) :226-228
  RELA_HIP(dev_copy2(a->hist_h + (size_t)slot * R * kHid, a->hid_h, HB, a->hist_c + (size_t)slot * R * kHid, a->hid_c, HB, s));
  float* eps_s = a->eps_hist + (size_t)slot * R;
  float* legal_s = a->legal_hist + (size_t)slot * R * a->A;
  RELA_HIP(dev_copy2(eps_s, a->eps, R * sizeof(float), legal_s, a->legal, R * a->A * sizeof(float), s));
  // (the dueling Q of this step goes to the slot's table: compute_priority's online_net(obs, hid) n ticks from now)
  a->qh_net[slot] = nullptr;
  int rc = rela_lstmnet_step(online, a->R, obs, legal_s, a->hid_h, a->hid_c, a->tmp_h, a->tmp_c,
                             a->q_hist + (size_t)slot * R * a->A, a->q, a->ws, a->ws_bytes, s);
  if (rc != RELA_OK) return rc;
  a->qh_net[slot] = online;
  a->qh_version[slot] = rela_lstmnet_version(online);
  std::swap(a->hid_h, a->tmp_h);  // hidden_ <- new state :241
  std::swap(a->hid_c, a->tmp_c);
  int64_t* act = a->act + (size_t)slot * R;
  rc = rela_apex_act_from_q(a->R, a->A, a->K, a->q, legal_s, eps_s, a->seed, a->act_calls * (uint64_t)a->R, act, s);
  if (rc != RELA_OK) return rc;
  a->q_net = online;
  a->q_version = rela_lstmnet_version(online);
  a->q_slot = slot;
  a->act_calls += 1;
  a->cur = slot;
  a->num_act += a->R;
  if (action_dev_out) *action_dev_out = act;
  if (action_host) {
    RELA_HIP(hipMemcpyAsync(action_host, act, R * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    RELA_HIP(hipStreamSynchronize(s));
  }
  return RELA_OK;
}

namespace {
int upload_ranges(rela_r2d2_actor* a, const std::vector<SeqRange>& rs, hipStream_t s) {
  if (rs.empty()) return RELA_OK;
  std::vector<int32_t> flat;
  int maxlen = 0;
  for (const auto& r : rs) {
    flat.insert(flat.end(), {r.env, r.begin, r.end, r.zero_prio});
    maxlen = std::max(maxlen, r.end - r.begin);
  }
  RELA_HIP(a->stage.h2d(a->d_ranges, flat.data(), flat.size() * sizeof(int32_t), s));
  hipLaunchKernelGGL(r2d2_pad, dim3((unsigned)rs.size(), (unsigned)maxlen), dim3(kT), 0, s, a->w, a->d_ranges);
  RELA_LAUNCH_CHECK();
  return RELA_OK;  // `flat` may die: the upload reads the pinned copy
}
}  // namespace

extern "C" int rela_r2d2_actor_post_step(rela_r2d2_actor* a, const float* reward_host, const uint8_t* terminal_host,
                                         const rela_lstmnet* online, const rela_lstmnet* target, int nonblocking,
                                         int* n_sequences, void* stream_) {
  RELA_CHECK(a && reward_host && terminal_host && online && target, RELA_EINVAL,
             "rela_r2d2_actor_post_step: bad arguments");
  RELA_CHECK(a->replay, RELA_ESTATE, "rela_r2d2_actor_post_step: evaluation actor has no replay");
  RELA_CHECK(a->cur >= 0, RELA_ESTATE, "rela_r2d2_actor_post_step: no act() to attach the reward to");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(a->device);
  if (n_sequences) *n_sequences = 0;
  const size_t R = (size_t)a->R;
  const int H = a->n + 1;
  struct StageCall {  // one pinned segment per call, released (event) on every return path
    HostStage& st;
    hipStream_t s;
    StageCall(HostStage& st_, hipStream_t s_) : st(st_), s(s_) { st.begin(); }
    ~StageCall() { st.end(s); }
  } stage_call(a->stage, s);
  RELA_HIP(a->stage.h2d(a->rew + (size_t)a->cur * R, reward_host, R * sizeof(float), s));
  RELA_HIP(a->stage.h2d(a->term + (size_t)a->cur * R, terminal_host, R, s));
  memcpy(&a->h_term[(size_t)a->cur * R], terminal_host, R);
  hipLaunchKernelGGL(r2d2_reset_hidden, dim3(a->R), dim3(128), 0, s, a->term + (size_t)a->cur * R, a->R, a->hid_h,
                     a->hid_c);
  a->cur = -1;
  a->count += 1;
  if (a->count < H) return RELA_OK;  // multiStepBuffer_.canPop :276-279

  const int first = a->head, last = (a->head + a->n) % H;
  int rc = rela_nstep_return(a->n, a->R, a->gamma, first, a->rew, a->term, a->out_r, a->out_b, a->out_t, s);
  if (rc != RELA_OK) return rc;
  const uint8_t* obs_t = a->obs + (size_t)first * R * kObs;
  const uint8_t* obs_n = a->obs + (size_t)last * R * kObs;
  const float *h_t = a->hist_h + (size_t)first * R * kHid, *c_t = a->hist_c + (size_t)first * R * kHid;
  const float *h_n = a->hist_h + (size_t)last * R * kHid, *c_n = a->hist_c + (size_t)last * R * kHid;
  const size_t QA = R * a->A;
  const float* legal_t = a->legal_hist + (size_t)first * R * a->A;
  const float* legal_n = a->legal_hist + (size_t)last * R * a->A;
  const float* eps_t = a->eps_hist + (size_t)first * R;
  // compute_priority  r2d2.py:76-100
  // online_net(obs, hid) :89 is the step act() ran n ticks ago on the same frames with the same recurrent state
  // (historyHidden_.front()) and legal mask: with unchanged weights its Q table in q_hist[first] is reused
  const float* q_online_t = a->q_hist + (size_t)first * QA;
  if (!(a->reuse_mode == 1 && a->qh_net[first] == online && a->qh_version[first] == rela_lstmnet_version(online))) {
    rc = rela_lstmnet_step(online, a->R, obs_t, legal_t, h_t, c_t, a->tmp_h, a->tmp_c, a->q + QA, nullptr, a->ws,
                           a->ws_bytes, s);
    if (rc != RELA_OK) return rc;
    q_online_t = a->q + QA;
  }
  // online_net.act(next_obs, next_hid) :91 is the very step act() ran on this tick (same frames, same
  // recurrent state, same legal mask): with unchanged weights its advantages in q[0] are reused
  const float* adv_next = a->q;
  if (!(a->reuse_mode != 0 && a->q_net == online && a->q_version == rela_lstmnet_version(online) && a->q_slot == last)) {
    rc = rela_lstmnet_step(online, a->R, obs_n, legal_n, h_n, c_n, a->tmp_h, a->tmp_c, nullptr, a->q + 2 * QA, a->ws,
                           a->ws_bytes, s);
    if (rc != RELA_OK) return rc;
    adv_next = a->q + 2 * QA;
  }
  rc = rela_lstmnet_step(target, a->R, obs_n, legal_n, h_n, c_n, a->tmp_h, a->tmp_c, a->q + 3 * QA, nullptr, a->ws,
                         a->ws_bytes, s);  // target_net(next_obs, next_hid, next_action) :93
  if (rc != RELA_OK) return rc;
  const int64_t* act_t = a->act + (size_t)first * R;
  rc = rela_apex_td_from_q(a->R, a->A, a->K, q_online_t, adv_next, a->q + 3 * QA, legal_n, act_t, a->out_r, a->out_b,
                           a->gamma_n, nullptr, a->prio_step, s);
  if (rc != RELA_OK) return rc;

  // r2d2Buffer_.push(transition, priority, hid) :289
  SeqPlan& plan = a->plan;
  a->book->step(&a->h_term[(size_t)first * R], &plan);
  rc = upload_ranges(a, plan.front_pad, s);
  if (rc != RELA_OK) return rc;
  RELA_HIP(a->stage.h2d(a->d_slot, plan.write_slot.data(), R * sizeof(int32_t), s));
  RELA_HIP(a->stage.h2d(a->d_flags, plan.flags.data(), R, s));
  hipLaunchKernelGGL(r2d2_write_step, dim3(a->R), dim3(kT), 0, s, a->w, a->d_slot, a->d_flags, obs_t, eps_t, legal_t,
                     act_t, a->out_r, a->out_t, a->out_b, a->prio_step, h_t, c_t);
  RELA_LAUNCH_CHECK();
  rc = upload_ranges(a, plan.tail_pad, s);
  if (rc != RELA_OK) return rc;

  if (plan.can_pop) {  // :292-301
    const int nseq = (int)plan.emits.size();
    std::vector<int32_t> em;
    for (const auto& e : plan.emits) em.insert(em.end(), {e.env, e.len, e.second});
    RELA_HIP(a->stage.h2d(a->d_emits, em.data(), em.size() * sizeof(int32_t), s));
    hipLaunchKernelGGL(r2d2_collect, dim3(nseq), dim3(128), 0, s, a->w, a->d_emits, nseq, a->prow, a->lens);
    hipLaunchKernelGGL(r2d2_aggregate, dim3(ceil_div(nseq, 64)), dim3(64), 0, s, a->prow, a->lens, nseq, a->seq, a->burn,
                       a->eta, a->one_minus_eta, a->agg);
    RELA_LAUNCH_CHECK();
    // A pop of a large shard can exceed what a blocking append can ever get (ring - capacity slots:
    // sample() evicts down to capacity only).  The pop therefore goes in pieces of at most that many
    // sequences, cut at env boundaries (an env's second, short sequence directly follows its first):
    // reserve, write the first sequences, carry those envs' windows, write their second sequences, commit.
    int cap = 0, ring = 0;
    rc = rela_replay_limits(a->replay, &cap, &ring);
    if (rc != RELA_OK) return rc;
    const int max_block = ring - cap > 1 ? ring - cap : 1;
    const Windows& w = a->w;
    int inserted = 0, dropped = 0;
    size_t ci = 0, pi = 0;  // cursors into carry_env / carry_pad (both ascending in env)
    for (int q0 = 0; q0 < nseq && rc == RELA_OK;) {
      int q1 = nseq - q0 <= max_block ? nseq : q0 + max_block;
      if (q1 < nseq && plan.emits[q1].second) q1 = (q1 - 1 > q0) ? q1 - 1 : q1 + 1;  // keep a pair together
      const int cnt = q1 - q0;
      const int env_hi = plan.emits[q1 - 1].env;
      std::vector<int32_t> carry;
      while (ci < plan.carry_env.size() && plan.carry_env[ci] <= env_hi) carry.push_back(plan.carry_env[ci++]);
      std::vector<SeqRange> pads;
      while (pi < plan.carry_pad.size() && plan.carry_pad[pi].env <= env_hi) pads.push_back(plan.carry_pad[pi++]);
      auto carry_piece = [&]() -> int {
        if (carry.empty()) return RELA_OK;
        RELA_HIP(a->stage.h2d(a->d_envs, carry.data(), carry.size() * sizeof(int32_t), s));
        hipLaunchKernelGGL(r2d2_carry, dim3((unsigned)carry.size()), dim3(kT), 0, s, a->w, a->d_envs);
        RELA_LAUNCH_CHECK();
        return upload_ranges(a, pads, s);
      };
      int slot0 = 0;
      rc = rela_replay_begin_add(a->replay, cnt, nonblocking, &slot0);
      if (rc == RELA_EWOULDBLOCK) {
        // dropped piece: the windows must still advance exactly as if it had been stored
        dropped = 1;
        rc = carry_piece();
        q0 = q1;
        continue;
      }
      if (rc != RELA_OK) break;
      // all sequences of one kind leave in ONE gathered write per field: destination offset q - q0, source
      // row = the env's window (lens is indexed by q itself)
      auto emit_batch = [&](bool second) -> int {
        std::vector<int32_t> dst, envs, qs;
        for (int q = q0; q < q1; ++q)
          if ((plan.emits[q].second != 0) == second) {
            dst.push_back(q - q0);
            envs.push_back(plan.emits[q].env);
            qs.push_back(q);
          }
        const int m = (int)dst.size();
        if (m == 0) return RELA_OK;
        int32_t* d_dst = a->d_gather + (second ? 3 * (size_t)a->R : 0);
        int32_t* d_env = d_dst + m;
        int32_t* d_q = d_env + m;
        std::vector<int32_t> all(dst);
        all.insert(all.end(), envs.begin(), envs.end());
        all.insert(all.end(), qs.begin(), qs.end());
        RELA_HIP(a->stage.h2d(d_dst, all.data(), all.size() * sizeof(int32_t), s));  // pinned copy: `all` may die
        const void* bases[10] = {w.s, w.eps, w.legal, w.a, w.reward, w.term, w.boot, w.h0, w.c0, a->lens};
        const int32_t* idx[10] = {d_env, d_env, d_env, d_env, d_env, d_env, d_env, d_env, d_env, d_q};
        return rela_replay_write_rows_gather(a->replay, slot0, m, d_dst, bases, idx, s);
      };
      rc = emit_batch(false);
      if (rc == RELA_OK) rc = carry_piece();
      if (rc == RELA_OK) rc = emit_batch(true);
      if (rc == RELA_OK) rc = rela_replay_commit_add(a->replay, slot0, cnt, a->agg + q0, s);
      if (rc == RELA_OK) inserted += cnt;
      else (void)rela_replay_abort_add(a->replay, slot0, cnt);  // never leave a reservation uncommitted
      q0 = q1;
    }
    if (rc != RELA_OK) return rc;
    if (n_sequences) *n_sequences = inserted;
    if (dropped) rc = RELA_EWOULDBLOCK;
    if (rc != RELA_OK && rc != RELA_EWOULDBLOCK) return rc;
  }
  a->head = (a->head + 1) % H;  // multiStepBuffer_ / historyHidden_ pop_front :283-286
  a->count -= 1;
  return plan.can_pop && rc == RELA_EWOULDBLOCK ? RELA_EWOULDBLOCK : RELA_OK;
}
