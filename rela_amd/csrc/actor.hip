// actor.hip -- device-resident Ape-X actor shard (C ABI: rela_apex_actor_*).
//
// Restates the per-step work BasicThreadLoop::mainLoop (rela/thread_loop.h:74-105) drives through
// DQNActor (rela/dqn_actor.h:126-211) and MultiStepTransitionBuffer (:15-124):
//   act        push (obs, action) :23-29,153-171      -> 1 trunk forward + eps-greedy
//   post_step  push (r, t) :31-40; canPop :46-48; popTransition :58-106; computePriority :193-203
//              (online(s_t), online(s_t+n), target(s_t+n), apex.py:30-45; online(s_t+n) is act()'s own
//              forward of this tick and online(s_t) is act()'s forward of n ticks ago: each is reused when
//              the online weights were not re-loaded in between); replay add :189
// The deque of :120-123 is a ring of multi_step+1 slots in HBM; "pop_front" is a head increment.
#include <atomic>
#include <cmath>
#include <vector>

#include "common.h"

using namespace rela_amd;

struct rela_apex_actor {
  int device = 0;
  int R = 0, K = 0, A = 0, n = 0;
  float gamma = 0.f, gamma_n = 0.f;
  rela_replay* replay = nullptr;
  uint64_t seed = 0;
  uint64_t act_calls = 0;
  std::atomic<int64_t> num_act{0};
  int head = 0, count = 0, cur = -1;
  // device state
  uint8_t* obs = nullptr;   // [n+1][R][28224]
  int64_t* act = nullptr;   // [n+1][R]
  float* rew = nullptr;     // [n+1][R]
  uint8_t* term = nullptr;  // [n+1][R]
  float* eps = nullptr;     // [R]      current values (callers may write them on the device) ...
  float* legal = nullptr;   // [R][A]
  float* eps_hist = nullptr;    // [n+1][R]     ... snapshotted per history slot by act(), because the
  float* legal_hist = nullptr;  // [n+1][R][A]  transition's obs side carries those of time t-n (:84-90)
  float* q = nullptr;       // [4][R][A]   tables recomputed in post_step (1: online(s_t), 2: online(s_t+n), 3: target)
  float* q_hist = nullptr;  // [n+1][R][A] act()'s own Q table of every history slot
  float *out_r = nullptr, *out_b = nullptr, *prio = nullptr;
  uint8_t* out_t = nullptr;
  void* ws = nullptr;
  int64_t ws_bytes = 0;
  // net and weight version act() evaluated every history slot with (q_hist[slot])
  std::vector<const rela_ffnet*> qh_net;
  std::vector<uint64_t> qh_version;
  int q_slot = -1;     // slot of the last act()
  int reuse_mode = 1;  // 0: recompute everything, 1: reuse both act() forwards, 2: only the one of s_t+n
  // frame-stack de-duplication (rela_apex_actor_set_dedup; replay side: rela_replay_set_schema_dedup)
  int dd_ups = 0;                    // 0 = off, 1 = one unit per stack, 4 = one unit per 84x84 plane
  int64_t dd_cap = 0;                // units in the replay's ring
  int32_t* ref_hist = nullptr;       // [n+1][R][ups] unit indices of every history slot's stack
  std::vector<uint8_t> refs_valid;   // [n+1] the slot's units were stored
  std::vector<int64_t> tick_seq;     // first unit sequence number of the last kTickWin ticks (ring by tick)
  int64_t tick = 0, key_tick = -1;   // ticks stored so far; tick of the last keyframe (all planes stored)
  uint8_t* restart = nullptr;
  uint8_t* fresh_planes = nullptr;  // [R][7056] staging of the newest plane of every row (rela_apex_actor_plane_stage)        // [R] rela_apex_actor_slide_stacks: 1 = the row's stack restarts with its new plane
};

namespace {
constexpr int64_t kObs = 4 * 84 * 84;
constexpr int64_t kPlane = 84 * 84;
constexpr int kTickWin = 64;

// references of the stack just acted on (history slot `cur`), per row:
//   ups == 1            : the stack's own unit
//   ups == 4, keyframe  : its four planes (stored together)
//   ups == 4, otherwise : an episode start repeats the new plane four times (GameState::computeFeature,
//                         atari/game_state.h:66-70); any other step slides the previous stack by one plane (:71-74)
__global__ void dedup_make_refs(int32_t* __restrict__ cur, const int32_t* __restrict__ prev,
                                const uint8_t* __restrict__ prev_term, int R, int ups, int keyframe, int32_t first_idx,
                                int64_t cap) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= R) return;
  if (ups == 1) {
    cur[row] = (int32_t)(((int64_t)first_idx + row) % cap);
    return;
  }
  int32_t* c = cur + (size_t)row * 4;
  if (keyframe) {
    for (int k = 0; k < 4; ++k) c[k] = (int32_t)(((int64_t)first_idx + 4 * row + k) % cap);
    return;
  }
  const int32_t fresh = (int32_t)(((int64_t)first_idx + row) % cap);
  if (prev_term[row]) {
    c[0] = c[1] = c[2] = c[3] = fresh;
  } else {
    const int32_t* p = prev + (size_t)row * 4;
    c[0] = p[1], c[1] = p[2], c[2] = p[3], c[3] = fresh;
  }
}
}

extern "C" int rela_apex_actor_create(rela_apex_actor** out, int rows, int group_rows, int num_action, int multi_step,
                                      float gamma, rela_replay* replay, uint64_t seed, int device) {
  RELA_CHECK(out && rows >= 1 && group_rows >= 1 && rows % group_rows == 0 && num_action >= 1 && num_action <= 31 &&
                 multi_step >= 1,
             RELA_EINVAL, "rela_apex_actor_create: bad arguments (rows=%d group=%d A=%d n=%d)", rows, group_rows,
             num_action, multi_step);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_last_error("rela_apex_actor_create: HIP device %d not available (%d visible); there is no CPU path", device,
                   ndev);
    return RELA_ENODEV;
  }
  DeviceGuard g(device);
  auto* a = new rela_apex_actor();
  a->device = device;
  a->R = rows;
  a->K = group_rows;
  a->A = num_action;
  a->n = multi_step;
  a->gamma = gamma;
  a->gamma_n = (float)pow((double)gamma, (double)multi_step);  // gamma ** multi_step, apex.py:44
  a->replay = replay;
  a->seed = seed;
  const size_t H = (size_t)multi_step + 1, R = (size_t)rows, A = (size_t)num_action;
  RELA_HIP(hipMalloc(&a->obs, H * R * kObs));
  RELA_HIP(hipMalloc(&a->act, H * R * sizeof(int64_t)));
  RELA_HIP(hipMalloc(&a->rew, H * R * sizeof(float)));
  RELA_HIP(hipMalloc(&a->term, H * R));
  RELA_HIP(hipMalloc(&a->eps, R * sizeof(float)));
  RELA_HIP(hipMalloc(&a->legal, R * A * sizeof(float)));
  RELA_HIP(hipMalloc(&a->eps_hist, H * R * sizeof(float)));
  RELA_HIP(hipMalloc(&a->legal_hist, H * R * A * sizeof(float)));
  RELA_HIP(hipMalloc(&a->q, 4 * R * A * sizeof(float)));
  RELA_HIP(hipMalloc(&a->q_hist, H * R * A * sizeof(float)));
  a->qh_net.assign(H, nullptr);
  a->qh_version.assign(H, 0);
  RELA_HIP(hipMalloc(&a->out_r, R * sizeof(float)));
  RELA_HIP(hipMalloc(&a->out_b, R * sizeof(float)));
  RELA_HIP(hipMalloc(&a->prio, R * sizeof(float)));
  RELA_HIP(hipMalloc(&a->out_t, R));
  a->ws_bytes = rela_ffnet_workspace_bytes(nullptr, rows);
  RELA_HIP(hipMalloc(&a->ws, (size_t)a->ws_bytes));
  RELA_HIP(hipMemset(a->obs, 0, H * R * kObs));
  RELA_HIP(hipMemset(a->act, 0, H * R * sizeof(int64_t)));
  RELA_HIP(hipMemset(a->rew, 0, H * R * sizeof(float)));
  RELA_HIP(hipMemset(a->term, 0, H * R));
  RELA_HIP(hipMemset(a->eps, 0, R * sizeof(float)));
  {  // legal_move defaults to all ones
    std::vector<float> ones(R * A, 1.0f);
    RELA_HIP(hipMemcpy(a->legal, ones.data(), R * A * sizeof(float), hipMemcpyHostToDevice));
  }
  *out = a;
  return RELA_OK;
}

extern "C" void rela_apex_actor_destroy(rela_apex_actor* a) {
  if (!a) return;
  DeviceGuard g(a->device);
  (void)hipDeviceSynchronize();
  void* ps[] = {a->obs, a->act, a->rew, a->term, a->eps, a->legal, a->q, a->out_r, a->out_b, a->prio, a->out_t, a->ws,
                a->eps_hist, a->legal_hist, a->ref_hist, a->q_hist, a->restart, a->fresh_planes};
  for (void* p : ps) (void)hipFree(p);
  delete a;
}

static inline int next_slot(const rela_apex_actor* a) { return (a->head + a->count) % (a->n + 1); }

extern "C" void* rela_apex_actor_obs_slot(rela_apex_actor* a) {
  return a ? a->obs + (size_t)next_slot(a) * a->R * kObs : nullptr;
}
extern "C" void* rela_apex_actor_plane_stage(rela_apex_actor* a) {
  if (!a) return nullptr;
  if (!a->fresh_planes) {
    DeviceGuard g(a->device);
    if (hipMalloc(&a->fresh_planes, (size_t)a->R * 84 * 84) != hipSuccess) a->fresh_planes = nullptr;
  }
  return a->fresh_planes;
}
extern "C" int rela_apex_actor_slide_stacks(rela_apex_actor* a, const uint8_t* restart_host, void* stream_) {
  RELA_CHECK(a && restart_host, RELA_EINVAL, "rela_apex_actor_slide_stacks: bad arguments");
  RELA_CHECK(a->fresh_planes, RELA_ESTATE, "rela_apex_actor_slide_stacks: no plane was staged (rela_apex_actor_plane_stage)");
  RELA_CHECK(a->act_calls > 0, RELA_ESTATE, "rela_apex_actor_slide_stacks: the first observation must be uploaded whole");
  RELA_CHECK(a->count <= a->n, RELA_ESTATE, "rela_apex_actor_slide_stacks: act() twice without post_step()");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(a->device);
  if (!a->restart) RELA_HIP(hipMalloc(&a->restart, (size_t)a->R));
  RELA_HIP(hipMemcpyAsync(a->restart, restart_host, (size_t)a->R, hipMemcpyHostToDevice, s));
  const int H = a->n + 1, slot = next_slot(a), prev = (slot + H - 1) % H;
  return slide_stacks(a->obs + (size_t)slot * a->R * kObs, a->obs + (size_t)prev * a->R * kObs, a->fresh_planes, a->restart,
                      a->R, s);
}
extern "C" int rela_apex_actor_set_reuse(rela_apex_actor* a, int on) {
  RELA_CHECK(a, RELA_EINVAL, "rela_apex_actor_set_reuse: bad arguments");
  RELA_CHECK(on >= 0 && on <= 2, RELA_EINVAL, "rela_apex_actor_set_reuse: 0 (off), 1 (on) or 2 (next_obs only)");
  a->reuse_mode = on;
  return RELA_OK;
}
extern "C" int rela_apex_actor_set_dedup(rela_apex_actor* a, int units_per_stack) {
  RELA_CHECK(a && a->replay && (units_per_stack == 1 || units_per_stack == 4), RELA_EINVAL,
             "rela_apex_actor_set_dedup: needs a replay and 1 (stack units) or 4 (plane units)");
  RELA_CHECK(a->count == 0 && a->tick == 0 && a->dd_ups == 0, RELA_ESTATE, "rela_apex_actor_set_dedup: call it once, before the first act()");
  int ups = 0;
  int64_t ub = 0, cap = 0;
  int rc = rela_replay_dedup_info(a->replay, &ups, &ub, &cap);
  if (rc != RELA_OK) return rc;
  RELA_CHECK(ups == units_per_stack && ub * ups == kObs, RELA_EINVAL,
             "rela_apex_actor_set_dedup: the replay's schema has %d units of %lld bytes per stack", ups, (long long)ub);
  DeviceGuard g(a->device);
  const size_t H = (size_t)a->n + 1;
  RELA_HIP(hipMalloc(&a->ref_hist, H * (size_t)a->R * ups * sizeof(int32_t)));
  RELA_HIP(hipMemset(a->ref_hist, 0, H * (size_t)a->R * ups * sizeof(int32_t)));
  a->dd_ups = ups;
  a->dd_cap = cap;
  a->refs_valid.assign(H, 0);
  a->tick_seq.assign(kTickWin, 0);
  return RELA_OK;
}
extern "C" float* rela_apex_actor_eps_dev(rela_apex_actor* a) { return a ? a->eps : nullptr; }
extern "C" float* rela_apex_actor_legal_dev(rela_apex_actor* a) { return a ? a->legal : nullptr; }
extern "C" int64_t rela_apex_actor_num_act(const rela_apex_actor* a) { return a ? a->num_act.load() : 0; }
extern "C" const float* rela_apex_actor_last_q_dev(const rela_apex_actor* a) {
  return a ? a->q_hist + (size_t)(a->q_slot < 0 ? 0 : a->q_slot) * a->R * a->A : nullptr;
}
extern "C" const float* rela_apex_actor_last_priority_dev(const rela_apex_actor* a) { return a ? a->prio : nullptr; }

extern "C" int rela_apex_actor_act(rela_apex_actor* a, const rela_ffnet* online, const uint8_t* obs_host,
                                   const float* eps_host, const float* legal_host, int64_t* action_host,
                                   const int64_t** action_dev_out, void* stream_) {
  RELA_CHECK(a && online, RELA_EINVAL, "rela_apex_actor_act: bad arguments");
  RELA_CHECK(rela_ffnet_num_action(online) == a->A, RELA_EINVAL, "rela_apex_actor_act: net has %d actions, actor %d",
             rela_ffnet_num_action(online), a->A);
  RELA_CHECK(a->count <= a->n, RELA_ESTATE, "rela_apex_actor_act: act() twice without post_step()");  // :24-25
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(a->device);
  const int slot = next_slot(a);
  uint8_t* obs = a->obs + (size_t)slot * a->R * kObs;
  if (obs_host) RELA_HIP(hipMemcpyAsync(obs, obs_host, (size_t)a->R * kObs, hipMemcpyHostToDevice, s));
  if (eps_host) RELA_HIP(hipMemcpyAsync(a->eps, eps_host, (size_t)a->R * sizeof(float), hipMemcpyHostToDevice, s));
  if (legal_host)
    RELA_HIP(hipMemcpyAsync(a->legal, legal_host, (size_t)a->R * a->A * sizeof(float), hipMemcpyHostToDevice, s));
  float* eps_s = a->eps_hist + (size_t)slot * a->R;
  float* legal_s = a->legal_hist + (size_t)slot * a->R * a->A;
  RELA_HIP(dev_copy2(eps_s, a->eps, (size_t)a->R * sizeof(float), legal_s, a->legal, (size_t)a->R * a->A * sizeof(float), s));
  float* q_s = a->q_hist + (size_t)slot * a->R * a->A;
  a->qh_net[slot] = nullptr;
  int rc = rela_ffnet_forward(online, a->R, obs, legal_s, q_s, a->ws, a->ws_bytes, s);
  if (rc != RELA_OK) return rc;
  int64_t* act = a->act + (size_t)slot * a->R;
  rc = rela_apex_act_from_q(a->R, a->A, a->K, q_s, legal_s, eps_s, a->seed, a->act_calls * (uint64_t)a->R, act, s);
  if (rc != RELA_OK) return rc;
  a->act_calls += 1;
  a->qh_net[slot] = online;
  a->qh_version[slot] = rela_ffnet_version(online);
  a->q_slot = slot;
  a->cur = slot;
  a->num_act += a->R;  // :169
  if (action_dev_out) *action_dev_out = act;
  if (action_host) {
    RELA_HIP(hipMemcpyAsync(action_host, act, (size_t)a->R * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    RELA_HIP(hipStreamSynchronize(s));
  }
  return RELA_OK;
}

extern "C" int rela_apex_actor_post_step(rela_apex_actor* a, const float* reward, const uint8_t* terminal,
                                         int on_device, const rela_ffnet* online, const rela_ffnet* target,
                                         int nonblocking, int* inserted, void* stream_) {
  RELA_CHECK(a && reward && terminal && online && target, RELA_EINVAL, "rela_apex_actor_post_step: bad arguments");
  RELA_CHECK(a->replay, RELA_ESTATE, "rela_apex_actor_post_step: evaluation actor has no replay");  // :175,182
  RELA_CHECK(a->cur >= 0, RELA_ESTATE, "rela_apex_actor_post_step: no act() to attach the reward to");  // :33
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(a->device);
  if (inserted) *inserted = 0;
  if (on_device) {  // one launch (common.h: dev_copy2), not two of the runtime's blit kernels
    RELA_HIP(dev_copy2(a->rew + (size_t)a->cur * a->R, reward, (size_t)a->R * sizeof(float), a->term + (size_t)a->cur * a->R, terminal,
                       (size_t)a->R, s));
  } else {
    RELA_HIP(hipMemcpyAsync(a->rew + (size_t)a->cur * a->R, reward, (size_t)a->R * sizeof(float), hipMemcpyHostToDevice, s));
    RELA_HIP(hipMemcpyAsync(a->term + (size_t)a->cur * a->R, terminal, (size_t)a->R, hipMemcpyHostToDevice, s));
  }
  const int H = a->n + 1;
  if (a->dd_ups > 0) {
    // de-duplicated replay: the stack acted on this tick enters the unit ring ONCE (one plane, or the whole
    // stack); transitions refer to it (as next_obs now, as obs n ticks from now)
    const int cur = a->cur, prev = (cur + H - 1) % H;
    const bool prev_ok = a->tick > 0 && a->refs_valid[prev];
    const int keyframe = (a->dd_ups == 4 && !prev_ok) ? 1 : 0;
    const int count = (a->dd_ups == 4 && keyframe) ? 4 * a->R : a->R;
    int64_t seq = 0;
    int32_t idx = 0;
    int rc = rela_replay_units_reserve(a->replay, count, nonblocking, &seq, &idx);
    if (rc == RELA_EWOULDBLOCK) {
      a->refs_valid[cur] = 0;  // ring full and non-blocking: this stack is not stored, its transitions are dropped
    } else {
      if (rc != RELA_OK) return rc;
      const uint8_t* stack = a->obs + (size_t)cur * a->R * kObs;
      if (a->dd_ups == 1) rc = rela_replay_units_write(a->replay, seq, count, stack, kObs, s);
      else if (keyframe) rc = rela_replay_units_write(a->replay, seq, count, stack, kPlane, s);
      else rc = rela_replay_units_write(a->replay, seq, count, stack + 3 * kPlane, kObs, s);  // the newest plane
      if (rc != RELA_OK) return rc;
      hipLaunchKernelGGL(dedup_make_refs, dim3(ceil_div(a->R, 256)), dim3(256), 0, s,
                         a->ref_hist + (size_t)cur * a->R * a->dd_ups, a->ref_hist + (size_t)prev * a->R * a->dd_ups,
                         a->term + (size_t)prev * a->R, a->R, a->dd_ups, keyframe, idx, a->dd_cap);
      RELA_LAUNCH_CHECK();
      a->refs_valid[cur] = 1;
      if (keyframe || a->dd_ups == 1) a->key_tick = a->tick;
      a->tick_seq[(size_t)(a->tick % kTickWin)] = seq;
    }
    a->tick += 1;
  }
  a->cur = -1;
  a->count += 1;
  if (a->count < a->n + 1) return RELA_OK;  // canPop :46-48
  const int first = a->head, last = (a->head + a->n) % H;
  int rc = rela_nstep_return(a->n, a->R, a->gamma, first, a->rew, a->term, a->out_r, a->out_b, a->out_t, s);
  if (rc != RELA_OK) return rc;
  // de-duplication: both stacks of the transition must be in the unit ring; the oldest unit it refers to is
  // the first plane of obs_t, stored at most 3 ticks before tick t = (tick - 1) - n (never before a keyframe)
  bool dd_drop = false;
  int64_t dd_min_seq = 0;
  if (a->dd_ups > 0) {
    dd_drop = !(a->refs_valid[first] && a->refs_valid[last]);
    int64_t t_first = a->tick - 1 - a->n;
    int64_t oldest = a->dd_ups == 4 ? t_first - 3 : t_first;
    if (oldest < 0) oldest = 0;
    // a keyframe at or before t_first bounds the chain; a later one cannot happen while refs stay valid
    if (a->key_tick >= 0 && a->key_tick <= t_first && oldest < a->key_tick) oldest = a->key_tick;
    if (a->tick - oldest >= kTickWin) oldest = a->tick - kTickWin + 1;
    dd_min_seq = a->tick_seq[(size_t)(oldest % kTickWin)];
  }
  const uint8_t* obs_t = a->obs + (size_t)first * a->R * kObs;
  const uint8_t* obs_n = a->obs + (size_t)last * a->R * kObs;
  const size_t QA = (size_t)a->R * a->A;
  const float* legal_t = a->legal_hist + (size_t)first * a->R * a->A;
  const float* legal_n = a->legal_hist + (size_t)last * a->R * a->A;
  const float* eps_t = a->eps_hist + (size_t)first * a->R;
  const float* eps_n = a->eps_hist + (size_t)last * a->R;
  // Both online forwards of compute_priority evaluate observations act() already ran the online net on:
  // obs is history.front(), acted on n ticks ago, and next_obs is history.back(), acted on this tick
  // (dqn_actor.h:84,161).  With the same weights (no load since: rela_ffnet_version), the same legal mask and the
  // same batch such a forward is bit-identical to the table act() left in q_hist[slot]: reuse it.
  const uint64_t ver = rela_ffnet_version(online);
  auto cached = [&](int slot) { return a->qh_net[slot] == online && a->qh_version[slot] == ver; };
  const float* q_online_t = a->q_hist + (size_t)first * QA;
  if (!(a->reuse_mode == 1 && cached(first))) {
    rc = rela_ffnet_forward(online, a->R, obs_t, legal_t, a->q + QA, a->ws, a->ws_bytes, s);  // apex.py:38
    if (rc != RELA_OK) return rc;
    q_online_t = a->q + QA;
  }
  const float* q_online_n = a->q_hist + (size_t)last * QA;  // greedy_act(next_obs) :41
  if (!(a->reuse_mode != 0 && cached(last) && a->q_slot == last)) {
    rc = rela_ffnet_forward(online, a->R, obs_n, legal_n, a->q + 2 * QA, a->ws, a->ws_bytes, s);
    if (rc != RELA_OK) return rc;
    q_online_n = a->q + 2 * QA;
  }
  rc = rela_ffnet_forward(target, a->R, obs_n, legal_n, a->q + 3 * QA, a->ws, a->ws_bytes, s);  // :42
  if (rc != RELA_OK) return rc;
  const int64_t* act_t = a->act + (size_t)first * a->R;
  rc = rela_apex_td_from_q(a->R, a->A, a->K, q_online_t, q_online_n, a->q + 3 * QA, legal_n, act_t, a->out_r,
                           a->out_b, a->gamma_n, nullptr, a->prio, s);
  if (rc != RELA_OK) return rc;
  // FFTransition rows (types.h:18-51): obs{s,eps,legal_move}, next_obs{...}, action{a}, reward, terminal, bootstrap
  const void* rows[10] = {obs_t, obs_n, eps_t, eps_n, legal_t, legal_n, act_t, a->out_r, a->out_t, a->out_b};
  if (a->dd_ups > 0) {
    rows[0] = a->ref_hist + (size_t)first * a->R * a->dd_ups;
    rows[1] = a->ref_hist + (size_t)last * a->R * a->dd_ups;
  }
  // One reference block per group of K rows (each batched actor thread's own add, :189).  The whole shard
  // is reserved at once when that can always be satisfied; a blocking append of more than ring - capacity
  // rows never can (sample() evicts down to capacity only), so a shard that large goes in pieces of whole
  // K-groups -- which is exactly what the reference's separate actor threads would issue.
  int cap = 0, ring = 0;
  rc = rela_replay_limits(a->replay, &cap, &ring);
  if (rc != RELA_OK) return rc;
  const int fit = ((ring - cap) / a->K) * a->K;
  const int piece = a->R <= ring - cap ? a->R : (fit > a->K ? fit : a->K);
  const int64_t stack_rb = a->dd_ups > 0 ? (int64_t)sizeof(int32_t) * a->dd_ups : kObs;
  const int64_t rb[10] = {stack_rb, stack_rb, 4, 4, 4 * a->A, 4 * a->A, 8, 4, 1, 4};
  int dropped = dd_drop ? 1 : 0;
  for (int off = 0; off < a->R && !dd_drop; off += piece) {
    const int cnt = a->R - off < piece ? a->R - off : piece;
    const void* prow[10];
    for (int f = 0; f < 10; ++f) prow[f] = static_cast<const uint8_t*>(rows[f]) + (int64_t)off * rb[f];
    int slot = 0;
    rc = rela_replay_begin_add(a->replay, cnt, nonblocking, &slot);
    if (rc == RELA_EWOULDBLOCK) {
      dropped = 1;
      continue;
    }
    if (rc != RELA_OK) break;
    if (a->dd_ups > 0) (void)rela_replay_set_block_min_unit(a->replay, slot, cnt, dd_min_seq);
    rc = rela_replay_write_rows(a->replay, slot, 0, cnt, prow, s);
    if (rc == RELA_OK) rc = rela_replay_commit_add_grouped(a->replay, slot, cnt, a->K, a->prio + off, s);
    if (rc != RELA_OK) {  // release the reservation so later blocks of other producers can still commit
      (void)rela_replay_abort_add(a->replay, slot, cnt);
      break;
    }
  }
  a->head = (a->head + 1) % H;  // pop_front :101-104
  a->count -= 1;
  if (rc == RELA_OK && dropped) rc = RELA_EWOULDBLOCK;
  if (rc == RELA_OK && inserted) *inserted = 1;
  return rc;
}
