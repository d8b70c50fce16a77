// Ape-X learner step on the device: pyrela/main.py:206-251 with ApexAgent.loss (apex.py:80-91),
// i.e.  sample -> td_err -> smooth_l1 * IS weight -> mean -> backward -> clip_grad_norm_ ->
// RMSprop -> (update_priority by the caller).  Replaces PyTorch autograd on the hot path
// (SURVEY 8f-2); AtariFFNet only (net.py:8-55).
//
//   forward   three rela_ffnet_forward calls (csrc/ffnet.hip); the online(obs) pass leaves its
//             activations a1/a2/a3/h in the workspace (ffnet_layout.h) for the backward pass
//   loss      rela_apex_td_from_q (signed TD error, batch-global q.min() as in apex.py:51) +
//             learner_loss_grad: Huber' * w / B pushed through the dueling head -> d_ha [B][32]
//   backward  every contraction is one instance of gemm_lds (both operands staged through LDS,
//             v_mfma_f32_16x16x4_f32, f32 throughout):
//               dgrad  d_h   = d_ha  x Wh            (relu mask in the epilogue)
//                      d_a3  = d_h   x Wfc'          (Wfc' = fc weight in channel-last k order)
//                      col3  = d_a3' x W3'  -> col2im3 -> d_a2     (d_a3' = rows (b,pos), 64 channels)
//                      col2  = d_a2' x W2'  -> col2im2 -> d_a1
//               wgrad  dWh   = d_ha^T x h,  dWfc = d_h^T x a3,
//                      dW3 = d_a3'^T x im2col(a2), dW2 = d_a2'^T x im2col(a1), dW1 = d_a1'^T x im2col(s)
//                      (im2col is index arithmetic in the B loader; the long reductions over
//                      B*pos are split over blockIdx.z into partial tiles that reduce_splits sums
//                      in a fixed order -- deterministic, no atomics)
//             bias gradients are column sums (colsum_*).
//   update    one flat parameter / gradient / optimiser-state buffer in state_dict order:
//             global 2-norm -> clip coefficient -> RMSprop (torch defaults alpha = 0.99), then the
//             kernel-layout copies of the new weights are re-packed.
// The gradient buffer is exposed (rela_apex_learner_flat) so data-parallel learners all-reduce
// it between rela_apex_learner_backward and rela_apex_learner_apply.
#include "learner_common.h"

using namespace rela_amd;

// flat parameter layout: rela_ffnet_params order, every segment padded to 4 floats
struct rela_apex_learner {
  int device = 0;
  int A = 0, Bmax = 0;
  float gamma_n = 0.f;
  int optimizer = 0;  // 0 RMSprop, 1 Adam
  float lr = 0.f, opt_eps = 0.f, clip = 0.f;
  int64_t adam_t = 0;
  int64_t off[13] = {0};  // segment offsets, off[12] = total
  float *P = nullptr, *PT = nullptr, *G = nullptr, *S1 = nullptr, *S2 = nullptr;
  rela_ffnet *online = nullptr, *target = nullptr;
  float *w2p = nullptr, *w3p = nullptr, *wfcp = nullptr;  // dgrad operand copies
  void *ws_on = nullptr, *ws_tmp = nullptr;
  int64_t ws_bytes = 0;
  float *q = nullptr;  // [3][B][A]
  float *td = nullptr, *d_ha = nullptr, *d_h = nullptr, *d_a3 = nullptr, *d_a2 = nullptr, *d_a1 = nullptr;
  float *col = nullptr;    // max(B*81*512, B*49*576)
  float *part = nullptr;   // split-K partial tiles
  float *cpart = nullptr;  // colsum partials [64][512]
  float *s32 = nullptr;
  double* npart = nullptr;
  float* norm = nullptr;  // [0] grad norm, [1] clip coefficient
  float* loss = nullptr;
  bool loaded = false;
  // batch of the last rela_apex_learner_loss, until rela_apex_learner_grad consumes it
  int pend_B = 0, last_B = 0;
  int pend_rows = 0;  // rows of the ffnet_ws layout the last forward used for ws_on (B, or 2 B for the merged forward)
  const uint8_t* pend_obs = nullptr;
  // The side lane of the backward pass (TrunkBwd in learner_common.h): the weight gradients of heads, fc, conv3 and
  // conv2, the column sums and the records -> f32 conversion of the forward's activations run on `side` next to the
  // data-gradient chain on the caller's stream; RELA_LEARNER_LANES=1 keeps everything on the caller's stream.
  hipStream_t side = nullptr;
  hipEvent_t ev[8] = {nullptr};  // 0 forward done | 1 unsplit done | 2 d_ha | 3 d_h | 4 d_a3 | 5 d_a2 | 6 side lane done
  float *part_side = nullptr, *cpart_side = nullptr;
  void *frag2 = nullptr, *frag3 = nullptr;  // conv2 / conv3 data-gradient weight fragments, re-packed with the weights
};

namespace {
rela_ffnet_params params_at(const rela_apex_learner* l, float* base) {
  rela_ffnet_params p;
  const float** f = reinterpret_cast<const float**>(&p);
  for (int i = 0; i < 12; ++i) f[i] = base + l->off[i];
  return p;
}

int repack(rela_apex_learner* l, bool online, bool target, hipStream_t s) {
  if (online) {
    const rela_ffnet_params p = params_at(l, l->P);
    FFNetExtraPacks extra;  // the dgrad operand copies ride along in the re-pack launch
    extra.w2p = l->w2p, extra.w3p = l->w3p, extra.wfcp = l->wfcp;
    int rc = ffnet_load_extra(l->online, &p, s, extra);
    if (rc != RELA_OK) return rc;
    if (l->frag2) dgfast::pack_frags(l->w2p, l->w3p, l->frag2, l->frag3, s);
  }
  if (target) {
    const rela_ffnet_params p = params_at(l, l->PT);
    int rc = rela_ffnet_load(l->target, &p, 1, s);
    if (rc != RELA_OK) return rc;
  }
  return RELA_OK;
}
}  // namespace

extern "C" int rela_apex_learner_create(rela_apex_learner** out, int num_action, int max_batch, int multi_step,
                                        float gamma, int optimizer, float lr, float eps, float grad_clip,
                                        int device) {
  RELA_CHECK(out && num_action >= 1 && num_action <= 31 && max_batch >= 1 && multi_step >= 1 &&
                 (optimizer == 0 || optimizer == 1),
             RELA_EINVAL, "rela_apex_learner_create: bad arguments (A=%d batch=%d n=%d optimizer=%d)", num_action,
             max_batch, multi_step, optimizer);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_last_error("rela_apex_learner_create: HIP device %d not available (%d visible); there is no CPU path",
                   device, ndev);
    return RELA_ENODEV;
  }
  DeviceGuard g(device);
  auto* l = new rela_apex_learner();
  l->device = device;
  l->A = num_action;
  l->Bmax = max_batch;
  l->gamma_n = (float)pow((double)gamma, (double)multi_step);  // apex.py:44
  l->optimizer = optimizer;
  l->lr = lr;
  l->opt_eps = eps;
  l->clip = grad_clip;
  const int64_t cnt[12] = {32 * 256, 32, 64 * 512, 64, 64 * 576, 64, (int64_t)512 * 3136, 512, 512, 1,
                           (int64_t)num_action * 512, num_action};
  for (int i = 0; i < 12; ++i) l->off[i + 1] = l->off[i] + (cnt[i] + 3) / 4 * 4;
  const size_t nb = sizeof(float) * (size_t)l->off[12];
  const size_t B = (size_t)max_batch, A = (size_t)num_action;
  RELA_HIP(hipMalloc(&l->P, nb));
  RELA_HIP(hipMalloc(&l->PT, nb));
  RELA_HIP(hipMalloc(&l->G, nb));
  RELA_HIP(hipMalloc(&l->S1, nb));
  RELA_HIP(hipMalloc(&l->S2, nb));
  RELA_HIP(hipMemset(l->P, 0, nb));
  RELA_HIP(hipMemset(l->PT, 0, nb));
  RELA_HIP(hipMemset(l->G, 0, nb));
  RELA_HIP(hipMemset(l->S1, 0, nb));
  RELA_HIP(hipMemset(l->S2, 0, nb));
  int rc = rela_ffnet_create(&l->online, num_action, device);
  if (rc != RELA_OK) return rc;
  rc = rela_ffnet_create(&l->target, num_action, device);
  if (rc != RELA_OK) return rc;
  ffnet_label_as_learner(l->online);
  ffnet_label_as_learner(l->target);
  ffnet_set_max_rows(l->online, max_batch);
  ffnet_set_max_rows(l->target, max_batch);
  RELA_HIP(hipMalloc(&l->w2p, sizeof(float) * 64 * 512));
  RELA_HIP(hipMalloc(&l->w3p, sizeof(float) * 64 * 576));
  RELA_HIP(hipMalloc(&l->wfcp, sizeof(float) * 512 * 3136));
  // (the merged split-bf16 forward runs the online net over 2 x batch rows: [s ; s'])
  l->ws_bytes = std::max(rela_ffnet_workspace_bytes(nullptr, max_batch), rela_ffnet_workspace_bytes(nullptr, 2 * max_batch));
  RELA_HIP(hipMalloc(&l->ws_on, (size_t)l->ws_bytes));
  RELA_HIP(hipMalloc(&l->ws_tmp, (size_t)l->ws_bytes));
  RELA_HIP(hipMalloc(&l->q, sizeof(float) * 3 * B * A));
  RELA_HIP(hipMalloc(&l->td, sizeof(float) * B));
  RELA_HIP(hipMalloc(&l->d_ha, sizeof(float) * B * 32));
  RELA_HIP(hipMalloc(&l->d_h, sizeof(float) * B * 512));
  RELA_HIP(hipMalloc(&l->d_a3, sizeof(float) * B * kA3));
  RELA_HIP(hipMalloc(&l->d_a2, sizeof(float) * B * kA2));
  RELA_HIP(hipMalloc(&l->d_a1, sizeof(float) * B * kA1));
  RELA_HIP(hipMalloc(&l->col, sizeof(float) * trunk_col_floats(B)));
  RELA_HIP(hipMalloc(&l->part, sizeof(float) * kTrunkPartFloats));
  RELA_HIP(hipMalloc(&l->cpart, sizeof(float) * kColsumBlocks * (32 + 512 + 64 + 64 + 32)));  // all five jobs
  RELA_HIP(hipMalloc(&l->s32, sizeof(float) * 32));
  RELA_HIP(hipMalloc(&l->npart, sizeof(double) * kNormBlocks));
  RELA_HIP(hipMalloc(&l->norm, sizeof(float) * 2));
  RELA_HIP(hipMalloc(&l->loss, sizeof(float)));
  static const bool lanes = !(getenv("RELA_LEARNER_LANES") && atoi(getenv("RELA_LEARNER_LANES")) == 1);
  if (lanes) {
    RELA_HIP(hipStreamCreateWithFlags(&l->side, hipStreamNonBlocking));
    for (hipEvent_t& e : l->ev) RELA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    RELA_HIP(hipMalloc(&l->part_side, sizeof(float) * kTrunkPartFloats));
    RELA_HIP(hipMalloc(&l->cpart_side, sizeof(float) * kColsumBlocks * (32 + 512 + 64 + 64 + 32)));
    RELA_HIP(hipMalloc(&l->frag2, dgfast::kFrag2Bytes));
    RELA_HIP(hipMalloc(&l->frag3, dgfast::kFrag3Bytes));
  }
  *out = l;
  return RELA_OK;
}

extern "C" void rela_apex_learner_destroy(rela_apex_learner* l) {
  if (!l) return;
  DeviceGuard g(l->device);
  (void)hipDeviceSynchronize();
  void* ps[] = {l->P,  l->PT,   l->G,    l->S1,   l->S2,   l->w2p, l->w3p,  l->wfcp,  l->ws_on, l->ws_tmp, l->q,
                l->td, l->d_ha, l->d_h,  l->d_a3, l->d_a2, l->d_a1, l->col, l->part,  l->cpart, l->s32,    l->npart,
                l->norm, l->loss, l->part_side, l->cpart_side, l->frag2, l->frag3};
  for (void* p : ps) (void)hipFree(p);
  for (hipEvent_t e : l->ev)
    if (e) (void)hipEventDestroy(e);
  if (l->side) (void)hipStreamDestroy(l->side);
  rela_ffnet_destroy(l->online);
  rela_ffnet_destroy(l->target);
  delete l;
}

extern "C" int rela_apex_learner_load(rela_apex_learner* l, const rela_ffnet_params* online,
                                      const rela_ffnet_params* target, int on_device, void* stream_) {
  RELA_CHECK(l && online, RELA_EINVAL, "rela_apex_learner_load: bad arguments");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(l->device);
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  const float* const* fo = reinterpret_cast<const float* const*>(online);
  const float* const* ft = reinterpret_cast<const float* const*>(target ? target : online);
  const int64_t cnt[12] = {32 * 256, 32, 64 * 512, 64, 64 * 576, 64, (int64_t)512 * 3136, 512, 512, 1,
                           (int64_t)l->A * 512, l->A};
  for (int i = 0; i < 12; ++i) {
    RELA_CHECK(fo[i] && ft[i], RELA_EINVAL, "rela_apex_learner_load: parameter %d is NULL", i);
    RELA_HIP(hipMemcpyAsync(l->P + l->off[i], fo[i], sizeof(float) * cnt[i], kind, s));
    RELA_HIP(hipMemcpyAsync(l->PT + l->off[i], ft[i], sizeof(float) * cnt[i], kind, s));
  }
  if (!on_device) RELA_HIP(hipStreamSynchronize(s));  // the host buffers may go away
  const size_t nb = sizeof(float) * (size_t)l->off[12];
  RELA_HIP(hipMemsetAsync(l->S1, 0, nb, s));
  RELA_HIP(hipMemsetAsync(l->S2, 0, nb, s));
  l->adam_t = 0;
  int rc = repack(l, true, true, s);
  if (rc != RELA_OK) return rc;
  l->loaded = true;
  return RELA_OK;
}

extern "C" int rela_apex_learner_set_precision(rela_apex_learner* l, int mode) {
  RELA_CHECK(l && mode >= 0 && mode <= 2, RELA_EINVAL, "rela_apex_learner_set_precision: mode must be 0, 1 or 2");
  int rc = rela_ffnet_set_precision(l->online, mode);
  if (rc != RELA_OK) return rc;
  return rela_ffnet_set_precision(l->target, mode);
}

extern "C" int rela_apex_learner_sync_target(rela_apex_learner* l, void* stream_) {
  RELA_CHECK(l && l->loaded, RELA_ESTATE, "rela_apex_learner_sync_target: parameters were never loaded");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(l->device);
  RELA_HIP(hipMemcpyAsync(l->PT, l->P, sizeof(float) * (size_t)l->off[12], hipMemcpyDeviceToDevice, s));  // apex.py:27
  return repack(l, false, true, s);
}

extern "C" int rela_apex_learner_params(rela_apex_learner* l, rela_ffnet_params* online_out,
                                        rela_ffnet_params* target_out) {
  RELA_CHECK(l, RELA_EINVAL, "rela_apex_learner_params: bad arguments");
  if (online_out) *online_out = params_at(l, l->P);
  if (target_out) *target_out = params_at(l, l->PT);
  return RELA_OK;
}

extern "C" int rela_apex_learner_grads(rela_apex_learner* l, rela_ffnet_params* grads_out) {
  RELA_CHECK(l && grads_out, RELA_EINVAL, "rela_apex_learner_grads: bad arguments");
  *grads_out = params_at(l, l->G);
  return RELA_OK;
}

extern "C" int rela_apex_learner_flat(rela_apex_learner* l, float** params_dev, float** grads_dev, int64_t* count) {
  RELA_CHECK(l, RELA_EINVAL, "rela_apex_learner_flat: bad arguments");
  if (params_dev) *params_dev = l->P;
  if (grads_dev) *grads_dev = l->G;
  if (count) *count = l->off[12];
  return RELA_OK;
}

extern "C" const float* rela_apex_learner_stats_dev(const rela_apex_learner* l) { return l ? l->norm : nullptr; }

extern "C" int rela_apex_learner_debug_activations(rela_apex_learner* l, float** a1, float** a2, float** a3, float** h,
                                                   int* batch) {
  RELA_CHECK(l && l->last_B > 0, RELA_ESTATE, "rela_apex_learner_debug_activations: no rela_apex_learner_loss yet");
  const FFNetWs w = ffnet_ws(l->ws_on, l->pend_rows);
  if (a1) *a1 = w.a1;
  if (a2) *a2 = w.a2;
  if (a3) *a3 = w.a3;
  if (h) *h = w.h;
  if (batch) *batch = l->last_B;
  return RELA_OK;
}

extern "C" int rela_apex_learner_backward(rela_apex_learner* l, int batch, const void* const* rows_dev,
                                          const float* weight_dev, float* priority_dev, float* loss_dev,
                                          void* stream_) {
  int rc = rela_apex_learner_loss(l, batch, rows_dev, weight_dev, priority_dev, loss_dev, stream_);
  if (rc != RELA_OK) return rc;
  return rela_apex_learner_grad(l, stream_);
}

// The forward half of the step: the three forwards of td_err, the priorities and the loss (with the head gradient
// d_ha the loss kernel leaves).  `priority_dev` is final when this returns' work is done, so a caller may hand it to
// update_priority and ask for the next batch while rela_apex_learner_grad still runs (the replay is not touched by
// the backward pass); the batch's `s` rows must stay untouched until then (conv1's weight gradient reads them).
// RELA_LEARNER_MERGE_ONLINE=0: the two online forwards of the f32x3 step as two launches per layer again (A/B switch)
static bool merge_online_forwards() {
  static const bool on = [] {
    const char* e = getenv("RELA_LEARNER_MERGE_ONLINE");
    return !(e && e[0] == '0');
  }();
  return on;
}

extern "C" int rela_apex_learner_loss(rela_apex_learner* l, int batch, const void* const* rows_dev,
                                      const float* weight_dev, float* priority_dev, float* loss_dev, void* stream_) {
  RELA_CHECK(l && l->loaded, RELA_ESTATE, "rela_apex_learner_loss: parameters were never loaded");
  RELA_CHECK(batch >= 1 && batch <= l->Bmax && rows_dev && weight_dev && priority_dev, RELA_EINVAL,
             "rela_apex_learner_loss: bad arguments (batch %d, max %d)", batch, l->Bmax);
  l->pend_B = 0;
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(l->device);
  const int Bn = batch, A = l->A;
  // FFTransition fields in the order rela_replay_sample fills them (types.h:18-51)
  const uint8_t* obs = static_cast<const uint8_t*>(rows_dev[0]);
  const uint8_t* nobs = static_cast<const uint8_t*>(rows_dev[1]);
  const float* legal = static_cast<const float*>(rows_dev[4]);
  const float* nlegal = static_cast<const float*>(rows_dev[5]);
  const int64_t* act = static_cast<const int64_t*>(rows_dev[6]);
  const float* reward = static_cast<const float*>(rows_dev[7]);
  const float* boot = static_cast<const float*>(rows_dev[9]);
  RELA_CHECK(obs && nobs && legal && nlegal && act && reward && boot, RELA_EINVAL,
             "rela_apex_learner_loss: a batch field is NULL");
  float* q_on = l->q;
  float* q_no = l->q + (size_t)Bn * A;
  float* q_nt = l->q + 2 * (size_t)Bn * A;
  // td_err (apex.py:30-45): greedy_act(next_obs) and target_net(next_obs) carry no gradient
  int rc;
  bool unsplit_side = false;
  l->pend_rows = Bn;
  if (rela_ffnet_precision(l->online) == 1 && ffnet_learner_forward_ok(l->online, l->target, Bn)) {
    // bf16x2: all three forwards on split-bf16 MFMA, one launch per layer (online over [s ; s'], target over s'); the
    // backward then reads the activations -- and the ReLU masks -- of THIS forward (turned back into f32 below)
    rc = ffnet_learner_forward(l->online, l->target, Bn, obs, nobs, legal, nlegal, q_on, q_no, q_nt, l->ws_on, l->ws_tmp,
                               l->ws_bytes, s);
    if (rc != RELA_OK) return rc;
    if (l->side) lane_dep(l->ev[0], s, l->side);  // (the loss kernel below reads Q only: the conversion runs next to it)
    rc = ffnet_learner_unsplit(Bn, l->ws_on, l->side ? l->side : s);
    if (rc != RELA_OK) return rc;
    unsplit_side = l->side != nullptr;
    l->pend_rows = 2 * Bn;  // the workspace layout the backward must address
  } else if (rela_ffnet_precision(l->online) == 2 && Bn >= 512 && merge_online_forwards() &&
             nobs == obs + (size_t)Bn * 28224 && nlegal == legal + (size_t)Bn * A) {
    // f32x3 (r5): the two ONLINE forwards as one launch per layer over [s ; s'] when the caller's batch has s' right behind s
    // and the legal moves likewise (rela_amd.replay.FFReplay's output buffers do).  Same-box A/B at B = 512: 2.033 against
    // 2.045 ms per bench step, 0.729 against 0.733 ms learner-only (profiles/r05_ab_merge_online.log) -- half a percent;
    // a caller with separate buffers keeps the three launches (staging copies would cost more than that).
    const uint8_t* in2 = obs;
    const float* legal2 = legal;
    rc = rela_ffnet_forward(l->target, Bn, nobs, nlegal, q_nt, l->ws_tmp, l->ws_bytes, s);
    if (rc != RELA_OK) return rc;
    // (the net's declared batch limit only says which weight layouts its loads pack; the three-part ones this launch reads
    // are packed from 512 rows up, and the workspaces are sized for 2 x the batch)
    ffnet_set_max_rows(l->online, 2 * l->Bmax);
    rc = ffnet_forward_mode(l->online, 2 * Bn, in2, legal2, q_on, l->ws_on, l->ws_bytes, s, 3);  // q_no = q_on + Bn * A
    ffnet_set_max_rows(l->online, l->Bmax);
    if (rc != RELA_OK) return rc;
    l->pend_rows = 2 * Bn;  // the workspace layout the backward must address (it reads the first Bn rows)
  } else {
    rc = rela_ffnet_forward(l->online, Bn, nobs, nlegal, q_no, l->ws_tmp, l->ws_bytes, s);
    if (rc != RELA_OK) return rc;
    rc = rela_ffnet_forward(l->target, Bn, nobs, nlegal, q_nt, l->ws_tmp, l->ws_bytes, s);
    if (rc != RELA_OK) return rc;
    // f32 activations for the backward, which reads a1..h (the f32x3 mode keeps that layout: it may serve this pass too)
    rc = ffnet_forward_mode(l->online, Bn, obs, legal, q_on, l->ws_on, l->ws_bytes, s, rela_ffnet_precision(l->online) == 2 ? 3 : 0);
    if (rc != RELA_OK) return rc;
  }
  if (Bn <= 1024) {
    ProfScope prof("learner_loss_grad", s);
    hipLaunchKernelGGL(learner_td_loss_grad, dim3(1), dim3(1024), 0, s, Bn, A, (const float*)q_on, (const float*)q_no,
                       (const float*)q_nt, nlegal, act, reward, boot, l->gamma_n, weight_dev, legal, l->td, priority_dev,
                       l->d_ha, l->loss);
  } else {
    rc = rela_apex_td_from_q(Bn, A, 0, q_on, q_no, q_nt, nlegal, act, reward, boot, l->gamma_n, l->td, priority_dev, s);
    if (rc != RELA_OK) return rc;
    ProfScope prof("learner_loss_grad", s);
    hipLaunchKernelGGL(learner_loss_grad, dim3(1), dim3(kLT), 0, s, (const float*)l->td, weight_dev, act, legal, Bn,
                       A, l->d_ha, l->loss);
  }
  if (loss_dev) RELA_HIP(dev_copy2(loss_dev, l->loss, sizeof(float), nullptr, nullptr, 0, s));  // (a kernel, not a blit: common.h)
  if (unsplit_side) lane_dep(l->ev[1], l->side, s);  // the caller's stream owns the workspace again from here
  RELA_LAUNCH_CHECK();
  l->pend_B = l->last_B = Bn;
  l->pend_obs = obs;
  return RELA_OK;
}

// The backward half: gradients of the last rela_apex_learner_loss into the flat gradient buffer.
extern "C" int rela_apex_learner_grad(rela_apex_learner* l, void* stream_) {
  RELA_CHECK(l && l->loaded && l->pend_B > 0, RELA_ESTATE, "rela_apex_learner_grad: no rela_apex_learner_loss to differentiate");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(l->device);
  const int Bn = l->pend_B, A = l->A;
  const uint8_t* obs = l->pend_obs;
  l->pend_B = 0;

  const FFNetWs w = ffnet_ws(l->ws_on, l->pend_rows);
  const rela_ffnet_params P = params_at(l, l->P);
  float* Gm[12];  // gradient tensors in rela_ffnet_params order
  for (int i = 0; i < 12; ++i) Gm[i] = l->G + l->off[i];

  ColsumJobs sums;  // the five bias gradients: queued here, one launch pair at the end of trunk_backward
  const bool g3 = rela_ffnet_precision(l->online) == 1 && gemm_bf16x3_on();  // bf16x2: the GEMMs on bf16 MFMA too
  const bool g6 = rela_ffnet_precision(l->online) == 2;  // f32x3: ... with three-part operands (f32 accuracy)
  const bool lanes = l->side != nullptr;
  hipStream_t sw = lanes ? l->side : s;  // the weight-gradient lane
  if (lanes) lane_dep(l->ev[2], s, sw);  // d_ha and the forward's activations are ready

  // heads: d_h, dWh, db
  {
    ProbHeadDgrad p{};
    p.M = Bn, p.N = 512, p.K = 32;
    p.d_ha = l->d_ha, p.a_w = P.a_w, p.v_w = P.v_w, p.h = w.h, p.d_h = l->d_h, p.A = A;
    if (g3) (void)gemm3::launch_gemm<Tile3Dgrad>(p, 1, s, "learner_dgrad_heads");
    else if (g6) (void)gemm3::launch_gemm<Tile6Dgrad>(p, 1, s, "learner_dgrad_heads");
    else launch_gemm<TileDgrad>(p, 1, s, "learner_dgrad_heads");
  }
  {
    ProbHeadWgrad p{};
    p.M = 32, p.N = 512, p.K = Bn;
    p.d_ha = l->d_ha, p.h = w.h, p.g_a_w = Gm[10], p.g_v_w = Gm[8], p.A = A;
    if (g3) (void)gemm3::launch_gemm<Tile3W32>(p, 1, sw, "learner_wgrad_heads");
    else if (g6) (void)gemm3::launch_gemm<Tile6W32>(p, 1, sw, "learner_wgrad_heads");
    else launch_gemm<TileW32>(p, 1, sw, "learner_wgrad_heads");
  }
  sums.add(l->d_ha, Bn, 32, l->s32);
  if (lanes) lane_dep(l->ev[3], s, sw);  // d_h is ready
  // fc: d_a3, dWfc, db
  {
    ProbFcDgrad p{};
    p.M = Bn, p.N = 3136, p.K = 512;
    p.d_h = l->d_h, p.wfcp = l->wfcp, p.a3 = w.a3, p.d_a3 = l->d_a3;
    if (g3) (void)gemm3::launch_gemm<Tile3Dgrad>(p, 1, s, "learner_dgrad_fc");
    else if (g6) (void)gemm3::launch_gemm<Tile6Dgrad>(p, 1, s, "learner_dgrad_fc");
    else launch_gemm<TileDgrad>(p, 1, s, "learner_dgrad_fc");
  }
  {
    ProbFcWgrad p{};
    p.M = 512, p.N = 3136, p.K = Bn;
    p.d_h = l->d_h, p.a3 = w.a3, p.g_fc_w = Gm[6];
    if (g3) (void)gemm3::launch_gemm<Tile3Wfc>(p, 1, sw, "learner_wgrad_fc");
    else if (g6) (void)gemm3::launch_gemm<Tile6Wfc>(p, 1, sw, "learner_wgrad_fc");
    else launch_gemm<TileWfc>(p, 1, sw, "learner_wgrad_fc");
  }
  sums.add(l->d_h, Bn, 512, Gm[7]);
  {
    TrunkBwd t{};
    t.Bn = Bn, t.obs = obs, t.a1 = w.a1, t.a2 = w.a2, t.d_a3 = l->d_a3, t.d_a2 = l->d_a2, t.d_a1 = l->d_a1;
    t.col = l->col, t.part = l->part, t.cpart = l->cpart, t.w2p = l->w2p, t.w3p = l->w3p;
    t.g_c1w = Gm[0], t.g_c1b = Gm[1], t.g_c2w = Gm[2], t.g_c2b = Gm[3], t.g_c3w = Gm[4], t.g_c3b = Gm[5];
    t.fast = rela_ffnet_precision(l->online) == 1;
    t.emu = g6;
    if (lanes) {
      t.side = sw, t.ev_da3 = l->ev[4], t.ev_da2 = l->ev[5], t.ev_side = l->ev[6];
      t.part_side = l->part_side, t.cpart_side = l->cpart_side;
      if (t.fast) t.frag2 = l->frag2, t.frag3 = l->frag3;
      t.s32 = l->s32, t.A = A, t.g_a_b = Gm[11], t.g_v_b = Gm[9];
    }
    trunk_backward(t, s, &sums);
    if (!lanes) hipLaunchKernelGGL(head_bias_grad, dim3(1), dim3(32), 0, s, (const float*)l->s32, A, Gm[11], Gm[9]);
  }
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

extern "C" int rela_apex_learner_apply(rela_apex_learner* l, void* stream_) {
  RELA_CHECK(l && l->loaded, RELA_ESTATE, "rela_apex_learner_apply: parameters were never loaded");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(l->device);
  OptimState o;
  o.optimizer = l->optimizer, o.lr = l->lr, o.eps = l->opt_eps, o.clip = l->clip, o.adam_t = l->adam_t;
  optimizer_apply(o, l->P, l->G, l->S1, l->S2, l->off[12], l->npart, l->norm, s);
  l->adam_t = o.adam_t;
  RELA_LAUNCH_CHECK();
  return repack(l, true, false, s);
}
