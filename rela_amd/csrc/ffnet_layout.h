// Workspace layout of rela_ffnet_forward (csrc/ffnet.hip), shared with the learner's backward pass
// (csrc/learner.hip), which reads the activations the forward left there.
//   a1 [N][20*20][32]  relu(conv1)   channel-last
//   a2 [N][ 9* 9][64]  relu(conv2)
//   a3 [N][ 7* 7][64]  relu(conv3)   (= fc input, k = pos*64 + c)
//   h  [N][512]        relu(fc)
//   ha [N][32]         heads: columns 0..A-1 = fc_a, column 31 = fc_v
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rela_amd {

constexpr int64_t kA1 = 400 * 32, kA2 = 81 * 64, kA3 = 49 * 64, kH = 512, kHA = 32;
constexpr int64_t kWsFloatsPerSample = kA1 + kA2 + kA3 + kH + kHA;
// split3 records of a2 / a3 (f32x3 mode, gemm_s3.h): 6 bytes per value where the f32 tensors have 4
constexpr int64_t kRec2Bytes = 81 * 384, kRec3Bytes = 49 * 384;

struct FFNetWs {
  float *a1, *a2, *a3, *h, *ha;
};

inline FFNetWs ffnet_ws(void* ws, int N) {
  FFNetWs w;
  w.a1 = static_cast<float*>(ws);
  w.a2 = w.a1 + kA1 * N;
  w.a3 = w.a2 + kA2 * N;
  w.h = w.a3 + kA3 * N;
  w.ha = w.h + kH * N;
  return w;
}

}  // namespace rela_amd

namespace rela_amd {
// weight copies in the k order the learner's dgrad GEMMs read (one element per call):
//   conv2: dst[oc][(kh*4+kw)*32+c], conv3: dst[oc][(kh*3+kw)*64+c], fc: dst[u][pos*64+c] <- src[u][c*49+pos]
enum { kPermConv2 = 0, kPermConv3 = 1, kPermFc = 2 };
__device__ __forceinline__ void permute_weight_at(int mode, int idx, const float* __restrict__ src, float* __restrict__ dst) {
  if (mode == kPermConv2) {
    const int oc = idx >> 9, n = idx & 511, c = n & 31, r = n >> 5;
    dst[idx] = src[((oc * 32 + c) * 4 + (r >> 2)) * 4 + (r & 3)];
  } else if (mode == kPermConv3) {
    const int oc = idx / 576, n = idx - oc * 576, c = n & 63, r = n >> 6;
    dst[idx] = src[((oc * 64 + c) * 3 + r / 3) * 3 + r % 3];
  } else {
    const int u = idx / 3136, n = idx - u * 3136, c = n & 63, pos = n >> 6;
    dst[idx] = src[(size_t)u * 3136 + c * 49 + pos];
  }
}
// the three copies above, made by the SAME launch that re-packs the net's own layouts (a learner re-packs after
// every optimiser step: three more launches otherwise); any pointer may be NULL
struct FFNetExtraPacks {
  float *w2p = nullptr, *w3p = nullptr, *wfcp = nullptr;
};
}  // namespace rela_amd

#include "../../include/rela_amd.h"
namespace rela_amd {
// rela_ffnet_load from device pointers + the learner's dgrad operand copies in the same launch
int ffnet_load_extra(rela_ffnet* n, const rela_ffnet_params* p, void* stream, const FFNetExtraPacks& extra);
// per-kernel timing labels "learner_fwd_*" instead of the actor-side names (prof.h)
void ffnet_label_as_learner(rela_ffnet* n);
// the owner never runs more than `rows` rows through this net: rela_ffnet_load skips the layouts only larger batches read
void ffnet_set_max_rows(rela_ffnet* n, int rows);
// rela_ffnet_forward with the precision chosen by the caller: mode -1 = the net's own (rela_ffnet_set_precision),
// 0 = exact f32 whatever the net says -- the learner's pass that keeps a1 / a2 / a3 / h for the backward kernels
int ffnet_forward_mode(const rela_ffnet* n, int N, const uint8_t* s_dev, const float* legal_dev, float* q_dev, void* ws,
                       int64_t ws_bytes, void* stream, int mode);
// The Ape-X learner's three forwards of td_err (apex.py:30-45) in split-bf16 with one launch per layer: online over
// [s ; s'] (2 B rows) and target over s' (B rows).  ws_on: ffnet_ws layout for 2 B rows, ws_tg for B rows; a1 (rows < B),
// a2, a3 come out as split records; ffnet_learner_unsplit turns the rows < B into f32 in place for the backward pass.
bool ffnet_learner_forward_ok(const rela_ffnet* on, const rela_ffnet* tg, int B);
int ffnet_learner_forward(const rela_ffnet* on, const rela_ffnet* tg, int B, const uint8_t* s_obs, const uint8_t* s_next,
                          const float* legal, const float* nlegal, float* q_on, float* q_no, float* q_nt, void* ws_on,
                          void* ws_tg, int64_t ws_bytes, hipStream_t s);
int ffnet_learner_unsplit(int B, void* ws_on, hipStream_t s);
}  // namespace rela_amd

struct rela_lstmnet;
namespace rela_amd {
// internal (not part of the C ABI): the conv trunk and the dueling heads of an AtariLSTMNet on their own,
// for the R2D2 learner, which batches the trunk over all T*B frames of a sequence batch and runs the
// recurrent part itself.  names: three per-kernel timing labels.
// fast: conv1 -> conv2 fused and conv3 on split-bf16 MFMA (a3 comes out in f32 as always; a1 is NOT produced and a2 holds
// split records, so only for passes whose activations nobody reads back)
// a3_records != NULL (with fast): a3 may stay in split records [rows][49][64 hi | 64 lo] (*a3_records says whether)
// s3_scratch != NULL (f32x3 nets, N >= 512): N * (kRec2Bytes + kRec3Bytes) bytes -- the trunk runs on split3 records
// (conv12_s3 -> conv3_img_s3), a3's records stay at s3_scratch + N * kRec2Bytes (*a3_records says whether) and a1 / a2 /
// a3 are written as f32 only if keep_f32
int lstmnet_trunk(const rela_lstmnet* n, int N, const uint8_t* s_dev, float* a1, float* a2, float* a3, hipStream_t s,
                  const char* const* names, bool fast = false, bool* a3_records = nullptr, uint8_t* s3_scratch = nullptr,
                  bool keep_f32 = true);
// weight_ih_l0 (2048, 3136; state_dict layout) -> the three-part bf16 fragments gemm_s3<ProbFcT<2048>> reads
// (gate_x3_packed_bytes() bytes); permuted: gate columns as 4 * unit + gate (the actors' fused cell) or as stored
int64_t gate_x3_packed_bytes();
int pack_gate_x3(const float* w_ih_dev, void* dst, bool permuted, hipStream_t s);
// the x part of the gates over a3's records: gx[M][2048] = bias + a3 . W_ih^T (bias may not be NULL)
int gate_x3_gemm(const uint8_t* a3_records, const void* packed, const float* bias, float* gx, int M, hipStream_t s);
int lstmnet_heads(const rela_lstmnet* n, int N, const float* o, const float* legal, float* ha, float* q, hipStream_t s,
                  const char* name);
// the online net's trunk on split-bf16 MFMA with conv1's records kept for the rows [a1_lo, N): a1 / a2 / a3 come out as
// split records; trunk_unsplit_rows turns `rows` rows of each (from the given pointers) back into f32 in place
int lstmnet_trunk_records(const rela_lstmnet* n, int N, const uint8_t* s_dev, float* a1, float* a2, float* a3, int a1_lo,
                          hipStream_t s, const char* const* names);
int trunk_unsplit_rows(float* a1, float* a2, float* a3, int rows, hipStream_t s);
}  // namespace rela_amd
