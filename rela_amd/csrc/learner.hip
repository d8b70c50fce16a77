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
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "ffnet_layout.h"
#include "gemm_lds.h"
#include "prof.h"

namespace rela_amd {
namespace {

using namespace gemm;

using TileDgrad = TileCfg<128, 64, 4, 2, false>;  // M = batch rows, A k-contiguous
using TileWfc = TileCfg<128, 64, 4, 2, true>;     // fc weight gradient (M = 512 units)
using TileW64 = TileCfg<64, 64, 2, 4, true>;      // conv2 / conv3 weight gradients (M = 64 channels)
using TileW32 = TileCfg<32, 64, 2, 4, true>;      // conv1 / head weight gradients (M = 32)


// d_h[b][u] = relu'(h) * sum_k d_ha[b][k] * Wh[k][u]      Wh rows: 0..A-1 = fc_a.weight, 31 = fc_v.weight
struct ProbHeadDgrad : ProbBase {
  const float *d_ha, *a_w, *v_w, *h;
  float* d_h;
  int A;
  __device__ float4 loadA(int m, int k) const { return m < M ? ld4(d_ha + (size_t)m * 32 + k) : zero4(); }
  __device__ float4 loadB(int k, int n) const {
    if (k < A) return ld4(a_w + (size_t)k * 512 + n);
    if (k == 31) return ld4(v_w + n);
    return zero4();
  }
  __device__ void store(int, int m, int n, float v) const {
    const size_t i = (size_t)m * 512 + n;
    d_h[i] = h[i] > 0.f ? v : 0.f;
  }
};

// dWh[k][u] = sum_b d_ha[b][k] * h[b][u]
struct ProbHeadWgrad : ProbBase {
  const float *d_ha, *h;
  float *g_a_w, *g_v_w;
  int A;
  __device__ float4 loadA(int b, int m) const { return b < K ? ld4(d_ha + (size_t)b * 32 + m) : zero4(); }
  __device__ float4 loadB(int b, int n) const { return b < K ? ld4(h + (size_t)b * 512 + n) : zero4(); }
  __device__ void store(int, int m, int n, float v) const {
    if (m < A) g_a_w[(size_t)m * 512 + n] = v;
    else if (m == 31) g_v_w[n] = v;
  }
};

// d_a3[b][j] = relu'(a3) * sum_u d_h[b][u] * Wfc'[u][j]     j = pos*64 + c (channel-last)
struct ProbFcDgrad : ProbBase {
  const float *d_h, *wfcp, *a3;
  float* d_a3;
  __device__ float4 loadA(int m, int k) const { return m < M ? ld4(d_h + (size_t)m * 512 + k) : zero4(); }
  __device__ float4 loadB(int k, int n) const { return ld4(wfcp + (size_t)k * 3136 + n); }
  __device__ void store(int, int m, int n, float v) const {
    const size_t i = (size_t)m * 3136 + n;
    d_a3[i] = a3[i] > 0.f ? v : 0.f;
  }
};

// dWfc[u][c*49+pos] = sum_b d_h[b][u] * a3[b][pos*64+c]   (written in state_dict order, net.py:49)
struct ProbFcWgrad : ProbBase {
  const float *d_h, *a3;
  float* g_fc_w;
  __device__ float4 loadA(int b, int m) const { return b < K ? ld4(d_h + (size_t)b * 512 + m) : zero4(); }
  __device__ float4 loadB(int b, int n) const { return b < K ? ld4(a3 + (size_t)b * 3136 + n) : zero4(); }
  __device__ void store(int, int m, int n, float v) const {
    const int pos = n >> 6, c = n & 63;
    g_fc_w[(size_t)m * 3136 + c * 49 + pos] = v;
  }
};

// col[(b,pos)][j] = sum_oc d_out[(b,pos)][oc] * Wp[oc][j]     j = (kh,kw,c)   (64 output channels)
struct ProbConvDgrad : ProbBase {
  const float *d_out, *wp;
  float* col;
  __device__ float4 loadA(int m, int k) const { return m < M ? ld4(d_out + (size_t)m * 64 + k) : zero4(); }
  __device__ float4 loadB(int k, int n) const { return ld4(wp + (size_t)k * N + n); }
  __device__ void store(int, int m, int n, float v) const { col[(size_t)m * N + n] = v; }
};

// partial[z][oc][j] = sum_{(b,pos) in slice z} d_out[(b,pos)][oc] * patch(in)[(b,pos)][j]
template <int OC, int CIN, int KH, int KW, int STRIDE, int OH, int OW, int IH, int IW>
struct ProbConvWgrad : ProbBase {
  const float *d_out, *in;  // d_out [(b,pos)][OC]; in [b][IH][IW][CIN] channel-last
  float* part;
  __device__ float4 loadA(int k, int m) const { return k < K ? ld4(d_out + (size_t)k * OC + m) : zero4(); }
  __device__ float4 loadB(int k, int n) const {
    if (k >= K) return zero4();
    const int b = k / (OH * OW), pos = k - b * (OH * OW);
    const int oy = pos / OW, ox = pos - oy * OW;
    const int r = n / CIN, c = n - r * CIN;
    const int kh = r / KW, kw = r - kh * KW;
    return ld4(in + (((size_t)b * IH + oy * STRIDE + kh) * IW + ox * STRIDE + kw) * CIN + c);
  }
  __device__ void store(int z, int m, int n, float v) const { part[((size_t)z * M + m) * N + n] = v; }
};
using ProbW3 = ProbConvWgrad<64, 64, 3, 3, 1, 7, 7, 9, 9>;
using ProbW2 = ProbConvWgrad<64, 32, 4, 4, 2, 9, 9, 20, 20>;

// conv1: the input is the u8 frame stack [b][4][84][84]; j = (c,kh,kw) = state_dict order.
// The forward folds s/255 into the weights (net.py:46), so dW1 = (d_a1'^T x im2col(u8)) / 255,
// applied by reduce_splits.
struct ProbW1 : ProbBase {
  const float* d_out;  // [(b,pos)][32]
  const uint8_t* obs;
  float* part;
  __device__ float4 loadA(int k, int m) const { return k < K ? ld4(d_out + (size_t)k * 32 + m) : zero4(); }
  __device__ float4 loadB(int k, int n) const {
    if (k >= K) return zero4();
    const int b = k / 400, pos = k - b * 400;
    const int oy = pos / 20, ox = pos - oy * 20;
    const int c = n >> 6, kh = (n >> 3) & 7, kw = n & 7;
    const uint32_t d = *reinterpret_cast<const uint32_t*>(obs + (size_t)b * 28224 + c * 7056 + (oy * 4 + kh) * 84 +
                                                          ox * 4 + kw);
    return make_float4((float)(d & 0xff), (float)((d >> 8) & 0xff), (float)((d >> 16) & 0xff), (float)(d >> 24));
  }
  __device__ void store(int z, int m, int n, float v) const { part[((size_t)z * M + m) * N + n] = v; }
};


// ---- col2im (gather form) with the ReLU mask of the layer below ------------------------------
// d_a2[b][y][x][c] = relu'(a2) * sum_{kh,kw} col3[(b, (y-kh)*7 + x-kw)][(kh*3+kw)*64 + c]
__global__ void col2im3(const float* __restrict__ col, const float* __restrict__ a2, float* __restrict__ d_a2,
                        int Bn) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Bn * 81 * 16) return;
  const int c4 = idx & 15, pix = idx >> 4;
  const int x = pix % 9, y = (pix / 9) % 9, b = pix / 81;
  float4 s = zero4();
  for (int kh = 0; kh < 3; ++kh) {
    const int oy = y - kh;
    if (oy < 0 || oy >= 7) continue;
    for (int kw = 0; kw < 3; ++kw) {
      const int ox = x - kw;
      if (ox < 0 || ox >= 7) continue;
      const float4 v = ld4(col + ((size_t)b * 49 + oy * 7 + ox) * 576 + (kh * 3 + kw) * 64 + c4 * 4);
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
  }
  const float4 a = ld4(a2 + (size_t)pix * 64 + c4 * 4);
  *reinterpret_cast<float4*>(d_a2 + (size_t)pix * 64 + c4 * 4) =
      make_float4(a.x > 0.f ? s.x : 0.f, a.y > 0.f ? s.y : 0.f, a.z > 0.f ? s.z : 0.f, a.w > 0.f ? s.w : 0.f);
}

// d_a1[b][y][x][c] = relu'(a1) * sum_{kh,kw: y-kh = 2*oy, x-kw = 2*ox} col2[(b, oy*9+ox)][(kh*4+kw)*32 + c]
__global__ void col2im2(const float* __restrict__ col, const float* __restrict__ a1, float* __restrict__ d_a1,
                        int Bn) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Bn * 400 * 8) return;
  const int c4 = idx & 7, pix = idx >> 3;
  const int x = pix % 20, y = (pix / 20) % 20, b = pix / 400;
  float4 s = zero4();
  for (int kh = 0; kh < 4; ++kh) {
    const int ty = y - kh;
    if (ty < 0 || (ty & 1) || ty > 16) continue;
    for (int kw = 0; kw < 4; ++kw) {
      const int tx = x - kw;
      if (tx < 0 || (tx & 1) || tx > 16) continue;
      const float4 v = ld4(col + ((size_t)b * 81 + (ty >> 1) * 9 + (tx >> 1)) * 512 + (kh * 4 + kw) * 32 + c4 * 4);
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
  }
  const float4 a = ld4(a1 + (size_t)pix * 32 + c4 * 4);
  *reinterpret_cast<float4*>(d_a1 + (size_t)pix * 32 + c4 * 4) =
      make_float4(a.x > 0.f ? s.x : 0.f, a.y > 0.f ? s.y : 0.f, a.z > 0.f ? s.z : 0.f, a.w > 0.f ? s.w : 0.f);
}

// ---- split-K reduction, written in state_dict order -----------------------------------------
enum { kRedConv1 = 0, kRedConv2 = 1, kRedConv3 = 2 };
__global__ void reduce_splits(const float* __restrict__ part, int splits, int M, int N, int mode,
                              float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * N) return;
  const int m = idx / N, n = idx - m * N;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += part[((size_t)z * M + m) * N + n];
  if (mode == kRedConv1) {
    out[(size_t)m * N + n] = s / 255.0f;
  } else if (mode == kRedConv2) {  // n = (kh*4+kw)*32 + c -> [oc][c][kh][kw]
    const int c = n & 31, r = n >> 5;
    out[((size_t)(m * 32 + c) * 4 + (r >> 2)) * 4 + (r & 3)] = s;
  } else {  // n = (kh*3+kw)*64 + c -> [oc][c][kh][kw]
    const int c = n & 63, r = n >> 6;
    out[((size_t)(m * 64 + c) * 3 + r / 3) * 3 + r % 3] = s;
  }
}

// ---- column sums (bias gradients): two deterministic stages ----------------------------------
constexpr int kColsumBlocks = 256;
// src [rows][C], C in {32, 64, 512}: a thread owns one float4 of columns and every L-th row of the
// block's row slice (coalesced 16-byte loads), LDS reduction over the L row lanes
__global__ __launch_bounds__(kLT) void colsum_partial(const float* __restrict__ src, int64_t rows, int C,
                                                      float* __restrict__ part) {
  __shared__ float4 sm[kLT];
  const int G = C / 4, L = kLT / G;
  const int cg = threadIdx.x % G, rl = threadIdx.x / G;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  float4 s = zero4();
  for (int64_t r = r0 + rl; r < r1; r += L) {
    const float4 v = ld4(src + r * C + cg * 4);
    s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0) {
    for (int l = 1; l < L; ++l) {
      const float4 v = sm[cg + l * G];
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * C + cg * 4) = s;
  }
}
// one block per float4 of columns: thread b holds partial b, fixed-shape tree reduction in LDS
__global__ __launch_bounds__(kColsumBlocks) void colsum_final(const float* __restrict__ part, int C,
                                                              float* __restrict__ out) {
  __shared__ float4 sm[kColsumBlocks];
  const int cg = blockIdx.x;
  sm[threadIdx.x] = ld4(part + (size_t)threadIdx.x * C + cg * 4);
  __syncthreads();
  for (int o = kColsumBlocks / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const float4 v = sm[threadIdx.x + o];
      float4& d = sm[threadIdx.x];
      d.x += v.x, d.y += v.y, d.z += v.z, d.w += v.w;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *reinterpret_cast<float4*>(out + cg * 4) = sm[0];
}
__global__ void head_bias_grad(const float* __restrict__ s32, int A, float* __restrict__ g_a_b,
                               float* __restrict__ g_v_b) {
  const int k = threadIdx.x;
  if (k < A) g_a_b[k] = s32[k];
  if (k == 31) g_v_b[0] = s32[31];
}

// ---- loss: smooth_l1(err) * w, mean over the batch (apex.py:87, main.py:228), and its gradient
// through the dueling head  q = v + a*legal - mean_A(a*legal)  (net.py:33-39) --------------------
__global__ __launch_bounds__(kLT) void learner_loss_grad(const float* __restrict__ td, const float* __restrict__ w,
                                                         const int64_t* __restrict__ act,
                                                         const float* __restrict__ legal, int Bn, int A,
                                                         float* __restrict__ d_ha, float* __restrict__ loss_out) {
  __shared__ float red[kLT];
  float lsum = 0.f;
  const float inv_b = 1.0f / (float)Bn, inv_a = 1.0f / (float)A;
  for (int i = threadIdx.x; i < Bn; i += kLT) {
    const float e = td[i], ae = fabsf(e);
    lsum += (ae < 1.0f ? 0.5f * e * e : ae - 0.5f) * w[i];
    // err = target - q[a]:  d mean(loss*w) / d q[a] = -w * clamp(err, -1, 1) / B
    const float g = -(w[i] * fminf(fmaxf(e, -1.0f), 1.0f)) * inv_b;
    const int a = (int)act[i];
    float* row = d_ha + (size_t)i * 32;
    for (int k = 0; k < 32; ++k) {
      float v = 0.f;
      if (k < A) v = legal[(size_t)i * A + k] * (g * ((k == a ? 1.0f : 0.0f) - inv_a));
      if (k == 31) v = g;
      row[k] = v;
    }
  }
  red[threadIdx.x] = lsum;
  __syncthreads();
  for (int o = kLT / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = red[0] * inv_b;
}

// ---- weight copies in the k order the dgrad GEMMs read ----------------------------------------
enum { kPermConv2 = 0, kPermConv3 = 1, kPermFc = 2 };
__global__ void permute_weights(int mode, const float* __restrict__ src, float* __restrict__ dst, int total) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  if (mode == kPermConv2) {  // dst[oc][(kh*4+kw)*32+c]
    const int oc = idx >> 9, n = idx & 511, c = n & 31, r = n >> 5;
    dst[idx] = src[((oc * 32 + c) * 4 + (r >> 2)) * 4 + (r & 3)];
  } else if (mode == kPermConv3) {  // dst[oc][(kh*3+kw)*64+c]
    const int oc = idx / 576, n = idx - oc * 576, c = n & 63, r = n >> 6;
    dst[idx] = src[((oc * 64 + c) * 3 + r / 3) * 3 + r % 3];
  } else {  // dst[u][pos*64+c] <- src[u][c*49+pos]
    const int u = idx / 3136, n = idx - u * 3136, c = n & 63, pos = n >> 6;
    dst[idx] = src[(size_t)u * 3136 + c * 49 + pos];
  }
}

// ---- clip_grad_norm_ + optimiser over the flat buffers ----------------------------------------
constexpr int kNormBlocks = 256;
__global__ __launch_bounds__(256) void sumsq_partial(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    s += (double)g[i] * (double)g[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
// out[0] = total 2-norm, out[1] = min(1, max_norm / (norm + 1e-6))   (torch clip_grad_norm_)
__global__ void clip_coef(const double* __restrict__ part, int nblk, float max_norm, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += part[i];
  const float norm = (float)sqrt(s);
  out[0] = norm;
  const float c = max_norm / (norm + 1e-6f);
  out[1] = c < 1.0f ? c : 1.0f;
}
// torch.optim.RMSprop (momentum 0, not centred): sq = alpha*sq + (1-alpha)*g*g; p -= lr * g / (sqrt(sq) + eps)
__global__ void rmsprop_update(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, int64_t n,
                               float lr, float alpha, float eps, const float* __restrict__ coef) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * coef[1];
  const float s = alpha * sq[i] + (1.0f - alpha) * gi * gi;
  sq[i] = s;
  p[i] -= lr * (gi / (sqrtf(s) + eps));
}
// torch.optim.Adam (no amsgrad, no weight decay); bias corrections computed on the host per step
__global__ void adam_update(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1,
                            float* __restrict__ m2, int64_t n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, const float* __restrict__ coef) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * coef[1];
  const float a = b1 * m1[i] + (1.0f - b1) * gi;
  const float b = b2 * m2[i] + (1.0f - b2) * gi * gi;
  m1[i] = a;
  m2[i] = b;
  p[i] -= (lr / bc1) * (a / (sqrtf(b) / bc2_sqrt + eps));
}

}  // namespace
}  // namespace rela_amd

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
};

namespace {
constexpr int kSplitW3 = 28, kSplitW2 = 27, kSplitW1 = 64;

rela_ffnet_params params_at(const rela_apex_learner* l, float* base) {
  rela_ffnet_params p;
  const float** f = reinterpret_cast<const float**>(&p);
  for (int i = 0; i < 12; ++i) f[i] = base + l->off[i];
  return p;
}

int repack(rela_apex_learner* l, bool online, bool target, hipStream_t s) {
  if (online) {
    const rela_ffnet_params p = params_at(l, l->P);
    int rc = rela_ffnet_load(l->online, &p, 1, s);
    if (rc != RELA_OK) return rc;
    hipLaunchKernelGGL(permute_weights, dim3(ceil_div(64 * 512, 256)), dim3(256), 0, s, kPermConv2, p.conv2_w, l->w2p,
                       64 * 512);
    hipLaunchKernelGGL(permute_weights, dim3(ceil_div(64 * 576, 256)), dim3(256), 0, s, kPermConv3, p.conv3_w, l->w3p,
                       64 * 576);
    hipLaunchKernelGGL(permute_weights, dim3(ceil_div(512 * 3136, 256)), dim3(256), 0, s, kPermFc, p.fc_w, l->wfcp,
                       512 * 3136);
    RELA_LAUNCH_CHECK();
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
  RELA_HIP(hipMalloc(&l->w2p, sizeof(float) * 64 * 512));
  RELA_HIP(hipMalloc(&l->w3p, sizeof(float) * 64 * 576));
  RELA_HIP(hipMalloc(&l->wfcp, sizeof(float) * 512 * 3136));
  l->ws_bytes = rela_ffnet_workspace_bytes(nullptr, max_batch);
  RELA_HIP(hipMalloc(&l->ws_on, (size_t)l->ws_bytes));
  RELA_HIP(hipMalloc(&l->ws_tmp, (size_t)l->ws_bytes));
  RELA_HIP(hipMalloc(&l->q, sizeof(float) * 3 * B * A));
  RELA_HIP(hipMalloc(&l->td, sizeof(float) * B));
  RELA_HIP(hipMalloc(&l->d_ha, sizeof(float) * B * 32));
  RELA_HIP(hipMalloc(&l->d_h, sizeof(float) * B * 512));
  RELA_HIP(hipMalloc(&l->d_a3, sizeof(float) * B * kA3));
  RELA_HIP(hipMalloc(&l->d_a2, sizeof(float) * B * kA2));
  RELA_HIP(hipMalloc(&l->d_a1, sizeof(float) * B * kA1));
  RELA_HIP(hipMalloc(&l->col, sizeof(float) * B * (81 * 512 > 49 * 576 ? 81 * 512 : 49 * 576)));
  RELA_HIP(hipMalloc(&l->part, sizeof(float) * (size_t)kSplitW3 * 64 * 576));  // the largest of the three
  static_assert(kSplitW3 * 64 * 576 >= kSplitW2 * 64 * 512 && kSplitW3 * 64 * 576 >= kSplitW1 * 32 * 256, "part size");
  RELA_HIP(hipMalloc(&l->cpart, sizeof(float) * kColsumBlocks * 512));
  RELA_HIP(hipMalloc(&l->s32, sizeof(float) * 32));
  RELA_HIP(hipMalloc(&l->npart, sizeof(double) * kNormBlocks));
  RELA_HIP(hipMalloc(&l->norm, sizeof(float) * 2));
  RELA_HIP(hipMalloc(&l->loss, sizeof(float)));
  *out = l;
  return RELA_OK;
}

extern "C" void rela_apex_learner_destroy(rela_apex_learner* l) {
  if (!l) return;
  DeviceGuard g(l->device);
  (void)hipDeviceSynchronize();
  void* ps[] = {l->P,  l->PT,   l->G,    l->S1,   l->S2,   l->w2p, l->w3p,  l->wfcp,  l->ws_on, l->ws_tmp, l->q,
                l->td, l->d_ha, l->d_h,  l->d_a3, l->d_a2, l->d_a1, l->col, l->part,  l->cpart, l->s32,    l->npart,
                l->norm, l->loss};
  for (void* p : ps) (void)hipFree(p);
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

extern "C" int rela_apex_learner_backward(rela_apex_learner* l, int batch, const void* const* rows_dev,
                                          const float* weight_dev, float* priority_dev, float* loss_dev,
                                          void* stream_) {
  RELA_CHECK(l && l->loaded, RELA_ESTATE, "rela_apex_learner_backward: parameters were never loaded");
  RELA_CHECK(batch >= 1 && batch <= l->Bmax && rows_dev && weight_dev && priority_dev, RELA_EINVAL,
             "rela_apex_learner_backward: bad arguments (batch %d, max %d)", batch, l->Bmax);
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
             "rela_apex_learner_backward: a batch field is NULL");
  float* q_on = l->q;
  float* q_no = l->q + (size_t)Bn * A;
  float* q_nt = l->q + 2 * (size_t)Bn * A;
  // td_err (apex.py:30-45): greedy_act(next_obs) and target_net(next_obs) carry no gradient
  int rc = rela_ffnet_forward(l->online, Bn, nobs, nlegal, q_no, l->ws_tmp, l->ws_bytes, s);
  if (rc != RELA_OK) return rc;
  rc = rela_ffnet_forward(l->target, Bn, nobs, nlegal, q_nt, l->ws_tmp, l->ws_bytes, s);
  if (rc != RELA_OK) return rc;
  rc = rela_ffnet_forward(l->online, Bn, obs, legal, q_on, l->ws_on, l->ws_bytes, s);
  if (rc != RELA_OK) return rc;
  rc = rela_apex_td_from_q(Bn, A, 0, q_on, q_no, q_nt, nlegal, act, reward, boot, l->gamma_n, l->td, priority_dev, s);
  if (rc != RELA_OK) return rc;
  {
    ProfScope prof("learner_loss_grad", s);
    hipLaunchKernelGGL(learner_loss_grad, dim3(1), dim3(kLT), 0, s, (const float*)l->td, weight_dev, act, legal, Bn,
                       A, l->d_ha, l->loss);
  }
  if (loss_dev) RELA_HIP(hipMemcpyAsync(loss_dev, l->loss, sizeof(float), hipMemcpyDeviceToDevice, s));

  const FFNetWs w = ffnet_ws(l->ws_on, Bn);
  const rela_ffnet_params P = params_at(l, l->P);
  float* Gm[12];  // gradient tensors in rela_ffnet_params order
  for (int i = 0; i < 12; ++i) Gm[i] = l->G + l->off[i];

  auto colsum = [&](const float* src, int64_t rows, int C, float* out) {
    ProfScope prof("learner_colsum", s);
    hipLaunchKernelGGL(colsum_partial, dim3(kColsumBlocks), dim3(kLT), 0, s, src, rows, C, l->cpart);
    hipLaunchKernelGGL(colsum_final, dim3(C / 4), dim3(kColsumBlocks), 0, s, (const float*)l->cpart, C, out);
  };

  // heads: d_h, dWh, db
  {
    ProbHeadDgrad p{};
    p.M = Bn, p.N = 512, p.K = 32;
    p.d_ha = l->d_ha, p.a_w = P.a_w, p.v_w = P.v_w, p.h = w.h, p.d_h = l->d_h, p.A = A;
    launch_gemm<TileDgrad>(p, 1, s, "learner_dgrad_heads");
  }
  {
    ProbHeadWgrad p{};
    p.M = 32, p.N = 512, p.K = Bn;
    p.d_ha = l->d_ha, p.h = w.h, p.g_a_w = Gm[10], p.g_v_w = Gm[8], p.A = A;
    launch_gemm<TileW32>(p, 1, s, "learner_wgrad_heads");
  }
  colsum(l->d_ha, Bn, 32, l->s32);
  hipLaunchKernelGGL(head_bias_grad, dim3(1), dim3(32), 0, s, (const float*)l->s32, A, Gm[11], Gm[9]);
  // fc: d_a3, dWfc, db
  {
    ProbFcDgrad p{};
    p.M = Bn, p.N = 3136, p.K = 512;
    p.d_h = l->d_h, p.wfcp = l->wfcp, p.a3 = w.a3, p.d_a3 = l->d_a3;
    launch_gemm<TileDgrad>(p, 1, s, "learner_dgrad_fc");
  }
  {
    ProbFcWgrad p{};
    p.M = 512, p.N = 3136, p.K = Bn;
    p.d_h = l->d_h, p.a3 = w.a3, p.g_fc_w = Gm[6];
    launch_gemm<TileWfc>(p, 1, s, "learner_wgrad_fc");
  }
  colsum(l->d_h, Bn, 512, Gm[7]);
  // conv3: dW3, db3, d_a2
  {
    ProbW3 p{};
    p.M = 64, p.N = 576, p.K = Bn * 49;
    p.d_out = l->d_a3, p.in = w.a2, p.part = l->part;
    launch_gemm<TileW64>(p, kSplitW3, s, "learner_wgrad_conv3");
    hipLaunchKernelGGL(reduce_splits, dim3(ceil_div(64 * 576, 256)), dim3(256), 0, s, (const float*)l->part, kSplitW3,
                       64, 576, kRedConv3, Gm[4]);
  }
  colsum(l->d_a3, (int64_t)Bn * 49, 64, Gm[5]);
  {
    ProbConvDgrad p{};
    p.M = Bn * 49, p.N = 576, p.K = 64;
    p.d_out = l->d_a3, p.wp = l->w3p, p.col = l->col;
    launch_gemm<TileDgrad>(p, 1, s, "learner_dgrad_conv3");
    ProfScope prof("learner_col2im", s);
    hipLaunchKernelGGL(col2im3, dim3(ceil_div(Bn * 81 * 16, 256)), dim3(256), 0, s, (const float*)l->col,
                       (const float*)w.a2, l->d_a2, Bn);
  }
  // conv2: dW2, db2, d_a1
  {
    ProbW2 p{};
    p.M = 64, p.N = 512, p.K = Bn * 81;
    p.d_out = l->d_a2, p.in = w.a1, p.part = l->part;
    launch_gemm<TileW64>(p, kSplitW2, s, "learner_wgrad_conv2");
    hipLaunchKernelGGL(reduce_splits, dim3(ceil_div(64 * 512, 256)), dim3(256), 0, s, (const float*)l->part, kSplitW2,
                       64, 512, kRedConv2, Gm[2]);
  }
  colsum(l->d_a2, (int64_t)Bn * 81, 64, Gm[3]);
  {
    ProbConvDgrad p{};
    p.M = Bn * 81, p.N = 512, p.K = 64;
    p.d_out = l->d_a2, p.wp = l->w2p, p.col = l->col;
    launch_gemm<TileDgrad>(p, 1, s, "learner_dgrad_conv2");
    ProfScope prof("learner_col2im", s);
    hipLaunchKernelGGL(col2im2, dim3(ceil_div(Bn * 400 * 8, 256)), dim3(256), 0, s, (const float*)l->col,
                       (const float*)w.a1, l->d_a1, Bn);
  }
  // conv1: dW1, db1 (no gradient flows into the frames)
  {
    ProbW1 p{};
    p.M = 32, p.N = 256, p.K = Bn * 400;
    p.d_out = l->d_a1, p.obs = obs, p.part = l->part;
    launch_gemm<TileW32>(p, kSplitW1, s, "learner_wgrad_conv1");
    hipLaunchKernelGGL(reduce_splits, dim3(ceil_div(32 * 256, 256)), dim3(256), 0, s, (const float*)l->part, kSplitW1,
                       32, 256, kRedConv1, Gm[0]);
  }
  colsum(l->d_a1, (int64_t)Bn * 400, 32, Gm[1]);
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

extern "C" int rela_apex_learner_apply(rela_apex_learner* l, void* stream_) {
  RELA_CHECK(l && l->loaded, RELA_ESTATE, "rela_apex_learner_apply: parameters were never loaded");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(l->device);
  const int64_t n = l->off[12];
  {
    ProfScope prof("learner_optimizer", s);
    hipLaunchKernelGGL(sumsq_partial, dim3(kNormBlocks), dim3(256), 0, s, (const float*)l->G, n, l->npart);
    hipLaunchKernelGGL(clip_coef, dim3(1), dim3(1), 0, s, (const double*)l->npart, kNormBlocks, l->clip, l->norm);
    if (l->optimizer == 0) {
      hipLaunchKernelGGL(rmsprop_update, dim3(ceil_div(n, 256)), dim3(256), 0, s, l->P, (const float*)l->G, l->S1, n,
                         l->lr, 0.99f, l->opt_eps, (const float*)l->norm);
    } else {
      l->adam_t += 1;
      const float b1 = 0.9f, b2 = 0.999f;
      const float bc1 = 1.0f - (float)pow((double)b1, (double)l->adam_t);
      const float bc2s = (float)sqrt(1.0 - pow((double)b2, (double)l->adam_t));
      hipLaunchKernelGGL(adam_update, dim3(ceil_div(n, 256)), dim3(256), 0, s, l->P, (const float*)l->G, l->S1, l->S2,
                         n, l->lr, b1, b2, l->opt_eps, bc1, bc2s, (const float*)l->norm);
    }
  }
  RELA_LAUNCH_CHECK();
  return repack(l, true, false, s);
}
