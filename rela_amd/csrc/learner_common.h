// learner_common.h -- pieces shared by the two hand-written learner steps (learner.hip: Ape-X /
// AtariFFNet; learner_r2d2.hip: R2D2 / AtariLSTMNet): the gemm_lds problem descriptions of the conv trunk's
// backward pass, col2im, split-K reduction, column sums, global-norm clipping and the optimisers.
// Everything sits in an anonymous namespace: each translation unit gets its own copy of the kernels.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"
#include "ffnet_layout.h"
#include "gemm_lds.h"
#include "gemm_bf16x3.h"
#include "prof.h"
#include "dgrad_conv_bf16.h"
#include "wgrad_conv1_bf16.h"
#include "wgrad_conv2_bf16.h"
#include "wgrad_conv3_bf16.h"

namespace rela_amd {
namespace {

using namespace gemm;

using TileDgrad = TileCfg<128, 64, 4, 2, false>;  // M = batch rows, A k-contiguous
using TileWfc = TileCfg<128, 64, 4, 2, true>;     // fc weight gradient (M = 512 units)
using TileW64 = TileCfg<64, 64, 2, 4, true>;      // conv2 / conv3 weight gradients (M = 64 channels)
using TileW32 = TileCfg<32, 64, 2, 4, true>;      // conv1 / head weight gradients (M = 32)
// the same shapes on the bf16 matrix cores (gemm_bf16x3.h: operands split into bf16 hi + lo on the way into LDS, three
// MFMAs per product): the learners' bf16x2 mode.
using Tile3Dgrad = gemm3::TileCfg<128, 64, 4, 2, false>;
using Tile3Wfc = gemm3::TileCfg<128, 64, 4, 2, true>;
using Tile3W64 = gemm3::TileCfg<64, 64, 2, 4, true>;
using Tile3W32 = gemm3::TileCfg<32, 64, 2, 4, true>;
inline bool gemm_bf16x3_on() { return true; }
// ... and with every operand as THREE bf16 parts, six products (gemm_bf16x3.h, PARTS = 3): f32 accuracy on the bf16 MFMA --
// the learners' f32x3 mode
using Tile6Dgrad = gemm3::TileCfg<128, 64, 4, 2, false, 3>;
using Tile6Wfc = gemm3::TileCfg<128, 64, 4, 2, true, 3>;
using Tile6W64 = gemm3::TileCfg<64, 64, 2, 4, true, 3>;
using Tile6W32 = gemm3::TileCfg<32, 64, 2, 4, true, 3>;


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

// the same over many rows (R2D2: 5,312 training rows), split over blockIdx.z into partial tiles part[z][32][512]
// that head_wgrad_reduce sums in z order (one 32 x 512 output: without the split the GEMM is 8 blocks)
struct ProbHeadWgradPart : ProbBase {
  const float *d_ha, *h;
  float* part;
  __device__ float4 loadA(int b, int m) const { return b < K ? ld4(d_ha + (size_t)b * 32 + m) : zero4(); }
  __device__ float4 loadB(int b, int n) const { return b < K ? ld4(h + (size_t)b * 512 + n) : zero4(); }
  __device__ void store(int z, int m, int n, float v) const { part[((size_t)z * 32 + m) * 512 + n] = v; }
};
__global__ void head_wgrad_reduce(const float* __restrict__ part, int splits, int A, float* __restrict__ g_a_w,
                                  float* __restrict__ g_v_w) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 32 * 512) return;
  const int m = idx >> 9, n = idx & 511;
  if (m >= A && m != 31) return;
  float s = 0.f;
#pragma unroll 8
  for (int z = 0; z < splits; ++z) s += part[(size_t)z * (32 * 512) + idx];
  if (m < A) g_a_w[(size_t)m * 512 + n] = s;
  else g_v_w[n] = s;
}

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
#pragma unroll 8  // (the adds stay in z order; eight loads in flight instead of one)
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
// several column sums in ONE launch pair (the five bias gradients of a learner step were ten ~7 us launches):
// blockIdx.y selects the job; every job keeps the two-stage order of colsum_partial / colsum_final, so the results
// are bit-identical to separate launches.
constexpr int kMaxColsumJobs = 6;
struct ColsumJobs {
  const float* src[kMaxColsumJobs];
  float* out[kMaxColsumJobs];
  float* part[kMaxColsumJobs];  // kColsumBlocks * C floats each
  int64_t rows[kMaxColsumJobs];
  int C[kMaxColsumJobs];
  int n = 0;
  void add(const float* s, int64_t r, int c, float* o) { src[n] = s, rows[n] = r, C[n] = c, out[n] = o, ++n; }
};
__global__ __launch_bounds__(kLT) void colsum_partial_multi(ColsumJobs jobs) {
  __shared__ float4 sm[kLT];
  const int j = blockIdx.y;
  const float* src = jobs.src[j];
  const int64_t rows = jobs.rows[j];
  const int C = jobs.C[j];
  const int G = C / 4, L = kLT / G;
  const int cg = threadIdx.x % G, rl = threadIdx.x / G;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  float4 s = zero4();
  // eight row loads in flight per thread (one at a time left the 26 MB of d_a1 latency-bound: 47 us per step); the
  // adds keep their order, so the sums are bit-identical to the one-load loop
  int64_t r = r0 + rl;
  for (; r + 7 * (int64_t)L < r1; r += 8 * (int64_t)L) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld4(src + (r + u * (int64_t)L) * C + cg * 4);
#pragma unroll
    for (int u = 0; u < 8; ++u) s.x += v[u].x, s.y += v[u].y, s.z += v[u].z, s.w += v[u].w;
  }
  for (; r < r1; r += L) {
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
    *reinterpret_cast<float4*>(jobs.part[j] + (size_t)blockIdx.x * C + cg * 4) = s;
  }
}
__global__ __launch_bounds__(kColsumBlocks) void colsum_final_multi(ColsumJobs jobs) {
  __shared__ float4 sm[kColsumBlocks];
  const int j = blockIdx.y, cg = blockIdx.x, C = jobs.C[j];
  if (cg >= C / 4) return;  // (whole block)
  sm[threadIdx.x] = ld4(jobs.part[j] + (size_t)threadIdx.x * C + cg * 4);
  __syncthreads();
  for (int o = kColsumBlocks / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const float4 v = sm[threadIdx.x + o];
      float4& d = sm[threadIdx.x];
      d.x += v.x, d.y += v.y, d.z += v.z, d.w += v.w;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *reinterpret_cast<float4*>(jobs.out[j] + cg * 4) = sm[0];
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

// td_err (apex.py:30-45: batch-global q.min() of greedy_act(next_obs), double-DQN target, un-fused arithmetic as
// agent_ops.hip's td_kernel) + the loss kernel above in ONE single-workgroup launch (two latency-bound launches
// before: 15 + 19 us).  Same arithmetic in the same order, so priorities, loss and d_ha are bit-identical.
__global__ __launch_bounds__(1024) void learner_td_loss_grad(int Bn, int A, const float* __restrict__ q,
                                                             const float* __restrict__ qno, const float* __restrict__ qnt,
                                                             const float* __restrict__ nlegal,
                                                             const int64_t* __restrict__ act,
                                                             const float* __restrict__ reward,
                                                             const float* __restrict__ bootstrap, float gamma_n,
                                                             const float* __restrict__ w, const float* __restrict__ legal,
                                                             float* __restrict__ td, float* __restrict__ prio,
                                                             float* __restrict__ d_ha, float* __restrict__ loss_out) {
  __shared__ float red[1024];
  float m = INFINITY;
  for (int i = threadIdx.x; i < Bn * A; i += 1024) m = fminf(m, qno[i]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] = fminf(red[threadIdx.x], red[threadIdx.x + off]);
    __syncthreads();
  }
  const float qmin = red[0];
  __syncthreads();
  float lsum = 0.f;
  const float inv_b = 1.0f / (float)Bn, inv_a = 1.0f / (float)A;
  for (int i = threadIdx.x; i < Bn; i += 1024) {
    int na = 0;
    float bv = -INFINITY;
#pragma unroll 6  // (the loads of several actions in flight: one at a time this loop was most of the kernel's 19 us)
    for (int j = 0; j < A; ++j) {  // greedy_act(next_obs): first maximal index of (1 + q - qmin) * legal (apex.py:51)
      const float lq = __fmul_rn(__fsub_rn(__fadd_rn(1.0f, qno[(size_t)i * A + j]), qmin), nlegal[(size_t)i * A + j]);
      if (lq > bv) bv = lq, na = j;
    }
    const int a = (int)act[i];
    const float qa = q[(size_t)i * A + a];
    const float bq = qnt[(size_t)i * A + na];
    const float tgt = __fadd_rn(reward[i], __fmul_rn(__fmul_rn(bootstrap[i], gamma_n), bq));
    const float e = __fsub_rn(tgt, qa), ae = fabsf(e);
    td[i] = e;
    prio[i] = ae;
    lsum += (ae < 1.0f ? 0.5f * e * e : ae - 0.5f) * w[i];
    const float g = -(w[i] * fminf(fmaxf(e, -1.0f), 1.0f)) * inv_b;
    float4* row = reinterpret_cast<float4*>(d_ha + (size_t)i * 32);
    float v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      v[k] = 0.f;
      if (k < A) v[k] = legal[(size_t)i * A + k] * (g * ((k == a ? 1.0f : 0.0f) - inv_a));
      if (k == 31) v[k] = g;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) row[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
  }
  // the same tree as learner_loss_grad's kLT = 512 lanes when Bn <= 512: lanes >= 512 hold zeros and fold in first
  red[threadIdx.x] = lsum;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = red[0] * inv_b;
}

// ---- weight copies in the k order the dgrad GEMMs read (ffnet_layout.h: permute_weight_at) ------
__global__ void permute_weights(int mode, const float* __restrict__ src, float* __restrict__ dst, int total) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < total) permute_weight_at(mode, idx, src, dst);
}

// ---- clip_grad_norm_ + optimiser over the flat buffers ----------------------------------------
constexpr int kNormBlocks = 256;
__global__ __launch_bounds__(256) void sumsq_partial(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
  // n is a multiple of 4 (every segment of the flat buffer is padded to 4 floats) and g is 16-byte aligned: float4
  // loads, four in flight per thread; fixed association (deterministic), f64 accumulation as before
  __shared__ double red[256];
  double s = 0.0;
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * 256;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = g4[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      s += (double)v[u].x * (double)v[u].x + (double)v[u].y * (double)v[u].y + (double)v[u].z * (double)v[u].z +
           (double)v[u].w * (double)v[u].w;
  }
  for (; i < n4; i += stride) {
    const float4 v = g4[i];
    s += (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z + (double)v.w * (double)v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) s += (double)g[(n4 << 2) + threadIdx.x] * (double)g[(n4 << 2) + threadIdx.x];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
// out[0] = total 2-norm, out[1] = min(1, max_norm / (norm + 1e-6))   (torch clip_grad_norm_)
// `abort_word` (may be NULL): a device word that is non-zero when the gradients of this step must not be applied (the
// R2D2 learner's grid-barrier timeout word): the coefficient becomes -1 and the update kernels leave everything alone.
__global__ __launch_bounds__(256) void clip_coef(const double* __restrict__ part, int nblk, float max_norm,
                                                 float* __restrict__ out, const unsigned* __restrict__ abort_word) {
  // one block of 256 threads: fixed-order tree over the (at most 256) partial sums (a single thread walking them took
  // 20 us of every learner step)
  __shared__ double red[256];
  red[threadIdx.x] = (int)threadIdx.x < nblk ? part[threadIdx.x] : 0.0;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double s = red[0];
  const float norm = (float)sqrt(s);
  out[0] = norm;
  const float c = max_norm / (norm + 1e-6f);
  out[1] = c < 1.0f ? c : 1.0f;
  if (abort_word && *abort_word != 0) out[1] = -1.0f;
}
// torch.optim.RMSprop (momentum 0, not centred): sq = alpha*sq + (1-alpha)*g*g; p -= lr * g / (sqrt(sq) + eps)
// (four elements per thread: 16-byte accesses; n4 = ceil(n / 4), the flat buffers are padded to 4 floats)
__global__ void rmsprop_update(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, int64_t n4,
                               float lr, float alpha, float eps, const float* __restrict__ coef) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float cf = coef[1];
  if (i >= n4 || cf < 0.f) return;
  const float4 gv = reinterpret_cast<const float4*>(g)[i];
  float4 sv = reinterpret_cast<float4*>(sq)[i], pv = reinterpret_cast<float4*>(p)[i];
  auto upd = [&](float gi, float& s, float& pi) {
    gi = gi * cf;
    s = alpha * s + (1.0f - alpha) * gi * gi;
    pi -= lr * (gi / (sqrtf(s) + eps));
  };
  upd(gv.x, sv.x, pv.x), upd(gv.y, sv.y, pv.y), upd(gv.z, sv.z, pv.z), upd(gv.w, sv.w, pv.w);
  reinterpret_cast<float4*>(sq)[i] = sv;
  reinterpret_cast<float4*>(p)[i] = pv;
}
// torch.optim.Adam (no amsgrad, no weight decay); bias corrections computed on the host per step
__global__ void adam_update(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1,
                            float* __restrict__ m2, int64_t n4, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, const float* __restrict__ coef) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float cf = coef[1];
  if (i >= n4 || cf < 0.f) return;
  const float4 gv = reinterpret_cast<const float4*>(g)[i];
  float4 av = reinterpret_cast<float4*>(m1)[i], bv = reinterpret_cast<float4*>(m2)[i], pv = reinterpret_cast<float4*>(p)[i];
  auto upd = [&](float gi, float& a, float& b, float& pi) {
    gi = gi * cf;
    a = b1 * a + (1.0f - b1) * gi;
    b = b2 * b + (1.0f - b2) * gi * gi;
    pi -= (lr / bc1) * (a / (sqrtf(b) / bc2_sqrt + eps));
  };
  upd(gv.x, av.x, bv.x, pv.x), upd(gv.y, av.y, bv.y, pv.y), upd(gv.z, av.z, bv.z, pv.z), upd(gv.w, av.w, bv.w, pv.w);
  reinterpret_cast<float4*>(m1)[i] = av;
  reinterpret_cast<float4*>(m2)[i] = bv;
  reinterpret_cast<float4*>(p)[i] = pv;
}


// ---- backward pass of the conv trunk (net.{0,2,4}), shared by both learners ---------------------
constexpr int kSplitW3 = 28, kSplitW2 = 27, kSplitW1 = 64;
// the largest partial buffer: conv3's split-K tiles, or one tile per block of the bf16 conv1 ([32][256]) / conv2
// ([64][512]) gradients
constexpr size_t kTrunkPartFloats = (size_t)w3fast::kMaxBlocks * 64 * 576;
static_assert(kTrunkPartFloats * 4 >= dgfast::kFrag3Bytes && kTrunkPartFloats * 4 >= dgfast::kFrag2Bytes, "frag scratch");
static_assert(kTrunkPartFloats >= (size_t)kSplitW3 * 8 * 64 * 576, "part size (split multiplier up to 8)");
static_assert(kTrunkPartFloats >= (size_t)kSplitW3 * 64 * 576 && kTrunkPartFloats >= (size_t)w1fast::kMaxBlocks * 32 * 256 &&
                  kTrunkPartFloats >= (size_t)w2fast::kMaxBlocks * 64 * 512,
              "part size");
static_assert(kSplitW3 * 64 * 576 >= kSplitW2 * 64 * 512 && kSplitW3 * 64 * 576 >= kSplitW1 * 32 * 256, "part size");
inline size_t trunk_col_floats(size_t frames) { return frames * (size_t)(81 * 512 > 49 * 576 ? 81 * 512 : 49 * 576); }

constexpr int kFastWgradMinFrames = 2048;
struct TrunkBwd {
  int Bn;              // frames
  const uint8_t* obs;  // [Bn][4][84][84] u8
  const float *a1, *a2;  // relu(conv1), relu(conv2), channel-last (ffnet_layout.h)
  const float* d_a3;   // [Bn][49][64] gradient w.r.t. relu(conv3), ALREADY masked by a3 > 0
  float *d_a2, *d_a1;  // scratch [Bn][81][64], [Bn][400][32]
  float* col;          // scratch, trunk_col_floats(Bn)
  float* part;         // scratch, kTrunkPartFloats
  float* cpart;        // scratch, kColsumBlocks * 512
  const float *w2p, *w3p;  // conv2 / conv3 weights in dgrad k order (permute_weights)
  float *g_c1w, *g_c1b, *g_c2w, *g_c2b, *g_c3w, *g_c3b;  // gradients, state_dict layout
  bool fast = false;   // the learner's bf16x2 mode: conv1's weight gradient on bf16 MFMA
  // the f32x3 mode: the weight-gradient GEMMs below (contraction over batch x positions) on the bf16 MFMA with three-part
  // operands, f32 accuracy (r4 at 512 frames, us f32 -> three-part: conv3 49 -> 35, conv2 63 -> 52, conv1 110 -> 102)
  bool emu = false;
  // Two-lane form (the Ape-X learner): conv3's and conv2's weight gradients, and the column sums of every tensor that
  // exists by then (the caller's pending jobs, d_a3, d_a2), run on `side` next to the data-gradient chain on the
  // caller's stream; the caller's stream waits for the side lane before trunk_backward returns.  Every kernel computes
  // what it computes on one lane (same grids, same summation orders): the results are bit-identical.
  hipStream_t side = nullptr;
  hipEvent_t ev_da3 = nullptr, ev_da2 = nullptr, ev_side = nullptr;
  float *part_side = nullptr, *cpart_side = nullptr;  // the side lane's own `part` / `cpart`
  const void *frag2 = nullptr, *frag3 = nullptr;  // dgrad weight fragments packed ahead (else per call, into `part`)
  const float* s32 = nullptr;                     // (side lane) head_bias_grad after the column sums
  float *g_a_b = nullptr, *g_v_b = nullptr;
  int A = 0;
};
// `to` continues only after everything queued on `from` so far
inline void lane_dep(hipEvent_t ev, hipStream_t from, hipStream_t to) {
  (void)hipEventRecord(ev, from);
  (void)hipStreamWaitEvent(to, ev, 0);
}

// all queued jobs in one launch pair; cpart must hold kColsumBlocks * (sum of the jobs' C) floats
inline void colsum_multi_launch(ColsumJobs& jobs, float* cpart, hipStream_t s) {
  if (jobs.n == 0) return;
  int maxC = 0;
  size_t off = 0;
  for (int j = 0; j < jobs.n; ++j) {
    jobs.part[j] = cpart + off;
    off += (size_t)kColsumBlocks * jobs.C[j];
    maxC = jobs.C[j] > maxC ? jobs.C[j] : maxC;
  }
  ProfScope prof("learner_colsum", s);
  hipLaunchKernelGGL(colsum_partial_multi, dim3(kColsumBlocks, jobs.n), dim3(kLT), 0, s, jobs);
  hipLaunchKernelGGL(colsum_final_multi, dim3(maxC / 4, jobs.n), dim3(kColsumBlocks), 0, s, jobs);
}

inline void colsum_launch(const float* src, int64_t rows, int C, float* cpart, float* out, hipStream_t s) {
  ProfScope prof("learner_colsum", s);
  hipLaunchKernelGGL(colsum_partial, dim3(kColsumBlocks), dim3(kLT), 0, s, src, rows, C, cpart);
  hipLaunchKernelGGL(colsum_final, dim3(C / 4), dim3(kColsumBlocks), 0, s, (const float*)cpart, C, out);
}

// `jobs`: the caller's pending column sums; the three bias gradients of the trunk are queued behind them and the
// whole queue goes out in one launch pair at the end (t.cpart: kColsumBlocks * (sum of all queued C) floats)
inline void trunk_backward(const TrunkBwd& t, hipStream_t s, ColsumJobs* pending = nullptr) {
  const int Bn = t.Bn;
  ColsumJobs own;
  ColsumJobs& jobs = pending ? *pending : own;
  const bool lanes = t.side != nullptr;
  hipStream_t sw = lanes ? t.side : s;  // the lane of conv3's / conv2's weight gradients
  float* partw = lanes ? t.part_side : t.part;
  if (lanes) lane_dep(t.ev_da3, s, sw);  // d_a3 (and everything the pending jobs read) is ready
  // conv2 / conv3 on bf16 MFMA only for many frames (R2D2: T * B): at 512 frames their per-block fixed costs (LDS
  // zero fill, 256 partial tiles of 128 / 144 KB for reduce_splits) cancel the gain (Ape-X: step 0.83 -> 0.91 ms)
  const bool fast23 = t.fast && Bn >= kFastWgradMinFrames;
  if (fast23) {  // conv3's weight gradient on bf16 MFMA (wgrad_conv3_bf16.h)
    int blocks = 0;
    {
      ProfScope prof("learner_wgrad_conv3", sw);
      (void)w3fast::launch(t.a2, t.d_a3, Bn, partw, sw, &blocks);
    }
    hipLaunchKernelGGL(reduce_splits, dim3(ceil_div(64 * 576, 256)), dim3(256), 0, sw, (const float*)partw, blocks, 64,
                       576, kRedConv3, t.g_c3w);
  } else {  // conv3: dW3, db3, d_a2
    ProbW3 p{};
    p.M = 64, p.N = 576, p.K = Bn * 49;
    p.d_out = t.d_a3, p.in = t.a2, p.part = partw;
    // (a few hundred frames: 3 x the splits = 3-4 blocks per CU instead of one; a block alone on its CU waits out
    // every chunk's load latency with two waves per SIMD)
    static const int mul = 3  /* (<= 8: what kTrunkPartFloats holds; flat between 2 and 4 in the r3 sweep) */;
    const int split3 = Bn <= 1024 ? kSplitW3 * mul : kSplitW3;
    if (t.fast && gemm_bf16x3_on()) (void)gemm3::launch_gemm<Tile3W64>(p, split3, sw, "learner_wgrad_conv3");
    else if (t.emu) (void)gemm3::launch_gemm<Tile6W64>(p, split3, sw, "learner_wgrad_conv3");
    else launch_gemm<TileW64>(p, split3, sw, "learner_wgrad_conv3");
    hipLaunchKernelGGL(reduce_splits, dim3(ceil_div(64 * 576, 256)), dim3(256), 0, sw, (const float*)partw, split3, 64,
                       576, kRedConv3, t.g_c3w);
  }
  jobs.add(t.d_a3, (int64_t)Bn * 49, 64, t.g_c3b);
  if (t.fast) {  // transposed convolution on bf16 MFMA, ReLU mask fused, no column buffer (dgrad_conv_bf16.h)
    ProfScope prof("learner_dgrad_conv3", s);
    (void)dgfast::launch_conv3(t.d_a3, t.w3p, t.a2, t.d_a2, Bn, t.part, s, t.frag3);
  } else {
    ProbConvDgrad p{};
    p.M = Bn * 49, p.N = 576, p.K = 64;
    p.d_out = t.d_a3, p.wp = t.w3p, p.col = t.col;
    launch_gemm<TileDgrad>(p, 1, s, "learner_dgrad_conv3");  // (f32x3: K = 64 -- the three-part GEMM measured 89 against 62 us)
    ProfScope prof("learner_col2im", s);
    hipLaunchKernelGGL(col2im3, dim3(ceil_div((int64_t)Bn * 81 * 16, 256)), dim3(256), 0, s, (const float*)t.col, t.a2,
                       t.d_a2, Bn);
  }
  if (lanes) lane_dep(t.ev_da2, s, sw);  // d_a2 is ready
  if (fast23) {  // conv2's weight gradient on bf16 MFMA (wgrad_conv2_bf16.h)
    int blocks = 0;
    {
      ProfScope prof("learner_wgrad_conv2", sw);
      (void)w2fast::launch(t.a1, t.d_a2, Bn, partw, sw, &blocks);
    }
    hipLaunchKernelGGL(reduce_splits, dim3(ceil_div(64 * 512, 256)), dim3(256), 0, sw, (const float*)partw, blocks, 64,
                       512, kRedConv2, t.g_c2w);
  } else {  // conv2: dW2, db2, d_a1
    ProbW2 p{};
    p.M = 64, p.N = 512, p.K = Bn * 81;
    p.d_out = t.d_a2, p.in = t.a1, p.part = partw;
    static const int mul = 3  /* (<= 8: what kTrunkPartFloats holds; flat between 2 and 4 in the r3 sweep) */;
    const int split2 = Bn <= 1024 ? kSplitW2 * mul : kSplitW2;
    if (t.fast && gemm_bf16x3_on()) (void)gemm3::launch_gemm<Tile3W64>(p, split2, sw, "learner_wgrad_conv2");
    else if (t.emu) (void)gemm3::launch_gemm<Tile6W64>(p, split2, sw, "learner_wgrad_conv2");
    else launch_gemm<TileW64>(p, split2, sw, "learner_wgrad_conv2");
    hipLaunchKernelGGL(reduce_splits, dim3(ceil_div(64 * 512, 256)), dim3(256), 0, sw, (const float*)partw, split2, 64,
                       512, kRedConv2, t.g_c2w);
  }
  jobs.add(t.d_a2, (int64_t)Bn * 81, 64, t.g_c2b);
  if (lanes) {  // the side lane ends here: the column sums of everything but d_a1, then the head biases
    colsum_multi_launch(jobs, t.cpart_side, sw);
    if (t.s32) hipLaunchKernelGGL(head_bias_grad, dim3(1), dim3(32), 0, sw, t.s32, t.A, t.g_a_b, t.g_v_b);
    jobs.n = 0;
  }
  if (t.fast) {
    ProfScope prof("learner_dgrad_conv2", s);
    (void)dgfast::launch_conv2(t.d_a2, t.w2p, t.a1, t.d_a1, Bn, t.part, s, t.frag2);
  } else {
    ProbConvDgrad p{};
    p.M = Bn * 81, p.N = 512, p.K = 64;
    p.d_out = t.d_a2, p.wp = t.w2p, p.col = t.col;
    launch_gemm<TileDgrad>(p, 1, s, "learner_dgrad_conv2");  // (f32x3: 88 against 77 us, as above)
    ProfScope prof("learner_col2im", s);
    hipLaunchKernelGGL(col2im2, dim3(ceil_div((int64_t)Bn * 400 * 8, 256)), dim3(256), 0, s, (const float*)t.col, t.a1,
                       t.d_a1, Bn);
  }
  if (t.fast) {  // conv1 on bf16 MFMA (wgrad_conv1_bf16.h): exact u8 frames x (hi + lo) gradients
    int blocks = 0;
    {
      ProfScope prof("learner_wgrad_conv1", s);
      (void)w1fast::launch(t.obs, t.d_a1, Bn, t.part, s, &blocks, lanes ? 128 : w1fast::kMaxBlocks);
    }
    hipLaunchKernelGGL(reduce_splits, dim3(ceil_div(32 * 256, 256)), dim3(256), 0, s, (const float*)t.part, blocks, 32,
                       256, kRedConv1, t.g_c1w);
  } else {  // conv1: dW1, db1 (no gradient flows into the frames)
    ProbW1 p{};
    p.M = 32, p.N = 256, p.K = Bn * 400;
    p.d_out = t.d_a1, p.obs = t.obs, p.part = t.part;
    if (t.emu) (void)gemm3::launch_gemm<Tile6W32>(p, kSplitW1, s, "learner_wgrad_conv1");
    else launch_gemm<TileW32>(p, kSplitW1, s, "learner_wgrad_conv1");
    hipLaunchKernelGGL(reduce_splits, dim3(ceil_div(32 * 256, 256)), dim3(256), 0, s, (const float*)t.part, kSplitW1, 32,
                       256, kRedConv1, t.g_c1w);
  }
  jobs.add(t.d_a1, (int64_t)Bn * 400, 32, t.g_c1b);
  colsum_multi_launch(jobs, t.cpart, s);
  if (lanes) lane_dep(t.ev_side, sw, s);
}

// ---- clip_grad_norm_ + optimiser over flat parameter / gradient / state buffers -------------------
struct OptimState {
  int optimizer = 0;  // 0 RMSprop (torch defaults alpha 0.99), 1 Adam (betas 0.9 / 0.999)
  float lr = 0.f, eps = 0.f, clip = 0.f;
  int64_t adam_t = 0;
};
inline void optimizer_apply(OptimState& o, float* P, const float* G, float* S1, float* S2, int64_t n, double* npart,
                            float* norm, hipStream_t s, const unsigned* abort_word = nullptr) {
  ProfScope prof("learner_optimizer", s);
  hipLaunchKernelGGL(sumsq_partial, dim3(kNormBlocks), dim3(256), 0, s, G, n, npart);
  hipLaunchKernelGGL(clip_coef, dim3(1), dim3(256), 0, s, (const double*)npart, kNormBlocks, o.clip, norm, abort_word);
  if (o.optimizer == 0) {
    hipLaunchKernelGGL(rmsprop_update, dim3(ceil_div(n / 4, 256)), dim3(256), 0, s, P, G, S1, n / 4, o.lr, 0.99f, o.eps,
                       (const float*)norm);
  } else {
    o.adam_t += 1;
    const float b1 = 0.9f, b2 = 0.999f;
    const float bc1 = 1.0f - (float)pow((double)b1, (double)o.adam_t);
    const float bc2s = (float)sqrt(1.0 - pow((double)b2, (double)o.adam_t));
    hipLaunchKernelGGL(adam_update, dim3(ceil_div(n / 4, 256)), dim3(256), 0, s, P, G, S1, S2, n / 4, o.lr, b1, b2, o.eps,
                       bc1, bc2s, (const float*)norm);
  }
}

}  // namespace
}  // namespace rela_amd
