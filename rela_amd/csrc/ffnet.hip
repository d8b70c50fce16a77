// ffnet.hip -- AtariFFNet forward (pyrela/net.py:8-55) as hand-written CDNA4 kernels.
//
// s u8[N,4,84,84] -> conv 8x8s4 (4->32) -> conv 4x4s2 (32->64) -> conv 3x3s1 (64->64)
//   -> fc 3136->512 -> {fc_v 512->1, fc_a 512->A} -> dueling (net.py:33-39).
// 9,352,704 MAC = 18.7 MFLOP per sample: MFMA-bound, not HBM-bound
// (660 FLOP per input byte), so every contraction runs on the matrix cores in exact fp32:
// v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered fmaf chain (guide: FP32-input MFMA), which
// keeps Q-values within fp32 round-off of the reference (tests use 1e-4 abs+rel).
//
// Formulation: implicit GEMM  Out[m = (sample, oy, ox)][n = out channel] = sum_k A[m][k] W[k][n]
//   MFMA A operand = im2col row, read straight out of an LDS copy of the input tile
//                    (no im2col matrix is ever materialised); lane (i = l&15, kk = l>>4)
//                    supplies A[row i][k0 + kk],
//   MFMA B operand = weights, pre-packed at load time in FRAGMENT ORDER
//                    Bfrag[(col_tile*KS + kstep)*64 + lane] so a wavefront's B load is one
//                    coalesced 256-byte read that stays L2-resident,
//   accumulators   = 16x16 tiles, 4 VGPRs each; one wave owns one 16-channel column tile
//                    and several row tiles, so each B fragment is reused across all of them.
// Activations are channel-last f32 ([N][pos][C]) between layers; fc weights are permuted
// at load time to match (k = pos*64 + c instead of torch's c*49 + pos).  LDS pixel strides
// are padded (33 / 66 floats) so the 16 rows of a fragment hit distinct banks.
// conv1 is special: its inputs are u8 frames, which are EXACT in bf16, and an fp32 weight is
// exactly the sum of three bf16 pieces (24-bit significand = 3 x 8).  So conv1 runs on
// v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate) as three passes hi/mid/lo with fp32
// accumulation: every product is exact and the result is fp32-accurate, while the layer drops
// from MFMA-bound to HBM-bound (28 KB in + 51 KB out per sample).  The /255 of net.py:46 is
// folded into conv1's weights at load time.
#include <atomic>
#include <cstdlib>

#include "common.h"
#include "gemm_bf16s.h"
#include "gemm_f32emu.h"
#include "gemm_s3.h"
#include "conv_img_s3.h"
#include "conv12_s3.h"
#include "ffnet_layout.h"
#include "gemm_lds.h"
#include "prof.h"

namespace rela_amd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FFNetDev {
  uint4* B1 = nullptr;                 // conv1 bf16 frags [piece 3][ct 2][ks 8][lane 64] x 8 bf16
  float* b1 = nullptr;                 // conv1 bias[32]
  float *B2 = nullptr, *b2 = nullptr;  // conv2 frags [4][128][64], bias[64]
  float *B3 = nullptr, *b3 = nullptr;  // conv3 frags [4][144][64], bias[64]
  float *Bf = nullptr, *bf = nullptr;  // fc    frags [32][784][64], bias[512]
  float* BfT = nullptr;                // fc    weights [3136 (k = pos*64+c)][512] for the small-batch split-K path
  float *Bh = nullptr, *bh = nullptr;  // heads frags [2][128][64], bias[32]  (cols 0..A-1 = fc_a, col 31 = fc_v)
  // split-bf16 fast path: [ct][ks][hi, lo][lane] x 8 bf16 (see "Split-bf16" below)
  uint4 *B2f = nullptr, *B3f = nullptr, *Bff = nullptr;
  float* Bhp = nullptr;  // heads weights [ct 2][wave 4][j 32][lane 64] in the k order of heads_duel
  // conv1 for the int8 matrix cores (conv12_i8): digits [hi, mid, lo][ct 2][tap 4][lane 64] x 16 int8, the channels'
  // scales s_c and the biases b_c + 128 s_c sum_k q_k
  uint4* W1d = nullptr;
  float *s1q = nullptr, *b1q = nullptr;
  // f32-accurate mode on the bf16 matrix cores (gemm_f32emu.h): bf16 triples in fragment order [cg][ks][u][part][lane]
  uint4 *B2e = nullptr, *B3e = nullptr, *Bfe = nullptr;
};

namespace {

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;

template <int CIN_, int IH_, int IW_, int KH_, int KW_, int STRIDE_, int OH_, int OW_, int OC_, int S_, bool U8_>
struct ConvCfg {
  static constexpr int CIN = CIN_, IH = IH_, IW = IW_, KH = KH_, KW = KW_, STRIDE = STRIDE_, OH = OH_, OW = OW_,
                       OC = OC_, S = S_;
  static constexpr bool U8 = U8_;
  static constexpr int P = OH * OW;
  static constexpr int M = S * P;
  static constexpr int RT = (M + 15) / 16;
  static constexpr int CT = OC / 16;
  static constexpr int RG = kWaves / CT;
  static constexpr int RPW = (RT + RG - 1) / RG;
  static constexpr int K = CIN * KH * KW;
  static constexpr int KS = K / 4;
  // LDS image of one f32 sample: pixel stride PIX floats, row stride ROW, sample stride SAMP, padded
  // so that the dword address of output position p (flattened over the samples of the block) is
  // 2*p + const (mod 32): the 16 rows x 2 k-lanes of a ds_read_b32 group then hit 32 distinct banks.
  //   STRIDE*PIX = 2 (mod 32);  STRIDE*ROW = 2*OW (mod 32);  SAMP = 2*OH*OW (mod 32)
  static constexpr int PIX = U8 ? 0 : (CIN + (STRIDE == 2 ? 1 : 2));  // floats per LDS pixel
  static constexpr int ROW_RAW = IW * PIX;
  static constexpr int ROW_NEED = STRIDE == 2 ? OW : 2 * OW;            // ROW mod (STRIDE == 2 ? 16 : 32)
  static constexpr int ROW_MOD = STRIDE == 2 ? 16 : 32;
  static constexpr int ROW = U8 ? 0 : ROW_RAW + ((ROW_NEED - ROW_RAW % ROW_MOD) % ROW_MOD + ROW_MOD) % ROW_MOD;
  static constexpr int SAMP_RAW = IH * ROW;
  static constexpr int SAMP = U8 ? 0 : SAMP_RAW + ((2 * OH * OW - SAMP_RAW) % 32 + 32) % 32;
  static constexpr int IN_ELEMS = CIN * IH * IW;                        // per sample
  static constexpr int LDS_BYTES = U8 ? S * IN_ELEMS : S * SAMP * 4;
  static constexpr bool BSTAT = !U8 && K / 4 <= 128;  // weight column tile fits the register budget
};

// Samples per block: conv2 keeps ONE sample resident (53 KB of LDS -> 3 blocks per CU, so one block's
// staging phase overlaps the others' MFMA phase; measured 458 -> 388 us per 6400 against S = 2 despite
// 81 -> 96 row padding); conv3 keeps four (S = 2 measured the same).
using Conv2 = ConvCfg<32, 20, 20, 4, 4, 2, 9, 9, 64, 1, false>;
using Conv3 = ConvCfg<64, 9, 9, 3, 3, 1, 7, 7, 64, 3, false>;

template <class C>
__global__ __launch_bounds__(kThreads) void conv_mfma(const void* __restrict__ in_, const float* __restrict__ Bfrag,
                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                      int N) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * C::S;
  const int ns = min(C::S, N - n0);

  // ---- stage the input tile of S samples in LDS -----------------------------------------
  if constexpr (C::U8) {
    const uint4* src = reinterpret_cast<const uint4*>(static_cast<const uint8_t*>(in_) + (size_t)n0 * C::IN_ELEMS);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    constexpr int per = C::IN_ELEMS / 16;
    // all loads of the tile in flight before the first LDS write (one HBM round trip, not IT)
    constexpr int IT = (C::S * per + kThreads - 1) / kThreads;
    uint4 v[IT];
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int i = tid + j * kThreads;
      v[j] = (i < ns * per) ? src[i] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int i = tid + j * kThreads;
      if (i < C::S * per) dst[i] = v[j];
    }
  } else {
    const float4* src = reinterpret_cast<const float4*>(static_cast<const float*>(in_) + (size_t)n0 * C::IN_ELEMS);
    float* dst = reinterpret_cast<float*>(smem);
    constexpr int cq_n = C::CIN / 4;
    constexpr int per = C::IN_ELEMS / 4;
    // all loads of the tile in flight before the first LDS write (one HBM round trip, not IT)
    constexpr int IT = (C::S * per + kThreads - 1) / kThreads;
    float4 v[IT];
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int i = tid + j * kThreads;
      v[j] = (i < ns * per) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int i = tid + j * kThreads;
      if (i < C::S * per) {
        const int pixel = i / cq_n, cq = i - pixel * cq_n;
        const int smp = pixel / (C::IH * C::IW), pin = pixel - smp * (C::IH * C::IW);
        const int y = pin / C::IW, x = pin - y * C::IW;
        float* d = dst + smp * C::SAMP + y * C::ROW + x * C::PIX + cq * 4;
        d[0] = v[j].x;
        d[1] = v[j].y;
        d[2] = v[j].z;
        d[3] = v[j].w;
      }
    }
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  const int ct = wave % C::CT, rg = wave / C::CT;
  const int li = lane & 15, kk = lane >> 4;

  int abase[C::RPW];
#pragma unroll
  for (int t = 0; t < C::RPW; ++t) {
    const int m = (rg + t * C::RG) * 16 + li;
    const int mm = (m < C::M) ? m : 0;
    const int s = mm / C::P, pos = mm - s * C::P;
    const int oy = pos / C::OW, ox = pos - oy * C::OW;
    if constexpr (C::U8)
      abase[t] = s * C::IN_ELEMS + oy * C::STRIDE * C::IW + ox * C::STRIDE + kk;
    else
      abase[t] = (s * C::SAMP + oy * C::STRIDE * C::ROW + ox * C::STRIDE * C::PIX + kk) * 4;
  }

  f32x4 acc[C::RPW];
#pragma unroll
  for (int t = 0; t < C::RPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* bptr = Bfrag + (size_t)ct * C::KS * 64 + lane;

  if constexpr (C::U8) {
    // k = (c, kh, kw), kw fastest: one k-step = 4 consecutive bytes of one input row
    for (int c = 0; c < C::CIN; ++c) {
#pragma unroll 2
      for (int kh = 0; kh < C::KH; ++kh) {
#pragma unroll
        for (int q = 0; q < C::KW / 4; ++q) {
          const int ks = (c * C::KH + kh) * (C::KW / 4) + q;
          const int koff = c * C::IH * C::IW + kh * C::IW + q * 4;
          const float b = bptr[(size_t)ks * 64];
#pragma unroll
          for (int t = 0; t < C::RPW; ++t) {
            const float a = (float)smem[abase[t] + koff];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
          }
        }
      }
    }
  } else {
    // k = (kh, kw, c), c fastest: one k-step = 4 consecutive channels of one input pixel.
    // The B fragments of one (kh, kw) chunk are fetched (L2) a whole chunk ahead of their use.
    constexpr int CQ = C::CIN / 4, NCH = C::KH * C::KW;
    float bcur[CQ], bnext[CQ];
#pragma unroll
    for (int cq = 0; cq < CQ; ++cq) bcur[cq] = bptr[(size_t)cq * 64];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int kh = ch / C::KW, kw = ch - kh * C::KW;
      if (ch + 1 < NCH) {
#pragma unroll
        for (int cq = 0; cq < CQ; ++cq) bnext[cq] = bptr[(size_t)((ch + 1) * CQ + cq) * 64];
      }
#pragma unroll
      for (int cq = 0; cq < CQ; ++cq) {
        const int koff = (kh * C::ROW + kw * C::PIX + cq * 4) * 4;
#pragma unroll
        for (int t = 0; t < C::RPW; ++t) {
          const float a = *reinterpret_cast<const float*>(smem + abase[t] + koff);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bcur[cq], acc[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int cq = 0; cq < CQ; ++cq) bcur[cq] = bnext[cq];
    }
  }

  // ---- epilogue: bias + ReLU, channel-last store ------------------------------------------
  const int col = ct * 16 + li;
  const float bv = bias[col];
  const int mlim = ns * C::P;
#pragma unroll
  for (int t = 0; t < C::RPW; ++t) {
    const int rt = rg + t * C::RG;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = rt * 16 + kk * 4 + r;
      if (m < mlim) {
        const float v = acc[t][r] + bv;
        out[((size_t)n0 * C::P + m) * C::OC + col] = v > 0.f ? v : 0.f;
      }
    }
  }
}

// ---- B-stationary persistent variant for large batches -------------------------------------
// Ablation of conv_mfma at N = 6400 (conv2: 372 us with, 306 us without the per-k-step B-fragment
// loads) shows the weight fetches, not the LDS reads or the tile staging, stall the MFMA stream.
// One column tile of the weights is K x 16 floats = KS registers per lane (128 / 144), so here
// every wave keeps ITS column tile in registers for the whole launch and the block (one per CU,
// 2 waves per SIMD) walks over its share of the samples with a double-buffered LDS input tile:
// the global loads of group g+1 are issued before the k-loop of group g and written to the other
// buffer halfway through it.  The k-loop then touches only LDS (A) and registers (B).
constexpr int kNumCU = 256;
// Blocks of a PERSISTENT kernel (one block per CU, each walking its share of the frames): a grid of exactly 256 has no
// slack -- one CU held by another stream's small kernel (the replay's single-workgroup chain, a learner GEMM) when the
// launch starts delays one block by that kernel's whole duration, and with it the launch.  RELA_CU_RESERVE = r leaves r
// CUs out of such grids (default below; 0 = the full chip).
std::atomic<int> g_cu_reserve{-1};  // -1: not set (RELA_CU_RESERVE, else 0)
inline int persistent_blocks(int n) {
  int r = g_cu_reserve.load(std::memory_order_relaxed);
  if (r < 0) {
    const char* e = std::getenv("RELA_CU_RESERVE");
    r = e ? std::atoi(e) : 0;
    r = r < 0 ? 0 : (r > 128 ? 128 : r);
    g_cu_reserve.store(r, std::memory_order_relaxed);
  }
  return std::min(kNumCU - r, n);
}

template <class C>
__global__ __launch_bounds__(kThreads) void conv_mfma_bstat(const float* __restrict__ in,
                                                            const float* __restrict__ Bfrag,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ out, int N) {
  static_assert(!C::U8, "f32 channel-last input only");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int ct = wave % C::CT, rg = wave / C::CT;
  const int li = lane & 15, kk = lane >> 4;
  const int ngroups = (N + C::S - 1) / C::S;

  float breg[C::KS];
  {
    const float* bptr = Bfrag + (size_t)ct * C::KS * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) breg[ks] = bptr[(size_t)ks * 64];
  }

  int abase[C::RPW];
#pragma unroll
  for (int t = 0; t < C::RPW; ++t) {
    const int m = (rg + t * C::RG) * 16 + li;
    const int mm = (m < C::M) ? m : 0;
    const int sm = mm / C::P, pos = mm - sm * C::P;
    const int oy = pos / C::OW, ox = pos - oy * C::OW;
    abase[t] = (sm * C::SAMP + oy * C::STRIDE * C::ROW + ox * C::STRIDE * C::PIX + kk) * 4;
  }
  const int col = ct * 16 + li;
  const float bv = bias[col];

  // staging in two phases of IT2 float4 per thread (keeps the live set below 256 registers)
  constexpr int cq_n = C::CIN / 4;
  constexpr int per = C::IN_ELEMS / 4;
  constexpr int IT = (C::S * per + kThreads - 1) / kThreads;
  constexpr int IT2 = (IT + 1) / 2;
  float4 v[IT2];
  auto stage_load = [&](int g, int phase) {
    const int n0 = g * C::S;
    const int ns = min(C::S, N - n0);
    const float4* src = reinterpret_cast<const float4*>(in + (size_t)n0 * C::IN_ELEMS);
#pragma unroll
    for (int j = 0; j < IT2; ++j) {
      const int i = tid + (phase * IT2 + j) * kThreads;
      v[j] = (i < ns * per) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage_store = [&](int buf, int phase) {
    float* dst = reinterpret_cast<float*>(smem + buf * C::LDS_BYTES);
#pragma unroll
    for (int j = 0; j < IT2; ++j) {
      const int i = tid + (phase * IT2 + j) * kThreads;
      if (i < C::S * per) {
        const int pixel = i / cq_n, cq = i - pixel * cq_n;
        const int sm = pixel / (C::IH * C::IW), pin = pixel - sm * (C::IH * C::IW);
        const int y = pin / C::IW, x = pin - y * C::IW;
        float* d = dst + sm * C::SAMP + y * C::ROW + x * C::PIX + cq * 4;
        d[0] = v[j].x;
        d[1] = v[j].y;
        d[2] = v[j].z;
        d[3] = v[j].w;
      }
    }
  };

  int g = blockIdx.x;
  if (g >= ngroups) return;
  stage_load(g, 0);
  stage_store(0, 0);
  stage_load(g, 1);
  stage_store(0, 1);
  __syncthreads();
  int buf = 0;
  constexpr int CQ = C::CIN / 4, NCH = C::KH * C::KW;
  for (; g < ngroups; g += gridDim.x) {
    const bool has_next = g + (int)gridDim.x < ngroups;
    if (has_next) stage_load(g + gridDim.x, 0);
    const uint8_t* tile = smem + buf * C::LDS_BYTES;
    f32x4 acc[C::RPW];
#pragma unroll
    for (int t = 0; t < C::RPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int kh = ch / C::KW, kw = ch - kh * C::KW;
#pragma unroll
      for (int cq = 0; cq < CQ; ++cq) {
        const int koff = (kh * C::ROW + kw * C::PIX + cq * 4) * 4;
#pragma unroll
        for (int t = 0; t < C::RPW; ++t) {
          const float a = *reinterpret_cast<const float*>(tile + abase[t] + koff);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[ch * CQ + cq], acc[t], 0, 0, 0);
        }
      }
      // the other buffer was last read one group ago (barrier since): fill it while this group computes
      if (has_next) {
        if (ch == NCH / 3) {
          stage_store(buf ^ 1, 0);
          stage_load(g + gridDim.x, 1);
        } else if (ch == (2 * NCH) / 3) {
          stage_store(buf ^ 1, 1);
        }
      }
    }
    const int n0 = g * C::S;
    const int mlim = min(C::S, N - n0) * C::P;
#pragma unroll
    for (int t = 0; t < C::RPW; ++t) {
      const int rt = rg + t * C::RG;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = rt * 16 + kk * 4 + r;
        if (m < mlim) {
          const float o = acc[t][r] + bv;
          out[((size_t)n0 * C::P + m) * C::OC + col] = o > 0.f ? o : 0.f;
        }
      }
    }
    __syncthreads();
    buf ^= 1;
  }
}

// Picks the B-stationary kernel once every CU gets at least two sample groups.  conv3 stays on
// conv_mfma: its 144 + 20 accumulator registers spill under the 256-register budget and with one
// block per CU the 2134 groups of N = 6400 quantise to 9 rounds (measured 259 us against 227 us).
// Splitting K over two wave groups (72 weight registers + 40 accumulators per wave, halves summed
// through LDS) removes the spills of the hot loop but measured 318 us: with one block per CU two
// waves per SIMD do not hide the LDS latency of ten row tiles.  Tried and dropped in round 1.
template <class C>
void launch_conv(const float* in, const float* Bfrag, const float* bias, float* out, int N, hipStream_t s) {
  const int ngroups = ceil_div(N, C::S);
  note_launch(C::K == 512 ? "conv_mfma<Conv2> (f32)" : "conv_mfma<Conv3> (f32)");
  if (C::BSTAT && ngroups >= 2 * kNumCU)
    hipLaunchKernelGGL(conv_mfma_bstat<C>, dim3(kNumCU), dim3(kThreads), 2 * C::LDS_BYTES, s, in, Bfrag, bias, out, N);
  else
    hipLaunchKernelGGL(conv_mfma<C>, dim3(ngroups), dim3(kThreads), C::LDS_BYTES, s, (const void*)in, Bfrag, bias,
                       out, N);
}

// gemm_mfma block height: makespan estimate = rounds over the CUs x rows per block; 112-row blocks win
// when they save a round (fc at N = 6400: 232 blocks in one round instead of 400 in two)
inline bool prefer_bm112(int N, int colgroups, int bm_default) {
  auto cost = [&](int bm) {
    const int64_t blocks = (int64_t)ceil_div(N, bm) * colgroups;
    return ((blocks + kNumCU - 1) / kNumCU) * bm;
  };
  return cost(112) < cost(bm_default);
}

// ---- conv1 on bf16 MFMA, exact-weight split ------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Conv1B {
  static constexpr int S = 2, P = 400, M = S * P, RT = M / 16, OC = 32, CT = 2, KS = 8;
  static constexpr int RPW = (RT + kWaves - 1) / kWaves;  // 7 row tiles per wave (last ones masked)
  static constexpr int IN_ELEMS = 4 * 84 * 84;
  static constexpr int LDS_BYTES = S * IN_ELEMS;
  static constexpr int FRAG_UINT4 = 3 * CT * KS * 64;  // [piece][ct][ks][lane] x 8 bf16
};

// 8 consecutive input bytes -> 8 bf16 (exact): v_cvt_f32_ubyteN + v_perm_b32 taking the high halves
__device__ __forceinline__ uint4 u8x8_to_bf16x8(uint32_t d0, uint32_t d1) {
  auto pk = [](uint32_t d, int lo) -> uint32_t {
    const float f0 = (float)((d >> (8 * lo)) & 0xff), f1 = (float)((d >> (8 * lo + 8)) & 0xff);
    return __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
  };
  return make_uint4(pk(d0, 0), pk(d0, 2), pk(d1, 0), pk(d1, 2));
}

__global__ __launch_bounds__(kThreads) void conv1_bf16x3(const uint8_t* __restrict__ in,
                                                         const uint4* __restrict__ Bfrag,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int N) {
  using C = Conv1B;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * C::S;
  const int ns = min(C::S, N - n0);
  {
    const uint4* src = reinterpret_cast<const uint4*>(in + (size_t)n0 * C::IN_ELEMS);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    constexpr int per = C::IN_ELEMS / 16;
    // all loads of the tile in flight before the first LDS write (one HBM round trip, not IT)
    constexpr int IT = (C::S * per + kThreads - 1) / kThreads;
    uint4 v[IT];
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int i = tid + j * kThreads;
      v[j] = (i < ns * per) ? src[i] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int i = tid + j * kThreads;
      if (i < C::S * per) dst[i] = v[j];
    }
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;

  // k = (c, kh, kw): a k-step of 32 = 4 kernel rows x 8 columns of one channel; lane group g
  // owns kernel row 4*(ks&1) + g, its 8 k's are 8 CONSECUTIVE input bytes (4-byte aligned).
  int abase[C::RPW];
#pragma unroll
  for (int t = 0; t < C::RPW; ++t) {
    const int rt = wave + t * kWaves;
    const int m = ((rt < C::RT) ? rt : 0) * 16 + li;
    const int s = m / C::P, pos = m - s * C::P;
    const int oy = pos / 20, ox = pos - oy * 20;
    abase[t] = s * C::IN_ELEMS + (4 * oy + g) * 84 + 4 * ox;
  }
  f32x4 acc[C::RPW][C::CT];
#pragma unroll
  for (int t = 0; t < C::RPW; ++t)
#pragma unroll
    for (int c = 0; c < C::CT; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Tried in round 1 without gain (157 us either way): a register double buffer for the weight fragments
  // (146 VGPRs: loses the second block per CU, 206 us) and staging them per k-step through LDS.  The kernel
  // moves 511 MB per 6,400 samples (3.2 TB/s), most of it the f32 output in 64-byte partial rows.
  // A persistent weight-stationary form (96 weight registers + 52 accumulators per wave, double-buffered
  // u8 tile) spills under the 256-register budget and measured 357-431 us.
#pragma unroll 1
  for (int ks = 0; ks < C::KS; ++ks) {
    const int koff = (ks >> 1) * 84 * 84 + (ks & 1) * 4 * 84;
    bf16x8 b[3][C::CT];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int c = 0; c < C::CT; ++c)
        b[p][c] = __builtin_bit_cast(bf16x8, Bfrag[((p * C::CT + c) * C::KS + ks) * 64 + lane]);
#pragma unroll
    for (int t = 0; t < C::RPW; ++t) {
      const uint32_t* ap = reinterpret_cast<const uint32_t*>(smem + abase[t] + koff);
      const bf16x8 a = __builtin_bit_cast(bf16x8, u8x8_to_bf16x8(ap[0], ap[1]));
#pragma unroll
      for (int c = 0; c < C::CT; ++c) {
        // smallest piece first so the big one is added last.  Operands swapped (weights as the MFMA's A operand): a
        // lane ends up with FOUR CONSECUTIVE CHANNELS of one pixel -- one 16-byte store where the pixel-major tile
        // took four 4-byte ones (r4: the f32 output is two thirds of this kernel's HBM bytes)
        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[2][c], a, acc[t][c], 0, 0, 0);
        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[1][c], a, acc[t][c], 0, 0, 0);
        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[0][c], a, acc[t][c], 0, 0, 0);
      }
    }
  }

  const int mlim = ns * C::P;
#pragma unroll
  for (int t = 0; t < C::RPW; ++t) {
    const int rt = wave + t * kWaves;
    if (rt >= C::RT) continue;
    const int m = rt * 16 + li;  // this lane's pixel; its channels: 16 c + 4 g .. + 3
    if (m >= mlim) continue;
#pragma unroll
    for (int c = 0; c < C::CT; ++c) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c * 16 + 4 * g);
      f32x4 v = acc[t][c] + bv;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      *reinterpret_cast<f32x4*>(out + ((size_t)n0 * C::P + m) * C::OC + c * 16 + 4 * g) = v;
    }
  }
}

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float x) {
  uint32_t u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// conv1 weights (32,4,8,8)/255 -> three bf16 planes in MFMA 16x16x32 fragment order.
// w == hi + mid + lo exactly for every finite fp32 weight (each residual is exact in fp32).
// plane_major: k-step ks = kernel row ks of all four planes (lane group g = plane; unused since the half-frame kernels went in r4)
// its LDS image in; otherwise k runs (c, kh, kw) linearly (conv1_bf16x3).
__device__ __forceinline__ void pack_conv1_bf16x3_at(int idx, const float* __restrict__ w, uint16_t* __restrict__ frag, int plane_major) {
  if (idx >= 2 * 8 * 64 * 8) return;
  const int j = idx & 7, lane = (idx >> 3) & 63, ks = (idx >> 9) & 7, ct = idx >> 12;
  const int k = plane_major ? (lane >> 4) * 64 + ks * 8 + j : ks * 32 + (lane >> 4) * 8 + j;
  const int oc = ct * 16 + (lane & 15);
  const float v = w[oc * 256 + k] / 255.0f;
  const uint16_t hi = f32_to_bf16_rne(v);
  const float r1 = v - bf16_to_f32(hi);
  const uint16_t mid = f32_to_bf16_rne(r1);
  const float r2 = r1 - bf16_to_f32(mid);
  const uint16_t lo = f32_to_bf16_rne(r2);
  const int plane = 2 * 8 * 64 * 8;
  frag[idx] = hi;
  frag[plane + idx] = mid;
  frag[2 * plane + idx] = lo;
}
__global__ void pack_conv1_bf16x3(const float* __restrict__ w, uint16_t* __restrict__ frag, int plane_major) {
  pack_conv1_bf16x3_at((int)blockIdx.x * blockDim.x + threadIdx.x, w, frag, plane_major);
}

// =====================================================================================================
// Split-bf16 ("fast") path: conv2 / conv3 / fc on v_mfma_f32_16x16x32_bf16 (16x the f32 MFMA rate).
// Every activation and weight x is kept as TWO bf16 numbers, hi = bf16(x), lo = bf16(x - hi) (16 mantissa
// bits together), and a product is evaluated as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  with f32 accumulation:
// the dropped terms (a_lo*b_lo and the residuals beyond 16 bits) are ~2^-16 of |a||b| per product, against
// 2^-24 for the exact f32 path, which stays the parity mode (rela_ffnet_set_precision).  Activations travel
// between the layers in "split records": per pixel, C channels of hi (2 B each) followed by C channels of
// lo -- the same 4 bytes per element as f32, so HBM traffic is unchanged while the MFMA work drops 5.3x.
// =====================================================================================================
// Four CONSECUTIVE channels of ONE pixel -> the hi and lo halves of its LDS record.  This is what a lane holds when the
// MFMA is issued with the operands swapped (weights as the A operand: D[channel 4g + r][pixel li]): ReLU, two packed
// RNE conversions per half and TWO 8-byte LDS stores for four values, where the pixel-major tile needed eight 2-byte
// stores (the epilogues' LDS stores were a fifth of the frame time).
__device__ __forceinline__ void split_store_lds4(uint8_t* rec, int C, int ch0, f32x4 v) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ a = {v[0] > 0.f ? v[0] : 0.f, v[1] > 0.f ? v[1] : 0.f}, b = {v[2] > 0.f ? v[2] : 0.f, v[3] > 0.f ? v[3] : 0.f};
  const bf16x2_ ha = __builtin_convertvector(a, bf16x2_), hb = __builtin_convertvector(b, bf16x2_);
  const bf16x2_ la = __builtin_convertvector(a - __builtin_convertvector(ha, f32x2_), bf16x2_);
  const bf16x2_ lb = __builtin_convertvector(b - __builtin_convertvector(hb, f32x2_), bf16x2_);
  *reinterpret_cast<uint2*>(rec + ch0 * 2) = make_uint2(__builtin_bit_cast(uint32_t, ha), __builtin_bit_cast(uint32_t, hb));
  *reinterpret_cast<uint2*>(rec + C * 2 + ch0 * 2) =
      make_uint2(__builtin_bit_cast(uint32_t, la), __builtin_bit_cast(uint32_t, lb));
}

// Makes the compiler treat a resident weight fragment as consumed HERE: its s_waitcnt for the load lands before
// the persistent loop instead of at the first use inside it, where on later rounds the same counter value
// would wait for the output stores and prefetches of the round before.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pin_loaded(bf16x8& f) {
  u32x4 t = __builtin_bit_cast(u32x4, f);
  asm volatile("" : "+v"(t));
  f = __builtin_bit_cast(bf16x8, t);
}

// Two values of one channel (rows r, r + 1 of an accumulator tile) -> the hi and lo halves of their LDS records:
// two packed RNE conversions (v_cvt_pk_bf16_f32) and four 2-byte LDS stores, no cross-lane exchange and no wait
// (pairing lanes through ds_bpermute costs an LDS round trip per value, which serialised the epilogues).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_store_lds2(uint8_t* rec0, uint8_t* rec1, int C, int col, float v0, float v1) {
  const f32x2 v = {v0, v1};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  const f32x2 hf = __builtin_convertvector(h, f32x2);
  const bf16x2 l = __builtin_convertvector(v - hf, bf16x2);
  const u16x2 hb = __builtin_bit_cast(u16x2, h), lb = __builtin_bit_cast(u16x2, l);
  *reinterpret_cast<uint16_t*>(rec0 + col * 2) = hb[0];
  *reinterpret_cast<uint16_t*>(rec1 + col * 2) = hb[1];
  *reinterpret_cast<uint16_t*>(rec0 + C * 2 + col * 2) = lb[0];
  *reinterpret_cast<uint16_t*>(rec1 + C * 2 + col * 2) = lb[1];
}

// conv on split records, weight-stationary and persistent (one block per CU walks over its sample groups with
// a double-buffered LDS tile).  k = (tap, c): one MFMA k-step = 32 channels of one input pixel, i.e. lane
// group g reads the 16 bytes of channels 8g..8g+7 from the hi part and from the lo part of the pixel record.
// LDS pixel / row / sample strides (in 16-byte units) are padded so that the 16 lanes ds_read_b128 services
// together hit 16 different 16-byte bank groups: unit(p, g) = 2p + g (mod 16) over output positions p.
template <int CIN_, int IH_, int IW_, int KH_, int KW_, int STRIDE_, int OH_, int OW_, int S_, int Q_, int RQ_, int SQ_>
struct ConvFastCfg {
  static constexpr int CIN = CIN_, IH = IH_, IW = IW_, KH = KH_, KW = KW_, STRIDE = STRIDE_, OH = OH_, OW = OW_, S = S_;
  static constexpr int OC = 64, CT = 4, RG = kWaves / CT;
  static constexpr int P = OH * OW, M = S * P, RT = (M + 15) / 16, RPW = (RT + RG - 1) / RG;
  static constexpr int KSUB = CIN / 32;              // k-steps per tap
  static constexpr int KS = KH * KW * KSUB;          // k-steps of 32
  static constexpr int REC = CIN * 4;                // bytes per input pixel record
  static constexpr int Q = Q_, RQ = RQ_, SQ = SQ_;   // pixel / row / sample stride in 16-byte units
  static constexpr int DEPTH = (CIN == 32) ? 4 : 3;  // A fragment pairs in flight per wave (register ring)
  static constexpr int LDS_BYTES = S * SQ * 16;      // one input buffer
  static constexpr int OROW = OC * 4 + 16;  // staged record stride: the 16 pixels of a store spread over the banks
  static constexpr int OUT_BYTES = S * OH * OW * OROW;  // output records of one group (staged for coalesced stores)
  static constexpr int LDS_TOTAL = 2 * LDS_BYTES + 2 * OUT_BYTES + OC * 4;  // + one spare record
  static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
  static constexpr int IN_BYTES = IH * IW * REC;     // per sample in HBM
  static constexpr int V16 = S * IN_BYTES / 16;      // 16-byte chunks per group
  static constexpr int IT = (V16 + kThreads - 1) / kThreads, IT2 = (IT + 1) / 2;
  static_assert(REC / 16 <= Q, "pixel stride too small");
};
constexpr int kFastMinN = 1024;  // below this fc_bf16s has too few blocks and the f32 split-K fc is faster
constexpr int kFastTrunkMinN = 128;  // from here up the split-bf16 convolutions beat the f32 ones
// f32-accurate bf16 mode (precision 2: the f32x3 arithmetic of gemm_f32emu.h).  r5: from kEmuConvMinN rows the whole trunk
// runs on pre-split activations ("split3 records", gemm_s3.h): conv1 -> conv2 fused per frame (conv12_s3.h), conv3 as an
// image kernel with resident weights (conv_img_s3.h), fc as an LDS-DMA GEMM over records (gemm_s3.h); smaller batches
// run the exact f32 MFMA kernels (same accuracy, channel-last f32).  Byte offsets inside the records are 32-bit.
constexpr int kEmuConvMinN = 512, kEmuFcMinN = 512, kEmuMaxN = 80000;
// conv2: 20x20x32 -> 9x9x64, stride 2: 2*Q = 2, 2*RQ = 18 = 2 (mod 16)
using Conv2F = ConvFastCfg<32, 20, 20, 4, 4, 2, 9, 9, 1, 9, 185, 20 * 185>;
// conv3: 9x9x64 -> 7x7x64, stride 1: Q = 2, RQ = 14 (7 positions per row), SQ = 98 = 2 (mod 16)
using Conv3F = ConvFastCfg<64, 9, 9, 3, 3, 1, 7, 7, 2, 18, 174, 1570>;

constexpr int kStampFrames = 8, kStampPoints = 12;
template <class C, bool STAMPS = false>
__device__ __forceinline__ void conv_bf16s_body(const uint8_t* __restrict__ in, const uint4* __restrict__ Bfrag,
                                                const float* __restrict__ bias, uint8_t* __restrict__ out, int N, int bid,
                                                int nblk, unsigned long long* stamps = nullptr) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  int stamp_frame = 0;
  auto stamp = [&](int point) {  // (diagnostic build only: rela_ffnet_debug_conv3_stamps)
    if constexpr (STAMPS) {
      if (bid == 0 && (threadIdx.x == 0 || threadIdx.x == 448) && stamp_frame < kStampFrames)
        stamps[((threadIdx.x == 0 ? 0 : 1) * kStampFrames + stamp_frame) * kStampPoints + point] = __builtin_amdgcn_s_memtime();
    }
  };
  const int wave = tid >> 6, lane = tid & 63;
  const int ct = wave % C::CT, rg = wave / C::CT;
  const int li = lane & 15, g = lane >> 4;
  const int ngroups = (N + C::S - 1) / C::S;

  // weights of this wave's 16 output channels: [ks][hi, lo] fragments, resident for the whole launch
  bf16x8 bh[C::KS], bl[C::KS];
  {
    const uint4* bp = Bfrag + (size_t)ct * C::KS * 2 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      bh[ks] = __builtin_bit_cast(bf16x8, bp[(size_t)(ks * 2) * 64]);
      bl[ks] = __builtin_bit_cast(bf16x8, bp[(size_t)(ks * 2 + 1) * 64]);
    }
  }
  int abase[C::RPW];
#pragma unroll
  for (int t = 0; t < C::RPW; ++t) {
    const int m = (rg + t * C::RG) * 16 + li;
    const int mm = (m < C::M) ? m : 0;
    const int sm = mm / C::P, pos = mm - sm * C::P;
    const int oy = pos / C::OW, ox = pos - oy * C::OW;
    abase[t] = (sm * C::SQ + oy * C::STRIDE * C::RQ + ox * C::STRIDE * C::Q + g) * 16;
  }
  const int ch0 = ct * 16 + 4 * g;  // this lane's four output channels (operands swapped: channels are the tile rows)
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + ch0);

  uint4 v[C::IT2];
  auto stage_load = [&](int grp, int phase) {
    const int n0 = grp * C::S;
    const int ns = min(C::S, N - n0);
    const uint4* src = reinterpret_cast<const uint4*>(in + (size_t)n0 * C::IN_BYTES);
#pragma unroll
    for (int j = 0; j < C::IT2; ++j) {
      const int i = tid + (phase * C::IT2 + j) * kThreads;
      v[j] = (i < ns * (C::IN_BYTES / 16)) ? src[i] : make_uint4(0, 0, 0, 0);
    }
  };
  auto stage_store = [&](int buf, int phase) {
    uint8_t* dst = smem + buf * C::LDS_BYTES;
    constexpr int UPP = C::REC / 16;  // 16-byte units per pixel record
#pragma unroll
    for (int j = 0; j < C::IT2; ++j) {
      const int i = tid + (phase * C::IT2 + j) * kThreads;
      if (i < C::V16) {
        const int pixel = i / UPP, u = i - pixel * UPP;
        const int sm = pixel / (C::IH * C::IW), pin = pixel - sm * (C::IH * C::IW);
        const int y = pin / C::IW, x = pin - y * C::IW;
        *reinterpret_cast<uint4*>(dst + (size_t)(sm * C::SQ + y * C::RQ + x * C::Q + u) * 16) = v[j];
      }
    }
  };

  int grp = bid;
  if (grp >= ngroups) return;
  stage_load(grp, 0);
  stage_store(0, 0);
  stage_load(grp, 1);
  stage_store(0, 1);
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    pin_loaded(bh[ks]);
    pin_loaded(bl[ks]);
  }
  __syncthreads();
  constexpr int LO = C::CIN * 2;  // byte offset of the lo part inside a pixel record
  uint8_t* const obase = smem + 2 * C::LDS_BYTES;
  uint8_t* const spare = obase + 2 * C::OUT_BYTES;  // rows past the group's last pixel land here
  // A row group whose LAST tile lies wholly past the group's pixels (conv3: 98 pixels = 7 tiles over RG = 2 row groups
  // of RPW = 4) skips that tile's fragment reads and MFMAs (wave-uniform): the two waves of a SIMD belong to different
  // row groups, so the SIMD issues 7 tiles' MFMAs where it issued 8.
  constexpr int RT_REAL = (C::M + 15) / 16;
  static_assert(C::RPW >= 2 && (RT_REAL - (C::RG - 1) + C::RG - 1) / C::RG >= C::RPW - 1, "at most the last tile of a row group is empty");
  const bool full = (RT_REAL - __builtin_amdgcn_readfirstlane(rg) + C::RG - 1) / C::RG >= C::RPW;
  {
    constexpr int NT = C::RPW;
    int buf = 0;
    int prev_n0 = 0, prev_nv = 0;  // the group whose records wait in the other output buffer, its 16-byte chunks
    constexpr int OV = C::S * C::P * (C::OC * 4 / 16), OIT = (OV + kThreads - 1) / kThreads;
    static_assert(OIT <= 4, "named chunk registers below");
    uint4 oc0, oc1, oc2, oc3;
    // chunk j of the previous group's records: read from LDS behind pair RD(j), stored to HBM behind pair RD(j) + 2 --
    // in the shadow of this group's MFMAs (after the loop, on all eight waves at once, the copy took 8 % of a group)
    auto o_read = [&](int j) {
      const uint8_t* ot = obase + (buf ^ 1) * C::OUT_BYTES;
      const int i = min(tid + j * kThreads, OV - 1);
      return *reinterpret_cast<const uint4*>(ot + (i >> 4) * C::OROW + (i & 15) * 16);
    };
    auto o_write = [&](int j, const uint4& val) {
      const int i = tid + j * kThreads;
      if (i < prev_nv) reinterpret_cast<uint4*>(out + (size_t)prev_n0 * C::P * (C::OC * 4))[i] = val;
    };
    for (; grp < ngroups; grp += nblk) {
      const bool has_next = grp + nblk < ngroups;
      stamp(0);
      if (has_next) stage_load(grp + nblk, 0);
      const uint8_t* tile = smem + buf * C::LDS_BYTES;
      f32x4 acc[NT];  // starts at the bias
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = bv;
      // A fragments run D (hi, lo) pairs ahead of the MFMAs that consume them, in a register ring over the
      // flattened (k-step, row tile) sequence: with two waves per SIMD nothing else hides the LDS latency.
      constexpr int TOT = C::KS * NT, D = C::DEPTH;
      uint4 ah[D], al[D];
      auto a_issue = [&](int idx, int slot) {
        const int ks = idx / NT, t = idx - ks * NT;
        const int tap = ks / C::KSUB, sub = ks - tap * C::KSUB;
        const int kh = tap / C::KW, kw = tap - kh * C::KW;
        const uint8_t* ap = tile + abase[t] + (kh * C::RQ + kw * C::Q) * 16 + sub * 64;
        ah[slot] = *reinterpret_cast<const uint4*>(ap);
        al[slot] = *reinterpret_cast<const uint4*>(ap + LO);
      };
#pragma unroll
      for (int i = 0; i < D; ++i)
        if (i % NT < NT - 1 || full) a_issue(i, i);
      __builtin_amdgcn_sched_barrier(0);
      static_assert(3 + 4 * (OIT - 1) < TOT, "copy-out slots inside the loop");
      // (two loops of TOT / 2: as ONE loop the body is too large for the unroller's full-unroll budget, it unrolls by
      // half the trip count instead, k-steps become run-time indices and the resident fragments move to scratch memory)
      auto pair = [&](int idx) {
        const int ks = idx / NT, t = idx - ks * NT;
        const int slot = idx % D;
        if (t < NT - 1 || full) {
          const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[slot]);
          const bf16x8 xl = __builtin_bit_cast(bf16x8, al[slot]);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[ks], xl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[ks], xh, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[ks], xh, acc[t], 0, 0, 0);
        }
        if (idx + D < TOT && ((idx + D) % NT < NT - 1 || full)) a_issue(idx + D, slot);
        if (idx == 1) oc0 = o_read(0);
        if (idx == 3) o_write(0, oc0);
        if (OIT > 1 && idx == 5) oc1 = o_read(1);
        if (OIT > 1 && idx == 7) o_write(1, oc1);
        if (OIT > 2 && idx == 9) oc2 = o_read(2);
        if (OIT > 2 && idx == 11) o_write(2, oc2);
        if (OIT > 3 && idx == 13) oc3 = o_read(3);
        if (OIT > 3 && idx == 15) o_write(3, oc3);
        __builtin_amdgcn_sched_barrier(0);  // keep the reads where they are (the scheduler sinks them to their use)
        // the other buffer was last read one group ago (barrier since): fill it while this group computes
        if (has_next) {
          if (idx == TOT / 3) {
            stage_store(buf ^ 1, 0);
            stage_load(grp + nblk, 1);
          } else if (idx == (2 * TOT) / 3) {
            stage_store(buf ^ 1, 1);
          }
        }
            };
      static_assert(TOT % 2 == 0, "two halves");
#pragma unroll
      for (int idx = 0; idx < TOT / 2; ++idx) pair(idx);
#pragma unroll
      for (int idx = TOT / 2; idx < TOT; ++idx) pair(idx);
      // epilogue: bias + ReLU + hi/lo split into an LDS copy of the group's output records; they leave for HBM as
      // coalesced 16-byte-per-lane copies during the NEXT group's MFMA loop (a wave's own 16 channels are only 32
      // contiguous bytes per pixel).  The output tile is double buffered like the input, so ONE barrier per group
      // orders everything: it publishes this group's records and the next group's staged input, and the records of
      // two groups ago were read out before the barrier in between.
      stamp(1);
      uint8_t* otile = obase + buf * C::OUT_BYTES;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int m = (rg + t * C::RG) * 16 + li;  // this lane's pixel of the tile
        split_store_lds4((m < C::M) ? otile + (size_t)m * C::OROW : spare, C::OC, ch0, acc[t]);
      }
      stamp(2);
      __syncthreads();
      stamp(3);
      prev_n0 = grp * C::S;
      prev_nv = min(C::S, N - prev_n0) * C::P * (C::OC * 4 / 16);
      stamp(4);
      stamp_frame += 1;
      buf ^= 1;
    }
    {  // the last group's records
      const uint8_t* ot = obase + (buf ^ 1) * C::OUT_BYTES;
      uint4* dst = reinterpret_cast<uint4*>(out + (size_t)prev_n0 * C::P * (C::OC * 4));
      for (int i = tid; i < prev_nv; i += kThreads)
        dst[i] = *reinterpret_cast<const uint4*>(ot + (i >> 4) * C::OROW + (i & 15) * 16);
    }
  }
}

template <class C>
__global__ __launch_bounds__(kThreads) void conv_bf16s(const uint8_t* __restrict__ in, const uint4* __restrict__ Bfrag,
                                                       const float* __restrict__ bias, uint8_t* __restrict__ out,
                                                       int N) {
  conv_bf16s_body<C>(in, Bfrag, bias, out, N, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(kThreads) void conv3_bf16s_stamps(const uint8_t* __restrict__ in, const uint4* __restrict__ Bfrag,
                                                               const float* __restrict__ bias, uint8_t* __restrict__ out,
                                                               int N, unsigned long long* stamps) {
  conv_bf16s_body<Conv3F, true>(in, Bfrag, bias, out, N, blockIdx.x, gridDim.x, stamps);
}

// ---- conv1 on the INT8 matrix cores, whole frame at a time (r3) ----------------------------------------------------
// The frames are u8, so conv1 needs no floating-point operand at all: the weights of an output channel c become
// 24-bit fixed point, w = s_c * q with |q| <= 127 * 65536 + 127 * 256 + 127 and q = 65536 d_hi + 256 d_mid + d_lo in
// balanced base-256 digits (int8 each), the pixels x - 128 (int8), and
//     conv1(x)[c] = s_c * (65536 S_hi + 256 S_mid + S_lo) + (b_c + 128 s_c sum_k q_k),   S_d = sum_k (x_k - 128) d_k
// with the three S exact in i32 (|S| <= 256 * 128 * 128 = 2^22).  v_mfma_i32_16x16x64_i8 runs at twice the bf16 rate:
// three digit products cost 0.75 of the two bf16 products (hi, mid) they replace -- and carry 24 bits of every weight
// relative to its channel's largest instead of 16, so conv1 of the split-bf16 mode is now within an f32 rounding or
// two of the exact mode and independent of summation order.  The float part is fixed (tests mirror it bit for bit):
//     u = f32(S_hi) * 65536 + f32(S_mid * 256 + S_lo);  y = u * s_c + b'_c      (the integer sum exact, no fused multiply-add)
//
// Layout.  K = 4 taps x 64: the 8 x 8 stride-4 kernel is a 2 x 2 stride-1 kernel over the 4 x 4 space-to-depth image
// (21 x 21 cells of [plane 4][yy 4][xx 4] bytes).  LDS holds the WHOLE frame as [plane][cell 441][16 B] (28 KB; the
// bf16 image of HALF a frame took 30): the B operand of tap (dy, dx) for output pixel (oy, ox) is one aligned
// ds_read_b128 at plane g, cell (oy + dy) * 21 + ox + dx -- 16 consecutive pixels of a tile read 16 consecutive cells,
// and the plane stride is 0 (mod 256), so the lanes a b128 pass serves ({li 0-3, 12-15 of g} + {li 4-11 of g + 1})
// cover the 64 banks once.  A thread stages a cell with four dword loads (rows 4 Y .. 4 Y + 3 of the plane; a wave
// reads 21-cell runs of 84 contiguous bytes), flips the sign bits and stores it with one ds_write_b128.
//
// Schedule (two barriers per frame where the half-frame kernel had five):
//   phase A  conv1: T1 -> T2 (conv2's padded input tile of split records); the previous frame's conv2 output tile O
//            leaves for HBM in the shadow of these MFMAs
//   phase B  conv2: T2 -> O; the next frame's cells are loaded at its start and stored into T1 in its second half
struct Conv12I {
  using C2 = Conv2F;
  static constexpr int GW = 21, NPIX = GW * GW, PLANE_ELEMS = 84 * 84, IN_ELEMS = 4 * PLANE_ELEMS;
  static constexpr int PLANE1 = 7168;  // 441 cells x 16 B, padded to a multiple of 256 B
  static_assert(PLANE1 >= NPIX * 16 && PLANE1 % 256 == 0, "plane stride");
  static constexpr int T1_BYTES = 4 * PLANE1, T2_BYTES = C2::LDS_BYTES, O_BYTES = C2::OUT_BYTES + 256;
  static constexpr int W1_UINT4 = 3 * 2 * 4 * 64;  // [digit hi, mid, lo][ct 2][tap 4][lane 64] x 16 int8
  static constexpr int LDS_TOTAL = T1_BYTES + T2_BYTES + O_BYTES + W1_UINT4 * 16;
  static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
  static constexpr int CELLS = 4 * NPIX;
  static constexpr int IT = (CELLS + kThreads - 1) / kThreads;  // 4 cells per thread
  static constexpr int RT = 25, RG = 4, RPW = 7;  // 400 pixels = 25 tiles of 16; wave (ct, rg) owns tiles rg + 4 t
  static constexpr int G0 = 3, G1 = 2, G2 = 2;    // three passes over the taps (tiles 0..2, 3..4, 5..6): 12 accumulator registers per tile
  static constexpr int OV16 = C2::P * (C2::OC * 4 / 16);  // 16-byte chunks of an output tile
  static constexpr int OIT = (OV16 + kThreads - 1) / kThreads;
};
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <bool JOBS, bool STAMPS = false>
__device__ __forceinline__ void conv12i_body(const uint8_t* __restrict__ in, const uint8_t* __restrict__ in1, int n_in0,
                                             const uint4* __restrict__ W1d, const float* __restrict__ scale1,
                                             const float* __restrict__ bias1q, const uint4* __restrict__ B2frag,
                                             const float* __restrict__ bias2, uint8_t* __restrict__ out,
                                             uint8_t* __restrict__ a1_out, int a1_lo, int n_a1, int N, int bid, int nblk,
                                             unsigned long long* stamps = nullptr) {
  using F = Conv12I;
  using C2 = Conv2F;
  int stamp_frame = 0;
  auto stamp = [&](int point) {
    if constexpr (STAMPS) {
      if (bid == 0 && (threadIdx.x == 0 || threadIdx.x == 448) && stamp_frame < kStampFrames)
        stamps[((threadIdx.x == 0 ? 0 : 1) * kStampFrames + stamp_frame) * kStampPoints + point] = __builtin_amdgcn_s_memtime();
    }
  };
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* t1 = smem;
  uint8_t* t2 = smem + F::T1_BYTES;
  uint8_t* otile = t2 + F::T2_BYTES;
  uint8_t* spare = otile + C2::OUT_BYTES;  // 256 B: rows past the last pixel of either layer land here
  uint4* w1s = reinterpret_cast<uint4*>(otile + F::O_BYTES);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;

  // ---- residents ----
  for (int i = tid; i < F::W1_UINT4; i += kThreads) w1s[i] = W1d[i];
  const int ct1 = wave & 1, rg1 = wave >> 1;
  const int ct2 = wave % C2::CT, rg2 = wave / C2::CT;
  bf16x8 bh[C2::KS], bl[C2::KS];
  {
    const uint4* bp = B2frag + (size_t)ct2 * C2::KS * 2 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < C2::KS; ++ks) {
      bh[ks] = __builtin_bit_cast(bf16x8, bp[(size_t)(ks * 2) * 64]);
      bl[ks] = __builtin_bit_cast(bf16x8, bp[(size_t)(ks * 2 + 1) * 64]);
    }
  }
  // operands swapped (weights as the A operand): a lane holds four consecutive channels of one pixel
  const int ch1 = ct1 * 16 + 4 * g, ch2 = ct2 * 16 + 4 * g;
  const f32x4 bv2 = *reinterpret_cast<const f32x4*>(bias2 + ch2);
  // conv1 B fragments: lane (li, g) reads plane g's cell of its pixel (+ the tap's cell offset)
  int a1base[F::RPW];
#pragma unroll
  for (int t = 0; t < F::RPW; ++t) {
    const int m = min((min(rg1 + t * F::RG, F::RT - 1)) * 16 + li, 399);
    const int oy = m / 20, ox = m - oy * 20;
    a1base[t] = g * F::PLANE1 + (oy * F::GW + ox) * 16;
  }
  int a2base[C2::RPW];
#pragma unroll
  for (int t = 0; t < C2::RPW; ++t) {
    const int m = (rg2 + t * C2::RG) * 16 + li;
    const int mm = (m < C2::M) ? m : 0;
    const int oy = mm / C2::OW, ox = mm - oy * C2::OW;
    a2base[t] = (oy * C2::STRIDE * C2::RQ + ox * C2::STRIDE * C2::Q + g) * 16;
  }

  // ---- staging of a frame's cells (unpredicated, clamped cell index) ----
  // (the cell addresses are derived per use from an opaque copy of tid: hoisted out of the frame loop they pin 8 registers)
  auto cell_of = [&](int j, int& goff, int& loff) {
    int tq = tid;
    asm volatile("" : "+v"(tq));
    const int c = min(tq + j * kThreads, F::CELLS - 1);
    const int pl = c / F::NPIX, P = c - pl * F::NPIX;
    const int Y = P / F::GW, X = P - Y * F::GW;
    goff = pl * F::PLANE_ELEMS + 4 * Y * 84 + 4 * X;
    loff = pl * F::PLANE1 + P * 16;
  };
  uint32_t st[F::IT][4];
  int st_loff[F::IT];  // (the cell's LDS address travels with its bytes from the load to the store)
  auto g_load1 = [&](int n, int j) {
    const uint8_t* src = (!JOBS || n < n_in0) ? in + (size_t)n * F::IN_ELEMS : in1 + (size_t)(n - n_in0) * F::IN_ELEMS;
    int goff;
    cell_of(j, goff, st_loff[j]);
#pragma unroll
    for (int r = 0; r < 4; ++r) st[j][r] = *reinterpret_cast<const uint32_t*>(src + goff + r * 84);
  };
  auto s_store = [&](int j) {  // x -> x - 128 as int8: flip the sign bits
    *reinterpret_cast<uint4*>(t1 + st_loff[j]) = make_uint4(st[j][0] ^ 0x80808080u, st[j][1] ^ 0x80808080u,
                                                            st[j][2] ^ 0x80808080u, st[j][3] ^ 0x80808080u);
  };

  // ---- conv1 over the tiles [T0, T0 + NT) of this wave: T1 -> split records in T2 ----
  // `hook(i)` runs after the MFMAs of pair i (the caller's copy-out slices)
  auto conv1_pass = [&](auto t0_tag, auto nt_tag, auto&& hook) {
    constexpr int T0 = decltype(t0_tag)::value, NT = decltype(nt_tag)::value;
    i32x4 s_hi[NT], s_mid[NT], s_lo[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) s_hi[t] = s_mid[t] = s_lo[t] = i32x4{0, 0, 0, 0};
    constexpr int TOT = 4 * NT, D = 4;
    uint4 x[D];
    auto a_issue = [&](int idx, int slot) {
      const int ks = idx / NT, t = T0 + idx - ks * NT;
      x[slot] = *reinterpret_cast<const uint4*>(t1 + a1base[t] + ((ks >> 1) * F::GW + (ks & 1)) * 16);
    };
    // this wave's digit fragments of the current tap: ONE register set (a second one spills: conv2's resident weights
    // hold 128 of the 256 registers); the next tap's are issued behind the last MFMAs that read these
    uint4 wd[3];
    auto w_issue = [&](int ks) {
#pragma unroll
      for (int d = 0; d < 3; ++d) wd[d] = w1s[((d * 2 + ct1) * 4 + ks) * 64 + lane];
    };
    w_issue(0);
#pragma unroll
    for (int i = 0; i < D; ++i) a_issue(i, i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int idx = ks * NT + t, slot = idx % D;
        const i32x4 xv = __builtin_bit_cast(i32x4, x[slot]);
        s_hi[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, wd[0]), xv, s_hi[t], 0, 0, 0);
        s_mid[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, wd[1]), xv, s_mid[t], 0, 0, 0);
        s_lo[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, wd[2]), xv, s_lo[t], 0, 0, 0);
        if (idx + D < TOT) a_issue(idx + D, slot);
        if (t == NT - 1 && ks + 1 < 4) w_issue(ks + 1);
        hook(idx);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // (scale and bias come from L1 per pass: kept across the frame loop they would cost eight registers)
    int chq = ch1;
    asm volatile("" : "+v"(chq));
    const f32x4 sc1 = *reinterpret_cast<const f32x4*>(scale1 + chq), bv1 = *reinterpret_cast<const f32x4*>(bias1q + chq);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int rt = rg1 + (T0 + t) * F::RG;
      if (rt >= F::RT) continue;  // (wave-uniform: only the group rg = 0 has a seventh tile)
      const int m = rt * 16 + li;
      const int y = m / 20, xx = m - y * 20;
      const i32x4 ml = s_mid[t] * 256 + s_lo[t];  // exact: |S_mid * 256 + S_lo| < 2^31
      const f32x4 u = __builtin_convertvector(s_hi[t], f32x4) * 65536.0f + __builtin_convertvector(ml, f32x4);
      const f32x4 v = u * sc1 + bv1;
      split_store_lds4((m < 400) ? t2 + (size_t)(y * C2::RQ + xx * C2::Q) * 16 : spare, 32, ch1, v);
    }
  };

  int n = bid;
  if (n >= N) return;
#pragma unroll
  for (int j = 0; j < F::IT; ++j) g_load1(n, j);
#pragma unroll
  for (int j = 0; j < F::IT; ++j) s_store(j);
#pragma unroll
  for (int ks = 0; ks < C2::KS; ++ks) {
    pin_loaded(bh[ks]);
    pin_loaded(bl[ks]);
  }
  __syncthreads();
  constexpr int LO = C2::CIN * 2;
  int prev = -1;  // the frame whose conv2 output waits in O
  // the output tile of frame `prev` -> HBM: chunk j is read from O after pair RD(j) of conv1's first pass and stored
  // after pair RD(j) + 2 (first frame: the reads return stale bytes and the stores are skipped)
  static_assert(F::OIT == 3, "three named chunk registers below (an array indexed inside the hook stays in scratch)");
  uint4 oc0, oc1, oc2;
  auto o_read = [&](int j) {
    const int i = min(tid + j * kThreads, F::OV16 - 1);
    return *reinterpret_cast<const uint4*>(otile + (i >> 4) * C2::OROW + (i & 15) * 16);
  };
  auto o_write = [&](int j, const uint4& v) {
    const int i = min(tid + j * kThreads, F::OV16 - 1);
    if (prev >= 0) reinterpret_cast<uint4*>(out + (size_t)prev * C2::P * (C2::OC * 4))[i] = v;
  };
  auto copy_hook = [&](int idx) {
    if (idx == 1) oc0 = o_read(0);
    if (idx == 3) o_write(0, oc0);
    if (idx == 5) oc1 = o_read(1);
    if (idx == 7) o_write(1, oc1);
    if (idx == 9) oc2 = o_read(2);
    if (idx == 11) o_write(2, oc2);
  };
  static_assert(3 + 4 * (F::OIT - 1) < 4 * F::G0 && F::G0 + F::G1 + F::G2 == F::RPW, "copy-out slots inside the first pass");
  for (; n < N; n += nblk) {
    const int nn = (n + nblk < N) ? n + nblk : n;  // (the last round re-stages its own frame)
    stamp(0);
    conv1_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, F::G0>{}, copy_hook);
    conv1_pass(std::integral_constant<int, F::G0>{}, std::integral_constant<int, F::G1>{}, [](int) {});
    conv1_pass(std::integral_constant<int, F::G0 + F::G1>{}, std::integral_constant<int, F::G2>{}, [](int) {});
    stamp(1);
    __syncthreads();  // T2 complete, T1 and O free
    stamp(2);
    if constexpr (JOBS) {
      if (a1_out && (unsigned)(n - a1_lo) < (unsigned)n_a1) {  // (block-uniform) conv1's records: [400 pixels][32 hi | 32 lo]
        uint4* dst = reinterpret_cast<uint4*>(a1_out + (size_t)n * (400 * 128));
        for (int i = tid; i < 400 * 8; i += kThreads) {
          const int px = i >> 3, u = i & 7;
          const int y = px / 20, x = px - y * 20;
          dst[i] = *reinterpret_cast<const uint4*>(t2 + (size_t)(y * C2::RQ + x * C2::Q + u) * 16);
        }
      }
    }
    // ---- conv2 from T2; the next frame's cells go into T1 in the second half ----
    {
      f32x4 acc[C2::RPW];
#pragma unroll
      for (int t = 0; t < C2::RPW; ++t) acc[t] = bv2;
      constexpr int TOT = C2::KS * C2::RPW, D = 3;
      uint4 ah[D], al[D];
      auto a_issue = [&](int idx, int slot) {
        const int ks = idx / C2::RPW, t = idx - ks * C2::RPW;
        const int kh = ks / C2::KW, kw = ks - kh * C2::KW;  // KSUB = 1: one k-step per tap
        const uint8_t* ap = t2 + a2base[t] + (kh * C2::RQ + kw * C2::Q) * 16;
        ah[slot] = *reinterpret_cast<const uint4*>(ap);
        al[slot] = *reinterpret_cast<const uint4*>(ap + LO);
      };
#pragma unroll
      for (int i = 0; i < D; ++i) a_issue(i, i);
      __builtin_amdgcn_sched_barrier(0);
      static_assert(TOT / 2 + 6 * (F::IT - 1) < TOT, "staging slots inside the loop");
#pragma unroll
      for (int idx = 0; idx < TOT; ++idx) {
        const int ks = idx / C2::RPW, t = idx - ks * C2::RPW;
        const int slot = idx % D;
        const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[slot]);
        const bf16x8 xl = __builtin_bit_cast(bf16x8, al[slot]);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[ks], xl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[ks], xh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[ks], xh, acc[t], 0, 0, 0);
        if (idx + D < TOT) a_issue(idx + D, slot);
        // the next frame's cells: loaded behind pairs 1, 4, 7, 10, stored (sign bits flipped) in the second half
#pragma unroll
        for (int j = 0; j < F::IT; ++j) {
          if (idx == 1 + 3 * j) g_load1(nn, j);
          if (idx == TOT / 2 + 6 * j) s_store(j);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      stamp(3);
#pragma unroll
      for (int t = 0; t < C2::RPW; ++t) {
        const int m = (rg2 + t * C2::RG) * 16 + li;
        split_store_lds4((m < C2::M) ? otile + (size_t)m * C2::OROW : spare, C2::OC, ch2, acc[t]);
      }
    }
    stamp(4);
    __syncthreads();  // O complete, T1 ready, T2 free
    stamp(5);
    stamp_frame += 1;
    prev = n;
  }
  {  // the last frame's output tile
    uint4* dst = reinterpret_cast<uint4*>(out + (size_t)prev * C2::P * (C2::OC * 4));
    for (int i = tid; i < F::OV16; i += kThreads)
      dst[i] = *reinterpret_cast<const uint4*>(otile + (i >> 4) * C2::OROW + (i & 15) * 16);
  }
}

__global__ __launch_bounds__(kThreads) void conv12_i8(const uint8_t* __restrict__ in, const uint4* __restrict__ W1d,
                                                      const float* __restrict__ scale1, const float* __restrict__ bias1q,
                                                      const uint4* __restrict__ B2frag, const float* __restrict__ bias2,
                                                      uint8_t* __restrict__ out, int N) {
  conv12i_body<false>(in, nullptr, N, W1d, scale1, bias1q, B2frag, bias2, out, nullptr, 0, 0, N, blockIdx.x, gridDim.x);
}

// Several trunk passes in one launch (the learner's three forwards: online over [s ; s'], target over s'): job j owns
// the blocks [block0, block0 + nblocks) and walks its own frames with its own weights.
struct TrunkJob {
  const uint8_t *in0, *in1;  // rows [0, n_in0) of in0, then rows of in1
  int n_in0;
  const uint4 *B2, *B3;  // conv2 / conv3 fragments of the job's net
  const float *b2, *b3;
  const uint4* W1d;  // conv1 for the int8 matrix cores (conv12_i8_jobs)
  const float *s1q, *b1q;
  uint8_t *a1_out;  // conv1's records [N][400][128 B], written for the rows [a1_lo, a1_lo + n_a1); or NULL
  int a1_lo, n_a1;
  uint8_t *a2, *a3;  // conv2's / conv3's split records [N][81][256 B] / [N][49][256 B]
  int N, block0, nblocks;
};
constexpr int kMaxTrunkJobs = 3;
struct TrunkJobs {
  TrunkJob j[kMaxTrunkJobs];
  int n;
};
__global__ __launch_bounds__(kThreads) void conv12_i8_stamps(const uint8_t* __restrict__ in, const uint4* __restrict__ W1d,
                                                             const float* __restrict__ scale1, const float* __restrict__ bias1q,
                                                             const uint4* __restrict__ B2frag, const float* __restrict__ bias2,
                                                             uint8_t* __restrict__ out, int N, unsigned long long* stamps) {
  conv12i_body<false, true>(in, nullptr, N, W1d, scale1, bias1q, B2frag, bias2, out, nullptr, 0, 0, N, blockIdx.x, gridDim.x, stamps);
}
__global__ __launch_bounds__(kThreads) void conv12_i8_jobs(TrunkJobs jobs) {
  int k = 0;
  while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.j[k + 1].block0) ++k;
  const TrunkJob& t = jobs.j[k];
  conv12i_body<true>(t.in0, t.in1, t.n_in0, t.W1d, t.s1q, t.b1q, t.B2, t.b2, t.a2, t.a1_out, t.a1_lo, t.n_a1, t.N,
                     (int)blockIdx.x - t.block0, t.nblocks);
}
__global__ __launch_bounds__(kThreads) void conv3_bf16s_jobs(TrunkJobs jobs) {
  int k = 0;
  while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.j[k + 1].block0) ++k;
  const TrunkJob& t = jobs.j[k];
  conv_bf16s_body<Conv3F>(t.a2, t.B3, t.b3, t.a3, t.N, (int)blockIdx.x - t.block0, t.nblocks);
}

// fc on split records: out[N][512] = relu(A x W + b), A = a3 records [N][49][hi 64 | lo 64], k = pos*64 + c.
// Block = BM rows x 128 columns (8 waves, one 16-column tile each).  A arrives per position (256 B per row) through
// registers into a THREE-deep LDS ring (row stride 288 B: conflict-free ds_read_b128), loaded from HBM two
// positions before it is stored; the A fragment reads run RT (hi, lo) pairs ahead of their MFMAs in a register
// ring that continues across positions (the next position's tile was published one barrier earlier); weight
// fragments stream from L2 one position ahead.  BM = 112 fills 232 of the 256 CUs at N = 6400 (128: 200).
template <int BM_>
struct FcFastT {
  static constexpr int OC = 512, BM = BM_, RT = BM / 16, NPOS = 49, KS = 98;  // K = 3136 = 98 k-steps of 32
  static constexpr int RS = 288;                                              // LDS row stride in bytes (18 units = 2 mod 16)
  static constexpr int TILE = BM * RS, NBUF = 3;
  static constexpr int LDS_BYTES = NBUF * TILE;
  static constexpr int V16 = BM * 16;  // 16-byte chunks per position
  static constexpr int IT = (V16 + kThreads - 1) / kThreads;
  static constexpr int TOT = 2 * RT, D = RT;  // fragment pairs per position, pairs in flight
};
using FcFast = FcFastT<112>;

// SPLIT (batches of a few hundred rows, where 4 x ceil(N / BM) blocks would leave most CUs idle): blockIdx.z owns the
// positions [z * per, z * per + per) of the contraction and writes its raw partial sums to out[z][N][512]; fc_reduce adds
// them up in z order with the bias and the ReLU.  Without SPLIT the range is the compile-time [0, 49).
// grid of fc_bf16s for `units` (row block, slice) pairs: the XCD-aware 1-D form (see the block map in the kernel)
inline dim3 fc_grid_xcd(int rb, int slices) { return dim3(32 * ceil_div(rb * slices, 8)); }
// Loads stay ordinary (compiler-tracked) loads.  Tried in r3 and dropped: issuing them through inline asm with
// hand-placed `s_waitcnt vmcnt(N)` (hipcc drains the loads in flight across the loop's back-edge every other position,
// vmcnt(4), i.e. a prefetch distance of one position where the source asks for two).  With the copy-behind-the-wait
// blocks that makes sound (a tied "+v" wait is not: the compiler satisfies the tie with a copy BEFORE the wait, and it
// reuses the registers of a prefetch past the end while that load can still land) the kernel was 4 % SLOWER
// (71.3 vs 68.2 us at N = 6400): the exposed wait is not what bounds a position.
template <class F, bool SPLIT = false, bool STAMPS = false>
__global__ __launch_bounds__(kThreads) void fc_bf16s(const uint8_t* __restrict__ A, const uint4* __restrict__ Bfrag,
                                                     const float* __restrict__ bias, float* __restrict__ out, int N,
                                                     int per, unsigned long long* stamps = nullptr) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  // STAMPS (rela_ffnet_debug_fc_stamps only): shader clock at the phase boundaries of positions 8..15, block 0,
  // waves 0 and 7 -> stamps [2][kStampFrames][kStampPoints]
  int stamp_pos = 0;
  auto stamp = [&](int point) {
    if constexpr (STAMPS) {
      if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 448) && stamp_pos >= 8 && stamp_pos < 8 + kStampFrames)
        stamps[((threadIdx.x == 0 ? 0 : 1) * kStampFrames + stamp_pos - 8) * kStampPoints + point] = __builtin_amdgcn_s_memtime();
    }
  };
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  // Block -> (column group, row block, slice).  A 1-D grid (fc_grid_xcd) is XCD-aware: workgroups go to the 8 XCDs
  // round-robin by linear id, so the four column groups that read the SAME rows of A are given ids with the same
  // id % 8 and consecutive id / 8 -- they run on one XCD at the same time and A comes from HBM once instead of four
  // times (the (4, rb) grid put them on four XCDs: 334 MB of HBM reads for 80 MB of records at N = 6400; 126 MB now,
  // the weights included, which no longer fit one XCD's L2).  Ids past the last unit leave at once (block-uniform,
  // before any barrier).
  int cg, rb_, sl;
  if (gridDim.x == 4) {
    cg = blockIdx.x, rb_ = blockIdx.y, sl = blockIdx.z;
  } else {
    const int rb = (N + F::BM - 1) / F::BM;
    const int units = rb * (SPLIT ? (F::NPOS + per - 1) / per : 1);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int unit = (j >> 2) * 8 + xcd;
    if (unit >= units) return;
    cg = j & 3, rb_ = unit % rb, sl = unit / rb;
  }
  const int ct = cg * kWaves + wave;
  const int row0 = rb_ * F::BM;
  const int p0 = SPLIT ? sl * per : 0;
  const int p1 = SPLIT ? min(F::NPOS, p0 + per) : F::NPOS;
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  // Staging registers: set s holds the tile chunks / weight fragments of the positions of parity s (relative to p0).
  // No predicates on the staging loads and stores (clamped indices repeat a neighbour's chunk, row or position).
  u32x4 st0[F::IT], st1[F::IT], bq0[4], bq1[4];
  const int nrow = min(F::BM, N - row0);
  const uint8_t* arow[F::IT];  // this thread's chunk j of position 0
  int soff[F::IT];
#pragma unroll
  for (int j = 0; j < F::IT; ++j) {
    const int i = min(tid + j * kThreads, F::V16 - 1);
    arow[j] = A + (size_t)(row0 + min(i >> 4, nrow - 1)) * (F::NPOS * 256) + (i & 15) * 16;
    soff[j] = (i >> 4) * F::RS + (i & 15) * 16;
  }
  auto load_chunk = [&](int pos, auto set, int j) {
    const uint8_t* src = arow[j] + min(pos, p1 - 1) * 256;
    if constexpr (decltype(set)::value == 0) st0[j] = *reinterpret_cast<const u32x4*>(src); else st1[j] = *reinterpret_cast<const u32x4*>(src);
  };
  auto store_chunk = [&](int buf, auto set, int j) {
    u32x4* dst = reinterpret_cast<u32x4*>(smem + buf * F::TILE + soff[j]);
    if constexpr (decltype(set)::value == 0) *dst = st0[j]; else *dst = st1[j];
  };
  const uint4* bp = Bfrag + (size_t)ct * F::KS * 2 * 64 + lane;
  auto load_w = [&](int pos, auto set) {  // k-steps 0, 1 of the position x (hi, lo)
    const u32x4* src = reinterpret_cast<const u32x4*>(bp + (size_t)(min(pos, p1 - 1) * 4) * 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (decltype(set)::value == 0) bq0[q] = src[q * 64]; else bq1[q] = src[q * 64];
    }
  };
  f32x4 acc[F::RT];
#pragma unroll
  for (int t = 0; t < F::RT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Prologue: tiles p0 and p0 + 1 into the ring; then the loads a steady-state position finds in flight, in its
  // order: W(p) | A(p + 2) | W(p + 1) | A(p + 3)
#pragma unroll
  for (int j = 0; j < F::IT; ++j) load_chunk(p0, Set0{}, j);
#pragma unroll
  for (int j = 0; j < F::IT; ++j) load_chunk(p0 + 1, Set1{}, j);
#pragma unroll
  for (int j = 0; j < F::IT; ++j) {
    store_chunk(0, Set0{}, j);
    store_chunk(1, Set1{}, j);
  }
  load_w(p0, Set0{});
#pragma unroll
  for (int j = 0; j < F::IT; ++j) load_chunk(p0 + 2, Set0{}, j);
  load_w(p0 + 1, Set1{});
#pragma unroll
  for (int j = 0; j < F::IT; ++j) load_chunk(p0 + 3, Set1{}, j);
  __syncthreads();

  // A fragment ring: pair i of a position = (sub = i / RT, row tile t = i % RT)
  uint4 ah[F::D], al[F::D];
  const int aoff = li * F::RS + g * 16;
  auto a_issue = [&](const uint8_t* tile, int i, int slot) {
    const int sub = i / F::RT, t = i - sub * F::RT;
    const uint8_t* ap = tile + aoff + t * 16 * F::RS + sub * 64;
    ah[slot] = *reinterpret_cast<const uint4*>(ap);
    al[slot] = *reinterpret_cast<const uint4*>(ap + 128);
  };
#pragma unroll
  for (int i = 0; i < F::D; ++i) a_issue(smem, i, i);

  // One position.  In flight when it starts (issue order): W(pos) | A(pos + 2) x IT | W(pos + 1) | A(pos + 3) x IT.
  // The chunks of the tile of position + 2 go to LDS and the loads of position + 4 are issued INSIDE the MFMA loop,
  // one 16-byte chunk per fragment pair: as a block after the loop, on all eight waves at once, they left the matrix
  // pipe idle for ~640 of a position's ~2980 cycles (tools/fc_phases.py).  The tile stored during position p lives in
  // the buffer that was last read during p - 1, which every wave left at the previous barrier.
  static_assert(2 * F::IT + 1 <= F::TOT, "a store and a load slot per chunk");
  int buf = 0;  // (pos - p0) % 3
  auto body = [&](int pos, auto set) {
    constexpr int SB = decltype(set)::value;
    u32x4 bcur[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (SB == 0) bcur[q] = bq0[q]; else bcur[q] = bq1[q];
    }
    load_w(pos + 2, set);
    const int nbuf = (buf == F::NBUF - 1) ? 0 : buf + 1;
    const int sbuf = (nbuf == F::NBUF - 1) ? 0 : nbuf + 1;
    const uint8_t* tile = smem + buf * F::TILE;
    const uint8_t* ntile = smem + nbuf * F::TILE;  // (after the last position: stale rows, read and dropped)
    __builtin_amdgcn_sched_barrier(0);
    stamp(0);
#pragma unroll
    for (int i = 0; i < F::TOT; ++i) {
      const int sub = i / F::RT, t = i - sub * F::RT;
      const int slot = i % F::D;
      const bf16x8 bh = __builtin_bit_cast(bf16x8, bcur[sub * 2]);
      const bf16x8 bl = __builtin_bit_cast(bf16x8, bcur[sub * 2 + 1]);
      const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[slot]);
      const bf16x8 xl = __builtin_bit_cast(bf16x8, al[slot]);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, bh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bl, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bh, acc[t], 0, 0, 0);
      if (i + F::D < F::TOT)
        a_issue(tile, i + F::D, slot);
      else
        a_issue(ntile, i + F::D - F::TOT, slot);
      // chunk j of the tile of position pos + 2 (past the end: rewrites a tile nobody reads again) goes to LDS after
      // pair 2 j + 1 and its registers take chunk j of position pos + 4 after pair 2 j + 2
      if ((i & 1) == 1 && i / 2 < F::IT) store_chunk(sbuf, set, i / 2);
      if ((i & 1) == 0 && i >= 2 && i / 2 - 1 < F::IT) load_chunk(pos + 4, set, i / 2 - 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp(1);
    __syncthreads();
    stamp(4);
    stamp_pos += 1;
    buf = nbuf;
  };
  for (int pos = p0; pos < p1; pos += 2) {
    body(pos, Set0{});
    if (pos + 1 < p1) body(pos + 1, Set1{});
  }
  const int col = ct * 16 + li;
  if constexpr (SPLIT) {
    float* part = out + (size_t)sl * N * F::OC;
#pragma unroll
    for (int t = 0; t < F::RT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + t * 16 + g * 4 + r;
        if (row < N) part[(size_t)row * F::OC + col] = acc[t][r];
      }
    return;
  }
  const float bv = bias[col];
#pragma unroll
  for (int t = 0; t < F::RT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + t * 16 + g * 4 + r;
      if (row < N) {
        const float o = acc[t][r] + bv;
        out[(size_t)row * F::OC + col] = o > 0.f ? o : 0.f;
      }
    }
}

// weights -> [ct][ks][hi, lo][lane] x 8 bf16 in MFMA 16x16x32 fragment order.  mode: kPackConv2 (k = tap*32 + c),
// kPackConv3 (k = tap*64 + c), kPackFc (k = pos*64 + c  <-  torch flatten c*49 + pos)
__device__ __forceinline__ void pack_frags_bf16s_at(int64_t idx, int mode, const float* __restrict__ w, uint16_t* __restrict__ frag, int CT, int KS) {
  if (idx >= (int64_t)CT * KS * 64 * 8) return;
  const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
  const int ks = (int)((idx >> 9) % KS), ct = (int)((idx >> 9) / KS);
  const int k = ks * 32 + (lane >> 4) * 8 + j;
  const int oc = ct * 16 + (lane & 15);
  float v;
  if (mode == 1) {  // conv2: (64,32,4,4)
    const int c = k & 31, tap = k >> 5;
    v = w[((oc * 32 + c) * 4 + (tap >> 2)) * 4 + (tap & 3)];
  } else if (mode == 2) {  // conv3: (64,64,3,3)
    const int c = k & 63, tap = k >> 6;
    v = w[((oc * 64 + c) * 3 + tap / 3) * 3 + tap % 3];
  } else {  // fc: (512,3136)
    const int c = k & 63, pos = k >> 6;
    v = w[(size_t)oc * 3136 + c * 49 + pos];
  }
  const uint16_t hi = f32_to_bf16_rne(v);
  const uint16_t lo = f32_to_bf16_rne(v - bf16_to_f32(hi));
  const size_t base = (((size_t)ct * KS + ks) * 2) * 64 * 8;
  frag[base + (size_t)lane * 8 + j] = hi;
  frag[base + 64 * 8 + (size_t)lane * 8 + j] = lo;
}
__global__ void pack_frags_bf16s(int mode, const float* __restrict__ w, uint16_t* __restrict__ frag, int CT, int KS) {
  pack_frags_bf16s_at((int64_t)blockIdx.x * blockDim.x + threadIdx.x, mode, w, frag, CT, KS);
}

// Dense layer  out[N][OC] = act(A[N][K] * W + bias)  on the same MFMA tiling.
// Block = BM rows x (CTB*16) columns; A is staged through LDS in K-chunks of 32 (double
// buffered, row stride 34 floats -> conflict-free fragment reads), B fragments stream from L2.
enum GemmEpilogue { kEpiBias = 0, kEpiBiasRelu = 1, kEpiLstmCell = 2 };

// K1_ = leading part of K that comes from the first A matrix ([N][K1]); the rest comes from a
// second matrix ([N][K-K1]) -- the LSTM gate GEMM reads [conv features | previous h].
// KSS / KSO: k-steps per column tile of the fragment array and the first one this GEMM uses (a GEMM over a k-range of a
// larger packed operand); INIT: the accumulators start from a [N][OC] tensor passed in A2's place
template <int K_, int OC_, int BM_, int EPI_, int K1_ = K_, int KSS_ = K_ / 4, int KSO_ = 0, bool INIT_ = false>
struct GemmCfg {
  static constexpr int K = K_, OC = OC_, BM = BM_, EPI = EPI_, K1 = K1_, KSS = KSS_, KSO = KSO_;
  static constexpr bool INIT = INIT_;
  static constexpr bool RELU = EPI_ == kEpiBiasRelu;
  static constexpr int CT = OC / 16;
  static constexpr int CTB = CT < kWaves ? CT : kWaves;  // column tiles per block
  static constexpr int RGB = kWaves / CTB;               // row groups per block
  static constexpr int RT = BM / 16;
  static constexpr int RPW = RT / RGB;
  static constexpr int KC = 32;
  static constexpr int NCH = K / KC;
  static constexpr int KS = K / 4;
  static constexpr int LDA = 34;
  static constexpr int V4 = BM * KC / 4;                       // float4 per chunk
  static constexpr int VPT = (V4 + kThreads - 1) / kThreads;  // per thread (the last one may be partial)
};

// Block height is picked per launch (launch_gemm_rows below): 400 half-height fc blocks at N = 6400 sit
// two-deep on 144 CUs and one-deep on 112; 232 blocks of 112 rows fill one round.
using GemmFc = GemmCfg<3136, 512, 64, kEpiBiasRelu>;
using GemmFc112 = GemmCfg<3136, 512, 112, kEpiBiasRelu>;
using GemmHeads = GemmCfg<512, 32, 128, kEpiBias>;
// LSTM gates: [x(3136) | h(512)] x [3648][2048]; columns permuted to 4*unit + gate (i,f,g,o) so the
// four gates of a hidden unit sit in four adjacent lanes of one accumulator tile.
using GemmLstm = GemmCfg<3648, 2048, 128, kEpiLstmCell, 3136>;
using GemmLstm112 = GemmCfg<3648, 2048, 112, kEpiLstmCell, 3136>;
// bf16x2 mode of the actors' LSTM step: the x part of the gates comes from the split-bf16 GEMM (gemm_bf16s.h) and
// this GEMM adds h x W_hh (k-steps 784 .. 911 of the same fragment array) and runs the cell
using GemmLstmH = GemmCfg<512, 2048, 128, kEpiLstmCell, 512, 912, 784, true>;
using GemmLstmH112 = GemmCfg<512, 2048, 112, kEpiLstmCell, 512, 912, 784, true>;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <class G>
__global__ __launch_bounds__(kThreads) void gemm_mfma(const float* __restrict__ A, const float* __restrict__ A2,
                                                      const float* __restrict__ Bfrag,
                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                      const float* __restrict__ c_in, float* __restrict__ c_out,
                                                      int N) {
  __shared__ __attribute__((aligned(16))) float sA[2][G::BM * G::LDA];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, kk = lane >> 4;
  const int ctw = wave % G::CTB, rg = wave / G::CTB;
  const int ct = blockIdx.x * G::CTB + ctw;
  const int row0 = blockIdx.y * G::BM;

  float4 stage[G::VPT];
  auto load_chunk = [&](int ch) {
#pragma unroll
    for (int v = 0; v < G::VPT; ++v) {
      const int idx = tid + v * kThreads;
      if (G::V4 % kThreads != 0 && idx >= G::V4) break;
      const int r = idx >> 3, q = idx & 7;  // KC/4 == 8 float4 per row
      const int row = row0 + r;
      const int k = ch * G::KC + q * 4;
      const float* src = (G::K1 == G::K || k < G::K1) ? A + (size_t)row * G::K1 + k
                                                        : A2 + (size_t)row * (G::K - G::K1) + (k - G::K1);
      stage[v] = (row < N) ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int v = 0; v < G::VPT; ++v) {
      const int idx = tid + v * kThreads;
      if (G::V4 % kThreads != 0 && idx >= G::V4) break;
      const int r = idx >> 3, q = idx & 7;
      float* d = &sA[buf][r * G::LDA + q * 4];
      *reinterpret_cast<float2*>(d) = make_float2(stage[v].x, stage[v].y);
      *reinterpret_cast<float2*>(d + 2) = make_float2(stage[v].z, stage[v].w);
    }
  };

  f32x4 acc[G::RPW];
#pragma unroll
  for (int t = 0; t < G::RPW; ++t) {
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (G::INIT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + (rg * G::RPW + t) * 16 + kk * 4 + r;
        if (row < N) acc[t][r] = A2[(size_t)row * G::OC + blockIdx.x * G::CTB * 16 + ctw * 16 + li];
      }
    }
  }

  const float* bptr = Bfrag + ((size_t)ct * G::KSS + G::KSO) * 64 + lane;

  load_chunk(0);
  store_chunk(0);
  // B fragments stream from L2 one K-chunk AHEAD of their use (register double buffer), so the
  // MFMAs of a chunk never wait on the fragment loads issued in the same chunk
  float bnext[G::KC / 4];
#pragma unroll
  for (int j = 0; j < G::KC / 4; ++j) bnext[j] = bptr[(size_t)j * 64];
  __syncthreads();
  for (int ch = 0; ch < G::NCH; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < G::NCH) load_chunk(ch + 1);
    float bfr[G::KC / 4];
#pragma unroll
    for (int j = 0; j < G::KC / 4; ++j) bfr[j] = bnext[j];
    if (ch + 1 < G::NCH) {
#pragma unroll
      for (int j = 0; j < G::KC / 4; ++j) bnext[j] = bptr[(size_t)((ch + 1) * (G::KC / 4) + j) * 64];
    }
#pragma unroll
    for (int j = 0; j < G::KC / 4; ++j) {
#pragma unroll
      for (int t = 0; t < G::RPW; ++t) {
        const int r = (rg * G::RPW + t) * 16 + li;
        const float a = sA[buf][r * G::LDA + j * 4 + kk];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfr[j], acc[t], 0, 0, 0);
      }
    }
    if (ch + 1 < G::NCH) store_chunk(buf ^ 1);
    __syncthreads();
  }

  const int col = ct * 16 + li;
  const float bv = bias[col];
#pragma unroll
  for (int t = 0; t < G::RPW; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + (rg * G::RPW + t) * 16 + kk * 4 + r;
      if constexpr (G::EPI == kEpiLstmCell) {
        // torch LSTM cell (gate order i,f,g,o): c' = sig(f)*c + sig(i)*tanh(g); h' = sig(o)*tanh(c')
        const float pre = acc[t][r] + bv;
        const int base = lane & ~3;
        const float gi = __shfl(pre, base + 0, 64), gf = __shfl(pre, base + 1, 64);
        const float gg = __shfl(pre, base + 2, 64), go = __shfl(pre, base + 3, 64);
        if ((li & 3) == 0 && row < N) {
          const int unit = col >> 2;
          const float cp = c_in[(size_t)row * (G::OC / 4) + unit];
          const float c = sigmoidf_(gf) * cp + sigmoidf_(gi) * tanhf(gg);
          c_out[(size_t)row * (G::OC / 4) + unit] = c;
          out[(size_t)row * (G::OC / 4) + unit] = sigmoidf_(go) * tanhf(c);
        }
      } else if (row < N) {
        float v = acc[t][r] + bv;
        if (G::RELU) v = v > 0.f ? v : 0.f;
        out[(size_t)row * G::OC + col] = v;
      }
    }
  }
}

// duel(): q = v + a*legal - mean_A(a*legal)   net.py:33-39 (mean over A, not over #legal)
// adv (optional) receives the raw fc_a outputs, which AtariLSTMNet.act ranks (net.py:119-123)
__global__ void dueling_kernel(const float* __restrict__ ha, const float* __restrict__ legal, float* __restrict__ q,
                               float* __restrict__ adv, int N, int A) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* row = ha + (size_t)n * 32;
  const float v = row[31];
  float la[31];
  float sum = 0.f;
  for (int j = 0; j < A; ++j) {
    if (adv) adv[(size_t)n * A + j] = row[j];
    la[j] = row[j] * legal[(size_t)n * A + j];
    sum += la[j];
  }
  const float mean = sum / (float)A;
  if (q)
    for (int j = 0; j < A; ++j) q[(size_t)n * A + j] = (v + la[j]) - mean;
}

// Heads + dueling in ONE launch (AtariFFNet): [N, 512] x [512, 32] (fc_a in columns 0..A-1, fc_v in column 31) and
// q = v + a*legal - mean_A(a*legal) (net.py:33-39).  The two separate launches (a 4-block-wide GEMM and a
// one-thread-per-row kernel) cost 18 + 10 us at N = 6,400 and 17 + 9 us at N = 512 for 0.2 GFLOP: launch and latency,
// not work.  Block = 16 rows, 4 waves; wave w multiplies the k-slice [128w, 128w + 128) on v_mfma_f32_16x16x4_f32 (a
// lane reads 32 CONTIGUOUS floats of its row: k-step j of lane group g is k = 128w + 32g + j, the weights are packed
// in that order by pack_heads_perm), the four partial tiles meet in LDS, then 16 threads per row finish the row.
constexpr int kHeadRows = 16;
__device__ __forceinline__ void pack_heads_perm_at(int idx, const float* __restrict__ a_w, const float* __restrict__ v_w, int A,
                                float* __restrict__ out) {
  if (idx >= 2 * 4 * 32 * 64) return;
  const int lane = idx & 63, j = (idx >> 6) & 31, w = (idx >> 11) & 3, ct = idx >> 13;
  const int k = 128 * w + 32 * (lane >> 4) + j, col = ct * 16 + (lane & 15);
  out[idx] = col < A ? a_w[(size_t)col * 512 + k] : (col == 31 ? v_w[k] : 0.f);
}
__global__ void pack_heads_perm(const float* __restrict__ a_w, const float* __restrict__ v_w, int A,
                                float* __restrict__ out) {
  pack_heads_perm_at((int)blockIdx.x * blockDim.x + threadIdx.x, a_w, v_w, A, out);
}

__device__ __forceinline__ void heads_duel_body(const float* __restrict__ h, const float* __restrict__ Bhp,
                                                const float* __restrict__ bias, const float* __restrict__ legal,
                                                float* __restrict__ ha, float* __restrict__ q, int N, int A, int bid) {
  __shared__ float part[4][kHeadRows][33];
  __shared__ float hs[kHeadRows][33];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int row0 = bid * kHeadRows;
  float av[32];
  {
    const int row = min(row0 + li, N - 1);
    const float4* hp = reinterpret_cast<const float4*>(h + (size_t)row * 512 + 128 * wave + 32 * g);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float4 v = hp[c];
      av[4 * c] = v.x, av[4 * c + 1] = v.y, av[4 * c + 2] = v.z, av[4 * c + 3] = v.w;
    }
  }
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const float* bp = Bhp + ((size_t)(ct * 4 + wave) * 32) * 64 + lane;
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bp[j * 64], acc[ct], 0, 0, 0);
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][4 * g + r][ct * 16 + li] = acc[ct][r];
  __syncthreads();
  const int r = tid >> 4, c = tid & 15, row = row0 + r;
#pragma unroll
  for (int hcol = 0; hcol < 2; ++hcol) {
    const int col = c + 16 * hcol;
    hs[r][col] = ((part[0][r][col] + part[1][r][col]) + (part[2][r][col] + part[3][r][col])) + bias[col];
  }
  __syncthreads();
  if (row >= N) return;
  if (ha) ha[(size_t)row * 32 + c] = hs[r][c], ha[(size_t)row * 32 + c + 16] = hs[r][c + 16];
  const float* lg = legal + (size_t)row * A;
  float sum = 0.f;  // in column order, as the one-thread-per-row kernel summed it
  for (int j = 0; j < A; ++j) sum += hs[r][j] * lg[j];
  const float mean = sum / (float)A, v = hs[r][31];
  if (c < A) q[(size_t)row * A + c] = (v + hs[r][c] * lg[c]) - mean;
  if (c + 16 < A) q[(size_t)row * A + c + 16] = (v + hs[r][c + 16] * lg[c + 16]) - mean;
}
__global__ __launch_bounds__(256) void heads_duel(const float* __restrict__ h, const float* __restrict__ Bhp,
                                                  const float* __restrict__ bias, const float* __restrict__ legal,
                                                  float* __restrict__ ha, float* __restrict__ q, int N, int A) {
  heads_duel_body(h, Bhp, bias, legal, ha, q, N, A, blockIdx.x);
}
// up to three head passes in one launch (the learner's three Q tables): segment j owns blocks [first[j], first[j + 1])
struct HeadJobs {
  const float *h[3], *Bhp[3], *bias[3], *legal[3];
  float *ha[3], *q[3];
  int N[3], first[4];
};
__global__ __launch_bounds__(256) void heads_duel_jobs(HeadJobs jb, int A) {
  int j = 0;
  while (j < 2 && (int)blockIdx.x >= jb.first[j + 1]) ++j;
  heads_duel_body(jb.h[j], jb.Bhp[j], jb.bias[j], jb.legal[j], jb.ha[j], jb.q[j], jb.N[j], A, (int)blockIdx.x - jb.first[j]);
}

// a3 split records -> f32 in place (per pixel: 64 x bf16 hi | 64 x bf16 lo  ->  64 x f32; hi + lo is exact in f32).
// One wave per pixel: all of its loads complete before its stores (same wave, program order).  Used where the trunk
// runs on split-bf16 MFMA but the batch is too small for fc_bf16s and fc goes through the f32 split-K GEMM.
__global__ void unsplit_records64(uint8_t* __restrict__ rec, int64_t pixels) {
  const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= pixels) return;
  const int c = threadIdx.x & 63;
  uint8_t* r = rec + p * 256;
  const uint16_t hi = reinterpret_cast<const uint16_t*>(r)[c], lo = reinterpret_cast<const uint16_t*>(r)[64 + c];
  const float v = bf16_to_f32(hi) + bf16_to_f32(lo);
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  reinterpret_cast<float*>(r)[c] = v;
}

// ---- weight packing (load_state_dict time) ------------------------------------------------
// ---- fc forward for small batches --------------------------------------------------------------
// Below kFcSplitBelow rows gemm_mfma<GemmFc> launches fewer than 128 blocks, each walking all 98
// K-chunks (123 us at N = 512, the same at N = 80).  There the product runs as a split-K instance of
// gemm_lds (gemm_lds.h) over ~256 blocks; fc_reduce sums the partial tiles in a fixed order and
// applies bias + ReLU.  The partial tiles live in the caller's workspace behind `ha`.
constexpr int kFcSplitBelow = 2048;
constexpr int64_t kFcPartFloats = (int64_t)8192 * 512;  // splits * N <= 8192 rows of partial sums (f32 split-K: <= 4096)
inline int fc_splits(int N) {
  const int rb = ceil_div(N, 128);
  const int sp = 32 / rb;
  return sp < 1 ? 1 : sp;
}

using TileFcSmall = gemm::TileCfg<128, 64, 4, 2, false>;
struct ProbFcFwd : gemm::ProbBase {
  const float *a3, *wt;  // a3 [N][3136] (k = pos*64+c), wt [3136][512]
  float* part;           // [splits][N][512]
  __device__ float4 loadA(int m, int k) const {
    return m < M ? *reinterpret_cast<const float4*>(a3 + (size_t)m * 3136 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ float4 loadB(int k, int n) const { return *reinterpret_cast<const float4*>(wt + (size_t)k * 512 + n); }
  __device__ void store(int z, int m, int n, float v) const { part[((size_t)z * M + m) * 512 + n] = v; }
};

__global__ void fc_reduce(const float* __restrict__ part, int splits, int N, const float* __restrict__ bias,
                          float* __restrict__ h) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of h
  if (idx >= N * 128) return;
  const int n4 = (idx & 127) * 4;
  const size_t off = (size_t)(idx >> 7) * 512 + n4;
  float4 s = *reinterpret_cast<const float4*>(bias + n4);
  for (int z = 0; z < splits; ++z) {
    const float4 v = *reinterpret_cast<const float4*>(part + (size_t)z * N * 512 + off);
    s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
  }
  *reinterpret_cast<float4*>(h + off) =
      make_float4(s.x > 0.f ? s.x : 0.f, s.y > 0.f ? s.y : 0.f, s.z > 0.f ? s.z : 0.f, s.w > 0.f ? s.w : 0.f);
}

// wt[k = pos*64 + c][u] = linear.0.weight[u][c*49 + pos]   (net.py:49 flattens channel-first)
__device__ __forceinline__ void pack_fc_t_at(int idx, const float* __restrict__ w, float* __restrict__ wt) {
  if (idx >= 3136 * 512) return;
  const int k = idx >> 9, u = idx & 511;
  const int c = k & 63, pos = k >> 6;
  wt[idx] = w[(size_t)u * 3136 + c * 49 + pos];
}
__global__ void pack_fc_t(const float* __restrict__ w, float* __restrict__ wt) {
  pack_fc_t_at((int)blockIdx.x * blockDim.x + threadIdx.x, w, wt);
}

enum PackMode { kPackConv1 = 0, kPackConv2 = 1, kPackConv3 = 2, kPackFc = 3, kPackHeads = 4, kPackLstm = 5 };

__device__ __forceinline__ void pack_frags_at(int64_t idx, int mode, const float* __restrict__ w, const float* __restrict__ w2, int num_action,
                           float* __restrict__ frag, int CT, int KS) {
  const int64_t total = (int64_t)CT * KS * 64;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const int ks = (int)((idx >> 6) % KS);
  const int ct = (int)((idx >> 6) / KS);
  const int k = ks * 4 + (lane >> 4);
  const int oc = ct * 16 + (lane & 15);
  float v = 0.f;
  switch (mode) {
    case kPackConv1:  // k = (c, kh, kw) = state_dict order; fold s/255 (net.py:46)
      v = w[oc * 256 + k] / 255.0f;
      break;
    case kPackConv2: {  // k = (kh, kw, c)
      const int c = k & 31, kw = (k >> 5) & 3, kh = k >> 7;
      v = w[((oc * 32 + c) * 4 + kh) * 4 + kw];
      break;
    }
    case kPackConv3: {  // k = (kh, kw, c)
      const int c = k & 63, r = k >> 6, kw = r % 3, kh = r / 3;
      v = w[((oc * 64 + c) * 3 + kh) * 3 + kw];
      break;
    }
    case kPackFc: {  // k = pos*64 + c  <-  torch flatten c*49 + pos (net.py:49)
      const int c = k & 63, p = k >> 6;
      v = w[(size_t)oc * 3136 + c * 49 + p];
      break;
    }
    case kPackLstm: {  // col' = 4*unit + gate  <-  row gate*512 + unit of weight_ih_l0 / weight_hh_l0
      const int row = (oc & 3) * 512 + (oc >> 2);
      if (k < 3136) {
        const int c = k & 63, p = k >> 6;  // k = pos*64 + c  <-  c*49 + pos (net.py:105-106)
        v = w[(size_t)row * 3136 + c * 49 + p];
      } else {
        v = w2[(size_t)row * 512 + (k - 3136)];
      }
      break;
    }
    case kPackHeads:
      if (oc < num_action)
        v = w[oc * 512 + k];  // fc_a
      else if (oc == 31)
        v = w2[k];  // fc_v
      break;
  }
  frag[idx] = v;
}
__global__ void pack_frags(int mode, const float* __restrict__ w, const float* __restrict__ w2, int num_action,
                           float* __restrict__ frag, int CT, int KS) {
  pack_frags_at((int64_t)blockIdx.x * blockDim.x + threadIdx.x, mode, w, w2, num_action, frag, CT, KS);
}

// weight_ih_l0 [gate*512 + unit][c*49 + pos] -> rec64 rows in the gate GEMM's permuted column order:
// rec[col = 4*unit + gate][chunk = pos][64 hi | 64 lo] over c (the B operand of gemm16::gemm_rec64_nt for the x part)
__global__ void pack_wih_rec64_perm(const float* __restrict__ wih, uint8_t* __restrict__ rec) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 8 channels
  if (i >= (int64_t)2048 * 392) return;
  const int col = (int)(i / 392), o = (int)(i - (int64_t)col * 392);
  const int pos = o >> 3, c0 = (o & 7) * 8;
  const float* w = wih + (size_t)((col & 3) * 512 + (col >> 2)) * 3136 + pos;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = w[(c0 + j) * 49];
  uint4 hi, lo;
  gemm16::split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
  uint8_t* r = rec + ((size_t)col * 49 + pos) * 256 + (o & 7) * 16;
  *reinterpret_cast<uint4*>(r) = hi;
  *reinterpret_cast<uint4*>(r + 128) = lo;
}
__global__ void pack_lstm_bias(const float* __restrict__ bih, const float* __restrict__ bhh, float* __restrict__ out) {
  const int oc = blockIdx.x * blockDim.x + threadIdx.x;
  if (oc < 2048) {
    const int row = (oc & 3) * 512 + (oc >> 2);
    out[oc] = bih[row] + bhh[row];
  }
}

__global__ void pack_head_bias(const float* __restrict__ ab, const float* __restrict__ vb, int A, float* __restrict__ out) {
  const int j = threadIdx.x;
  if (j < 32) out[j] = j < A ? ab[j] : (j == 31 ? vb[0] : 0.f);
}

// conv1 for conv12_i8: one block of 256 threads per output channel c (thread = weight k = p*64 + ky*8 + kx of
// state_dict's [32][4][8][8]).  s_c = max|w| / QMAX rounded UP to f32 (so that |w / s_c| <= QMAX), q = rint(w / s_c) in
// double, balanced base-256 digits; b' = b + 128 s_c sum q in double, rounded once.  Digit d of (c, k) is byte
// j = (ky % 4) * 4 + kx % 4 of lane (g = p) * 16 + c % 16 of fragment [d][c / 16][tap (ky / 4) * 2 + kx / 4].
constexpr int kConv1QMax = 127 * 65536 + 127 * 256 + 127;
__device__ __forceinline__ void pack_conv1_i8_block(int c, const float* __restrict__ w, const float* __restrict__ b,
                                                    uint8_t* __restrict__ W1d, float* __restrict__ s1q, float* __restrict__ b1q) {
  __shared__ float amax[256];
  __shared__ long long qsum[256];
  const int k = threadIdx.x;
  const float wv = w[c * 256 + k] / 255.0f;  // the s / 255 of net.py:46, folded as in the other conv1 packs
  amax[k] = fabsf(wv);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (k < o) amax[k] = fmaxf(amax[k], amax[k + o]);
    __syncthreads();
  }
  const float mx = amax[0];
  float sc = 1.0f;
  if (mx > 0.f) {
    sc = (float)((double)mx / (double)kConv1QMax);
    if (sc <= 0.f || (double)mx / (double)sc > (double)kConv1QMax) sc = nextafterf(sc, INFINITY);
  }
  long long q = llrint((double)wv / (double)sc);
  q = q > kConv1QMax ? kConv1QMax : (q < -kConv1QMax ? -kConv1QMax : q);
  qsum[k] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (k < o) qsum[k] += qsum[k + o];
    __syncthreads();
  }
  if (k == 0) {
    s1q[c] = sc;
    b1q[c] = (float)((double)b[c] + 128.0 * (double)sc * (double)qsum[0]);
  }
  // balanced digits: lo, mid in [-128, 127], hi = the rest (|hi| <= 127 by the bound on q)
  const int qi = (int)q;
  const int lo = ((qi + 128) & 255) - 128;
  const int q1 = (qi - lo) >> 8;
  const int mid = ((q1 + 128) & 255) - 128;
  const int hi = (q1 - mid) >> 8;
  const int pl = k >> 6, ky = (k >> 3) & 7, kx = k & 7;
  const int tap = (ky >> 2) * 2 + (kx >> 2), j = (ky & 3) * 4 + (kx & 3);
  const int lane = pl * 16 + (c & 15), ct = c >> 4;
  const int dig[3] = {hi, mid, lo};
#pragma unroll
  for (int d = 0; d < 3; ++d) W1d[((size_t)((d * 2 + ct) * 4 + tap) * 64 + lane) * 16 + j] = (uint8_t)(int8_t)dig[d];
}

__global__ __launch_bounds__(256) void pack_conv1_i8(const float* __restrict__ w, const float* __restrict__ b, uint8_t* __restrict__ W1d,
                                                     float* __restrict__ s1q, float* __restrict__ b1q) {
  pack_conv1_i8_block((int)blockIdx.x, w, b, W1d, s1q, b1q);
}

__global__ void pack_f32emu(int mode, const float* __restrict__ w, uint16_t* __restrict__ frag, int NCG, int KS) {
  f32emu::pack_f32emu_at((int64_t)blockIdx.x * blockDim.x + threadIdx.x, mode, w, frag, NCG, KS);
}

// Every kernel-layout copy of an AtariFFNet's weights in ONE launch (a learner re-packs after every optimiser step:
// twelve pack kernels, four device copies and the head bias were seventeen launches).  A block finds its job in a
// table of first-block indices; the job bodies are the *_at functions of the separate kernels.
struct PackAllArgs {
  const float* p[12];  // rela_ffnet_params order
  uint16_t *B1, *B2f, *B3f, *Bff;
  float *B2, *B3, *Bf, *BfT, *Bh, *Bhp, *b1, *b2, *b3, *bf, *bh;
  int A;
  float *w2p, *w3p, *wfcp;  // the learner's dgrad operand copies (ffnet_layout.h), or NULL
  uint8_t* W1d;             // conv1 for the int8 matrix cores (pack_conv1_i8_block)
  float *s1q, *b1q;
  uint16_t *B2e, *B3e, *Bfe;  // bf16 triples of the f32-accurate mode (f32emu::pack_f32emu_at)
  int first[21];  // first block of job j; first[20] = total
};
__global__ void pack_ffnet_all(PackAllArgs a) {
  const int b = blockIdx.x;
  int j = 0;
  while (b >= a.first[j + 1]) ++j;
  const int64_t idx = (int64_t)(b - a.first[j]) * 256 + threadIdx.x;
  switch (j) {
    case 0: pack_conv1_bf16x3_at((int)idx, a.p[0], a.B1, 0); break;
    case 1: break;  // (was: plane-major conv1 fragments of the half-frame bf16 kernels, removed in r4; zero blocks)
    case 2: pack_frags_at(idx, kPackConv2, a.p[2], nullptr, a.A, a.B2, 4, 128); break;
    case 3: pack_frags_at(idx, kPackConv3, a.p[4], nullptr, a.A, a.B3, 4, 144); break;
    case 4: pack_frags_at(idx, kPackFc, a.p[6], nullptr, a.A, a.Bf, 32, 784); break;
    case 5: pack_fc_t_at((int)idx, a.p[6], a.BfT); break;
    case 6: pack_frags_at(idx, kPackHeads, a.p[10], a.p[8], a.A, a.Bh, 2, 128); break;
    case 7: pack_heads_perm_at((int)idx, a.p[10], a.p[8], a.A, a.Bhp); break;
    case 8: pack_frags_bf16s_at(idx, 1, a.p[2], a.B2f, Conv2F::CT, Conv2F::KS); break;
    case 9: pack_frags_bf16s_at(idx, 2, a.p[4], a.B3f, Conv3F::CT, Conv3F::KS); break;
    case 10: pack_frags_bf16s_at(idx, 3, a.p[6], a.Bff, 32, FcFast::KS); break;
    case 11: {  // biases: conv1 32 | conv2 64 | conv3 64 | fc 512
      const int i = (int)idx;
      if (i < 32) a.b1[i] = a.p[1][i];
      else if (i < 96) a.b2[i - 32] = a.p[3][i - 32];
      else if (i < 160) a.b3[i - 96] = a.p[5][i - 96];
      else if (i < 672) a.bf[i - 160] = a.p[7][i - 160];
      break;
    }
    case 12: {  // head bias
      const int i = (int)idx;
      if (i < 32) a.bh[i] = i < a.A ? a.p[11][i] : (i == 31 ? a.p[9][0] : 0.f);
      break;
    }
    case 13: if (idx < 64 * 512) permute_weight_at(kPermConv2, (int)idx, a.p[2], a.w2p); break;
    case 14: if (idx < 64 * 576) permute_weight_at(kPermConv3, (int)idx, a.p[4], a.w3p); break;
    case 15: if (idx < 512 * 3136) permute_weight_at(kPermFc, (int)idx, a.p[6], a.wfcp); break;
    case 16: pack_conv1_i8_block(b - a.first[16], a.p[0], a.p[1], a.W1d, a.s1q, a.b1q); break;  // (block-uniform: it synchronises)
    case 17: f32emu::pack_f32emu_at(idx, 1, a.p[2], a.B2e, f32emu::ProbConv2::NCG, f32emu::ProbConv2::KS); break;
    case 18: f32emu::pack_f32emu_at(idx, 2, a.p[4], a.B3e, f32emu::ProbConv3::NCG, f32emu::ProbConv3::KS); break;
    default: f32emu::pack_f32emu_at(idx, 3, a.p[6], a.Bfe, f32emu::ProbFc::NCG, f32emu::ProbFc::KS); break;
  }
}

}  // namespace
}  // namespace rela_amd

using namespace rela_amd;

namespace {
const char* const kProfActor[6] = {"conv1_bf16x3", "conv2_mfma", "conv3_mfma", "fc_mfma", "heads_mfma", "dueling"};
const char* const kProfLearner[6] = {"learner_fwd_conv1", "learner_fwd_conv2", "learner_fwd_conv3",
                                     "learner_fwd_fc",    "learner_fwd_heads", "learner_fwd_dueling"};
}  // namespace

struct rela_ffnet {
  int device = 0;
  int num_action = 0;
  FFNetDev d;
  bool loaded = false;
  uint64_t version = 0;  // bumped by every load
  const char* const* prof_names = nullptr;  // per-kernel timing labels (actor-side by default)
  // 0 = exact f32 MFMA | 1 = split-bf16 MFMA for conv2 / conv3 / fc (two bf16 parts per operand: 16-bit significands, the
  // "fast" mode) | 2 = f32 operands as THREE bf16 parts on the bf16 MFMA (gemm_f32emu.h: f32 accuracy)
  int precision = 0;
  int max_rows = 0;  // > 0: the owner never runs more rows (a learner's batch): layouts only larger batches read are not packed
  // BfT (the f32 split-K fc's weights) of such a net in the split-bf16 mode: no kernel of that mode reads it from 128
  // rows up, so the per-step re-pack skips it (6.4 MB) and the f32 path packs it on demand from the owner's buffer
  mutable bool bft_stale = false;
  const float* bft_src = nullptr;
};


extern "C" int rela_ffnet_create(rela_ffnet** out, int num_action, int device) {
  RELA_CHECK(out && num_action >= 1 && num_action <= 31, RELA_EINVAL,
             "rela_ffnet_create: num_action must be in 1..31 (got %d)", num_action);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_last_error("rela_ffnet_create: HIP device %d not available (%d visible); there is no CPU path", device, ndev);
    return RELA_ENODEV;
  }
  DeviceGuard g(device);
  auto* n = new rela_ffnet();
  n->device = device;
  n->num_action = num_action;
  FFNetDev& d = n->d;
  RELA_HIP(hipMalloc(&d.B1, sizeof(uint4) * Conv1B::FRAG_UINT4));
  RELA_HIP(hipMalloc(&d.b1, sizeof(float) * 32));
  RELA_HIP(hipMalloc(&d.B2, sizeof(float) * 4 * 128 * 64));
  RELA_HIP(hipMalloc(&d.b2, sizeof(float) * 64));
  RELA_HIP(hipMalloc(&d.B3, sizeof(float) * 4 * 144 * 64));
  RELA_HIP(hipMalloc(&d.b3, sizeof(float) * 64));
  RELA_HIP(hipMalloc(&d.Bf, sizeof(float) * 32 * 784 * 64));
  RELA_HIP(hipMalloc(&d.BfT, sizeof(float) * 3136 * 512));
  RELA_HIP(hipMalloc(&d.bf, sizeof(float) * 512));
  RELA_HIP(hipMalloc(&d.Bh, sizeof(float) * 2 * 128 * 64));
  RELA_HIP(hipMalloc(&d.bh, sizeof(float) * 32));
  RELA_HIP(hipMalloc(&d.W1d, sizeof(uint4) * Conv12I::W1_UINT4));
  RELA_HIP(hipMalloc(&d.s1q, sizeof(float) * 32));
  RELA_HIP(hipMalloc(&d.b1q, sizeof(float) * 32));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_i8), hipFuncAttributeMaxDynamicSharedMemorySize,
                               Conv12I::LDS_TOTAL));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_i8_jobs), hipFuncAttributeMaxDynamicSharedMemorySize,
                               Conv12I::LDS_TOTAL));
  RELA_HIP(hipMalloc(&d.Bhp, sizeof(float) * 2 * 4 * 32 * 64));
  RELA_HIP(hipMalloc(&d.B2f, sizeof(uint4) * Conv2F::CT * Conv2F::KS * 2 * 64));
  RELA_HIP(hipMalloc(&d.B3f, sizeof(uint4) * Conv3F::CT * Conv3F::KS * 2 * 64));
  RELA_HIP(hipMalloc(&d.Bff, sizeof(uint4) * 32 * FcFast::KS * 2 * 64));
  RELA_HIP(hipMalloc(&d.B2e, sizeof(uint4) * f32emu::packed_u4<f32emu::ProbConv2>()));
  RELA_HIP(hipMalloc(&d.B3e, sizeof(uint4) * f32emu::packed_u4<f32emu::ProbConv3>()));
  RELA_HIP(hipMalloc(&d.Bfe, sizeof(uint4) * f32emu::packed_u4<f32emu::ProbFc>()));
  // opt in to > 64 KB of dynamic LDS once per process/device
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&s3::conv12_s3<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               s3::Conv12S::LDS_TOTAL));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&s3::conv12_s3<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               s3::Conv12S::LDS_TOTAL));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1_bf16x3),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Conv1B::LDS_BYTES));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16s<Conv3F>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Conv3F::LDS_TOTAL));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_bf16s<FcFast, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, FcFast::LDS_BYTES));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_bf16s<FcFast>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, FcFast::LDS_BYTES));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma<Conv2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Conv2::LDS_BYTES));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma<Conv3>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Conv3::LDS_BYTES));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_bstat<Conv2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Conv2::LDS_BYTES));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_bstat<Conv3>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Conv3::LDS_BYTES));
  *out = n;
  return RELA_OK;
}

extern "C" void rela_ffnet_destroy(rela_ffnet* n) {
  if (!n) return;
  DeviceGuard g(n->device);
  (void)hipDeviceSynchronize();
  void* ps[] = {n->d.B1, n->d.b1, n->d.B2, n->d.b2, n->d.B3, n->d.b3, n->d.Bf, n->d.bf, n->d.Bh, n->d.bh,
                n->d.BfT, n->d.B2f, n->d.B3f, n->d.Bff, n->d.Bhp, n->d.W1d, n->d.s1q, n->d.b1q, n->d.B2e, n->d.B3e, n->d.Bfe};
  for (void* p : ps) (void)hipFree(p);
  delete n;
}

// Diagnostic: the fused conv1 -> conv2 kernel (conv12_i8) with shader-clock stamps at its phase boundaries (block 0,
// waves 0 and 7, first 8 frames): out_host receives [2][8][12] u64.  Points: 0 frame start | 1 conv1 done | 2 barrier |
// 3 conv2 MFMAs done | 4 epilogue | 5 barrier.
extern "C" int rela_ffnet_debug_conv12_stamps(const rela_ffnet* n, int N, const uint8_t* s_dev, unsigned long long* out_host,
                                              void* stream_) {
  RELA_CHECK(n && n->loaded && N >= 1 && s_dev && out_host, RELA_EINVAL, "rela_ffnet_debug_conv12_stamps: bad arguments");
  DeviceGuard g(n->device);
  hipStream_t s = (hipStream_t)stream_;
  uint8_t* a2 = nullptr;
  unsigned long long* st = nullptr;
  const size_t nst = (size_t)2 * kStampFrames * kStampPoints;
  RELA_HIP(hipMalloc(&a2, (size_t)N * 81 * 256));
  RELA_HIP(hipMalloc(&st, nst * 8));
  RELA_HIP(hipMemsetAsync(st, 0, nst * 8, s));
  const FFNetDev& d = n->d;
  static const hipError_t attr8 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_i8_stamps),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, Conv12I::LDS_TOTAL);
  RELA_HIP(attr8);
  hipLaunchKernelGGL(conv12_i8_stamps, dim3(std::min(kNumCU, N)), dim3(kThreads), Conv12I::LDS_TOTAL, s, s_dev,
                     (const uint4*)d.W1d, (const float*)d.s1q, (const float*)d.b1q, (const uint4*)d.B2f, (const float*)d.b2, a2, N, st);
  RELA_HIP(hipStreamSynchronize(s));
  RELA_HIP(hipMemcpy(out_host, st, nst * 8, hipMemcpyDeviceToHost));
  (void)hipFree(a2);
  (void)hipFree(st);
  return RELA_OK;
}

// Diagnostic: conv3 of the split-bf16 mode with shader-clock stamps (block 0, waves 0 and 7, its first 8 groups of two
// frames): out_host [2][8][12] u64, points 0 group start | 1 MFMA loop (+ staging) done | 2 epilogue stored | 3 barrier |
// 4 copy-out issued.  a2_records: [N][81][hi 64 | lo 64] (any bytes do for timing).
extern "C" int rela_ffnet_debug_conv3_stamps(const rela_ffnet* n, int N, const uint8_t* a2_records, unsigned long long* out_host,
                                             void* stream_) {
  RELA_CHECK(n && n->loaded && N >= 2 && a2_records && out_host, RELA_EINVAL, "rela_ffnet_debug_conv3_stamps: bad arguments");
  DeviceGuard g(n->device);
  hipStream_t s = (hipStream_t)stream_;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_bf16s_stamps),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, Conv3F::LDS_TOTAL);
  RELA_HIP(attr);
  uint8_t* a3 = nullptr;
  unsigned long long* st = nullptr;
  const size_t nst = (size_t)2 * kStampFrames * kStampPoints;
  RELA_HIP(hipMalloc(&a3, (size_t)N * 49 * 256));
  RELA_HIP(hipMalloc(&st, nst * 8));
  RELA_HIP(hipMemsetAsync(st, 0, nst * 8, s));
  const FFNetDev& d = n->d;
  hipLaunchKernelGGL(conv3_bf16s_stamps, dim3(std::min(kNumCU, ceil_div(N, Conv3F::S))), dim3(kThreads), Conv3F::LDS_TOTAL, s,
                     a2_records, (const uint4*)d.B3f, (const float*)d.b3, a3, N, st);
  RELA_HIP(hipStreamSynchronize(s));
  RELA_HIP(hipMemcpy(out_host, st, nst * 8, hipMemcpyDeviceToHost));
  (void)hipFree(a3);
  (void)hipFree(st);
  return RELA_OK;
}

// Diagnostic: fc_bf16s with shader-clock stamps (block 0, waves 0 and 7, positions 8..15): out_host [2][8][12] u64, points
// 0 position start (weight fragments in registers) | 1 MFMA loop done | 2 next tile stored | 3 loads issued | 4 barrier.
// a3_records: [N][49][hi 64 | lo 64] split records (any bytes do for timing).
extern "C" int rela_ffnet_debug_fc_stamps(const rela_ffnet* n, int N, const uint8_t* a3_records, unsigned long long* out_host,
                                          void* stream_) {
  RELA_CHECK(n && n->loaded && N >= FcFast::BM && a3_records && out_host, RELA_EINVAL, "rela_ffnet_debug_fc_stamps: bad arguments");
  DeviceGuard g(n->device);
  hipStream_t s = (hipStream_t)stream_;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_bf16s<FcFast, false, true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, FcFast::LDS_BYTES);
  RELA_HIP(attr);
  float* h = nullptr;
  unsigned long long* st = nullptr;
  const size_t nst = (size_t)2 * kStampFrames * kStampPoints;
  RELA_HIP(hipMalloc(&h, (size_t)N * 512 * 4));
  RELA_HIP(hipMalloc(&st, nst * 8));
  RELA_HIP(hipMemsetAsync(st, 0, nst * 8, s));
  const FFNetDev& d = n->d;
  hipLaunchKernelGGL((fc_bf16s<FcFast, false, true>), fc_grid_xcd(ceil_div(N, FcFast::BM), 1), dim3(kThreads),
                     FcFast::LDS_BYTES, s, a3_records, (const uint4*)d.Bff, (const float*)d.bf, h, N, 0, st);
  RELA_HIP(hipStreamSynchronize(s));
  RELA_HIP(hipMemcpy(out_host, st, nst * 8, hipMemcpyDeviceToHost));
  (void)hipFree(h);
  (void)hipFree(st);
  return RELA_OK;
}

// Diagnostic / test hook: conv1 -> conv2 of the split-bf16 mode for n frames through the job form of the kernel, which
// also copies conv1's records out: a1_records [n][400][hi 32 | lo 32] bf16 (128 B per pixel), a2_records [n][81][hi 64 |
// lo 64] (256 B per pixel), both device pointers; plus the packed conv1 scales and biases (host, 32 floats each) when
// conv1 runs on the int8 matrix cores (zeros otherwise).
extern "C" int rela_ffnet_debug_conv12_records(const rela_ffnet* n, int N, const uint8_t* s_dev, uint8_t* a1_records,
                                               uint8_t* a2_records, float* scale_host, float* bias_host, void* stream_) {
  RELA_CHECK(n && n->loaded && N >= 1 && s_dev && a1_records && a2_records, RELA_EINVAL, "rela_ffnet_debug_conv12_records: bad arguments");
  DeviceGuard g(n->device);
  hipStream_t s = (hipStream_t)stream_;
  const FFNetDev& d = n->d;
  TrunkJobs jobs{};
  jobs.n = 1;
  TrunkJob& j0 = jobs.j[0];
  j0.in0 = s_dev, j0.in1 = s_dev, j0.n_in0 = N;
  j0.B2 = (const uint4*)d.B2f, j0.B3 = (const uint4*)d.B3f;
  j0.b2 = d.b2, j0.b3 = d.b3;
  j0.W1d = d.W1d, j0.s1q = d.s1q, j0.b1q = d.b1q;
  j0.a1_out = a1_records, j0.a1_lo = 0, j0.n_a1 = N;
  j0.a2 = a2_records, j0.a3 = nullptr;
  j0.N = N, j0.block0 = 0, j0.nblocks = std::min(kNumCU, N);
  note_launch("conv12_i8_jobs");
  hipLaunchKernelGGL(conv12_i8_jobs, dim3(j0.nblocks), dim3(kThreads), Conv12I::LDS_TOTAL, s, jobs);
  RELA_HIP(hipStreamSynchronize(s));
  if (scale_host) RELA_HIP(hipMemcpy(scale_host, d.s1q, 32 * sizeof(float), hipMemcpyDeviceToHost));
  if (bias_host) RELA_HIP(hipMemcpy(bias_host, d.b1q, 32 * sizeof(float), hipMemcpyDeviceToHost));
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

extern "C" int rela_runtime_set_cu_reserve(int cus) {
  RELA_CHECK(cus >= 0 && cus <= 128, RELA_EINVAL, "rela_runtime_set_cu_reserve: 0..128 compute units");
  if (!std::getenv("RELA_CU_RESERVE")) rela_amd::g_cu_reserve.store(cus, std::memory_order_relaxed);
  return RELA_OK;
}
extern "C" int rela_ffnet_num_action(const rela_ffnet* n) { return n ? n->num_action : 0; }
namespace rela_amd {
int ffnet_load_impl(rela_ffnet* n, const rela_ffnet_params* p, int on_device, void* stream_, const FFNetExtraPacks& extra);
void ffnet_label_as_learner(rela_ffnet* n) { n->prof_names = kProfLearner; }
void ffnet_set_max_rows(rela_ffnet* n, int rows) { n->max_rows = rows; }
}  // namespace rela_amd
extern "C" uint64_t rela_ffnet_version(const rela_ffnet* n) { return n ? n->version : 0; }
extern "C" int rela_ffnet_set_precision(rela_ffnet* n, int mode) {
  RELA_CHECK(n && mode >= 0 && mode <= 2, RELA_EINVAL,
             "rela_ffnet_set_precision: mode must be 0 (f32 MFMA), 1 (split-bf16, 16-bit operands) or 2 (f32 as three bf16 parts)");
  n->precision = mode;
  n->version += 1;  // results of the two modes differ in the last bits: a cached forward must not be reused across them
  return RELA_OK;
}
extern "C" int rela_ffnet_precision(const rela_ffnet* n) { return n ? n->precision : 0; }

// split3 records of a2 / a3 (f32x3 mode, gemm_s3.h) live behind the f32 layout of ffnet_layout.h (and the fc split-K
// partial tiles): 6 bytes per value where the f32 tensors have 4
static inline int64_t ws_records_offset(int batch) {
  const int64_t b = batch > 0 ? batch : 0;
  const int64_t f = (int64_t)sizeof(float) * (kWsFloatsPerSample * b + (b < kFcSplitBelow ? kFcPartFloats : 0)) + 256;
  return (f + 255) & ~(int64_t)255;
}
extern "C" int64_t rela_ffnet_workspace_bytes(const rela_ffnet* n, int batch) {
  (void)n;
  const int64_t b = batch > 0 ? batch : 0;
  return ws_records_offset((int)b) + b * (kRec2Bytes + kRec3Bytes) + 256;
}

extern "C" int rela_ffnet_load(rela_ffnet* n, const rela_ffnet_params* p, int on_device, void* stream_) {
  return rela_amd::ffnet_load_impl(n, p, on_device, stream_, rela_amd::FFNetExtraPacks{});
}
int rela_amd::ffnet_load_extra(rela_ffnet* n, const rela_ffnet_params* p, void* stream_, const FFNetExtraPacks& extra) {
  return ffnet_load_impl(n, p, 1, stream_, extra);
}

int rela_amd::ffnet_load_impl(rela_ffnet* n, const rela_ffnet_params* p, int on_device, void* stream_,
                              const FFNetExtraPacks& extra) {
  RELA_CHECK(n && p, RELA_EINVAL, "rela_ffnet_load: bad arguments");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(n->device);
  const int A = n->num_action;
  const size_t cnt[12] = {32 * 256, 32, 64 * 512, 64, 64 * 576, 64, (size_t)512 * 3136, 512, 512, 1, (size_t)A * 512,
                          (size_t)A};
  const float* src[12] = {p->conv1_w, p->conv1_b, p->conv2_w, p->conv2_b, p->conv3_w, p->conv3_b,
                          p->fc_w,    p->fc_b,    p->v_w,     p->v_b,     p->a_w,     p->a_b};
  const float* dv[12];
  float* tmp = nullptr;
  if (on_device) {
    for (int i = 0; i < 12; ++i) {
      RELA_CHECK(src[i], RELA_EINVAL, "rela_ffnet_load: parameter %d is NULL", i);
      dv[i] = src[i];
    }
  } else {
    size_t total = 0;
    for (int i = 0; i < 12; ++i) total += cnt[i];
    RELA_HIP(hipMalloc(&tmp, sizeof(float) * total));
    size_t off = 0;
    for (int i = 0; i < 12; ++i) {
      RELA_CHECK(src[i], RELA_EINVAL, "rela_ffnet_load: parameter %d is NULL", i);
      RELA_HIP(hipMemcpyAsync(tmp + off, src[i], sizeof(float) * cnt[i], hipMemcpyHostToDevice, s));
      dv[i] = tmp + off;
      off += cnt[i];
    }
  }
  {
    PackAllArgs a{};
    for (int i = 0; i < 12; ++i) a.p[i] = dv[i];
    a.B1 = reinterpret_cast<uint16_t*>(n->d.B1);
    a.B2f = reinterpret_cast<uint16_t*>(n->d.B2f), a.B3f = reinterpret_cast<uint16_t*>(n->d.B3f);
    a.Bff = reinterpret_cast<uint16_t*>(n->d.Bff);
    a.B2 = n->d.B2, a.B3 = n->d.B3, a.Bf = n->d.Bf, a.BfT = n->d.BfT, a.Bh = n->d.Bh, a.Bhp = n->d.Bhp;
    a.b1 = n->d.b1, a.b2 = n->d.b2, a.b3 = n->d.b3, a.bf = n->d.bf, a.bh = n->d.bh, a.A = A;
    a.w2p = extra.w2p, a.w3p = extra.w3p, a.wfcp = extra.wfcp;
    a.W1d = reinterpret_cast<uint8_t*>(n->d.W1d), a.s1q = n->d.s1q, a.b1q = n->d.b1q;
    a.B2e = reinterpret_cast<uint16_t*>(n->d.B2e), a.B3e = reinterpret_cast<uint16_t*>(n->d.B3e);
    a.Bfe = reinterpret_cast<uint16_t*>(n->d.Bfe);
    // (one thread per bf16 TRIPLE of the f32-accurate layouts: a third of their 2-byte elements)
    const int64_t emu2 = f32emu::packed_u4<f32emu::ProbConv2>() * 8 / 3, emu3 = f32emu::packed_u4<f32emu::ProbConv3>() * 8 / 3,
                  emuf = f32emu::packed_u4<f32emu::ProbFc>() * 8 / 3;
    const int64_t elems[20] = {2 * 8 * 64 * 8, 0, 4 * 128 * 64, 4 * 144 * 64, (int64_t)32 * 784 * 64,
                               (int64_t)3136 * 512, 2 * 128 * 64, 2 * 4 * 32 * 64,
                               (int64_t)Conv2F::CT * Conv2F::KS * 64 * 8, (int64_t)Conv3F::CT * Conv3F::KS * 64 * 8,
                               (int64_t)32 * FcFast::KS * 64 * 8, 672, 32,
                               extra.w2p ? 64 * 512 : 0, extra.w3p ? 64 * 576 : 0, extra.wfcp ? (int64_t)512 * 3136 : 0,
                               32 * 256, emu2, emu3, emuf};
    // a net whose owner never runs large batches (a learner re-packs after every step) skips the two fc layouts
    // only large batches read: Bf (f32 fragments, N >= kFcSplitBelow) and Bff (bf16 fragments, N >= kFastMinN)
    int64_t el[20];
    for (int jn = 0; jn < 20; ++jn) el[jn] = elems[jn];
    // (... nor the layouts of the f32x3 mode its batches are too small for)
    if (n->max_rows > 0 && n->max_rows < kEmuConvMinN) el[17] = el[18] = 0;
    if (n->max_rows > 0 && n->max_rows < kEmuFcMinN) el[19] = 0;
    if (n->max_rows > 0 && n->max_rows < kFcSplitBelow) el[4] = 0;
    if (n->max_rows > 0 && n->max_rows < kFastTrunkMinN) el[10] = 0;  // (the split-K fc_bf16s serves 128 rows and up)
    n->bft_stale = false;
    if (n->max_rows >= kFastTrunkMinN && on_device && n->precision == 1) {  // (the owner's buffer outlives this call)
      el[5] = 0;
      n->bft_stale = true;
      n->bft_src = dv[6];
    }
    a.first[0] = 0;
    for (int jn = 0; jn < 20; ++jn) a.first[jn + 1] = a.first[jn] + (int)ceil_div(el[jn], 256);
    hipLaunchKernelGGL(pack_ffnet_all, dim3(a.first[20]), dim3(256), 0, s, a);
  }
  RELA_LAUNCH_CHECK();
  if (tmp) {
    RELA_HIP(hipStreamSynchronize(s));
    (void)hipFree(tmp);
  }
  n->loaded = true;
  n->version += 1;
  return RELA_OK;
}

extern "C" int rela_ffnet_forward(const rela_ffnet* n, int N, const uint8_t* s_dev, const float* legal_dev,
                                  float* q_dev, void* ws, int64_t ws_bytes, void* stream_) {
  return rela_amd::ffnet_forward_mode(n, N, s_dev, legal_dev, q_dev, ws, ws_bytes, stream_, -1);
}

int rela_amd::ffnet_forward_mode(const rela_ffnet* n, int N, const uint8_t* s_dev, const float* legal_dev, float* q_dev,
                                 void* ws, int64_t ws_bytes, void* stream_, int mode) {
  RELA_CHECK(n && n->loaded, RELA_ESTATE, "rela_ffnet_forward: parameters were never loaded");
  RELA_CHECK(N >= 1 && s_dev && legal_dev && q_dev && ws, RELA_EINVAL, "rela_ffnet_forward: bad arguments");
  RELA_CHECK(ws_bytes >= rela_ffnet_workspace_bytes(n, N), RELA_EINVAL,
             "rela_ffnet_forward: workspace of %lld bytes is too small for batch %d", (long long)ws_bytes, N);
  RELA_CHECK(((uintptr_t)s_dev & 15) == 0 && ((uintptr_t)ws & 15) == 0, RELA_EINVAL,
             "rela_ffnet_forward: s_dev and workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream_;
  float* a1 = static_cast<float*>(ws);
  float* a2 = a1 + kA1 * N;
  float* a3 = a2 + kA2 * N;
  float* h = a3 + kA3 * N;
  float* ha = h + kH * N;
  const FFNetDev& d = n->d;
  const char* const* names = n->prof_names ? n->prof_names : kProfActor;
  const char* name12 = n->prof_names ? "learner_fwd_conv12" : "conv12_fused";  // conv1 -> conv2 in one launch
  RELA_CHECK(n->max_rows <= 0 || N <= n->max_rows, RELA_EINVAL,
             "rela_ffnet_forward: batch %d on a net whose owner declared at most %d rows", N, n->max_rows);
  // (a net packed for small batches only has no bf16 fc fragments: it keeps the f32 fc whatever the threshold says)
  const int fast_min_n = (n->max_rows > 0 && n->max_rows < kFastMinN) ? n->max_rows + 1 : kFastMinN;
  const bool keep_f32 = mode == 3;  // f32x3 that also leaves a1 / a2 / a3 in channel-last f32 (the learner's online(obs) pass)
  const int precision = mode < 0 ? n->precision : (mode == 3 ? 2 : mode);
  // Between kFastTrunkMinN and kFastMinN rows the convolutions still win on split-bf16 MFMA (N = 512: 39 us against
  // 90 us in f32) but fc_bf16s has too few blocks (55 us against the 24 us of the f32 split-K GEMM): the trunk runs
  // fast, a3 is turned back into f32 in place and fc takes the f32 path.
  const bool fast_trunk_only = precision == 1 && N < fast_min_n && N >= kFastTrunkMinN;
  // ... and (r3) fc too, as a split-K launch of fc_bf16s, when this net packs the bf16 fc fragments
  const bool fc_split_bf16 = fast_trunk_only && !(n->max_rows > 0 && n->max_rows < kFastTrunkMinN);
  const bool emu_conv = precision == 2 && N >= kEmuConvMinN && N <= kEmuMaxN && !(n->max_rows > 0 && n->max_rows < kEmuConvMinN);
  if (fast_trunk_only) {
    uint8_t *r2 = reinterpret_cast<uint8_t*>(a2), *r3 = reinterpret_cast<uint8_t*>(a3);
    {
      ProfScope prof(name12, s);
      note_launch("conv12_i8"); hipLaunchKernelGGL(conv12_i8, dim3(persistent_blocks(N)), dim3(kThreads), Conv12I::LDS_TOTAL, s, s_dev,
                         (const uint4*)d.W1d, (const float*)d.s1q, (const float*)d.b1q, (const uint4*)d.B2f, (const float*)d.b2, r2, N);
    }
    {
      ProfScope prof(names[2], s);
      note_launch("conv_bf16s<Conv3F>"); hipLaunchKernelGGL(conv_bf16s<Conv3F>, dim3(persistent_blocks(ceil_div(N, Conv3F::S))), dim3(kThreads),
                         Conv3F::LDS_TOTAL, s, (const uint8_t*)r2, (const uint4*)d.B3f, (const float*)d.b3, r3, N);
      if (!fc_split_bf16) {
        note_launch("unsplit_records64"); hipLaunchKernelGGL(unsplit_records64, dim3(ceil_div((int64_t)N * 49, 4)), dim3(256), 0, s, r3, (int64_t)N * 49);
      }
    }
    if (fc_split_bf16) {
      // fc on split-bf16 MFMA straight from a3's records, the contraction split over blockIdx.z so that ~256 blocks
      // run (r3; the f32 split-K GEMM after an unsplit pass took 21 + 6 + 6 us at 512 rows); a3 STAYS in records
      const int rb = ceil_div(N, FcFast::BM);
      int slices = std::max(1, std::min(FcFast::NPOS, kNumCU / (4 * rb)));
      slices = std::min(slices, (int)(8192 / N));
      const int per = ceil_div(FcFast::NPOS, slices);
      slices = ceil_div(FcFast::NPOS, per);
      float* part = ha + kHA * N;
      part += (64 - ((part - static_cast<float*>(ws)) & 63)) & 63;
      {
        ProfScope prof(names[3], s);
        note_launch("fc_bf16s (split-K)");
        hipLaunchKernelGGL((fc_bf16s<FcFast, true>), fc_grid_xcd(rb, slices), dim3(kThreads), FcFast::LDS_BYTES, s,
                           (const uint8_t*)r3, (const uint4*)d.Bff, (const float*)d.bf, part, N, per);
      }
      note_launch("fc_reduce"); hipLaunchKernelGGL(fc_reduce, dim3(ceil_div(N * 128, 256)), dim3(256), 0, s, (const float*)part, slices, N,
                         (const float*)d.bf, h);
    }
  }
  if (precision == 1 && N >= fast_min_n) {
    // split-bf16 fast path: a1 / a2 / a3 hold split records (same bytes as the f32 tensors they replace)
    uint8_t *r2 = reinterpret_cast<uint8_t*>(a2), *r3 = reinterpret_cast<uint8_t*>(a3);
    {  // conv1 (int8 matrix cores) -> conv2 (split-bf16), fused per frame through LDS
      ProfScope prof(name12, s);
      note_launch("conv12_i8"); hipLaunchKernelGGL(conv12_i8, dim3(persistent_blocks(N)), dim3(kThreads), Conv12I::LDS_TOTAL, s, s_dev,
                         (const uint4*)d.W1d, (const float*)d.s1q, (const float*)d.b1q, (const uint4*)d.B2f, (const float*)d.b2, r2, N);
    }
    {
      ProfScope prof(names[2], s);
      note_launch("conv_bf16s<Conv3F>"); hipLaunchKernelGGL(conv_bf16s<Conv3F>, dim3(persistent_blocks(ceil_div(N, Conv3F::S))), dim3(kThreads),
                         Conv3F::LDS_TOTAL, s, (const uint8_t*)r2, (const uint4*)d.B3f, (const float*)d.b3, r3, N);
    }
    {
      ProfScope prof(names[3], s);
      note_launch("fc_bf16s");
      hipLaunchKernelGGL(fc_bf16s<FcFast>, fc_grid_xcd(ceil_div(N, FcFast::BM), 1), dim3(kThreads), FcFast::LDS_BYTES, s,
                           (const uint8_t*)r3, (const uint4*)d.Bff, (const float*)d.bf, h, N, 0);
    }
  } else {
  if (emu_conv) {
    // f32x3 (precision 2; mode 3 = the learner's pass that also leaves a1 / a2 / a3 in f32 for the backward kernels): the
    // trunk on split3 records.  conv1 -> conv2 fused per frame; conv3 from LDS images with resident weights; fc below.
    uint8_t* rec2 = static_cast<uint8_t*>(ws) + ws_records_offset(N);
    uint8_t* rec3 = rec2 + (int64_t)N * kRec2Bytes;
    {
      ProfScope prof(name12, s);
      note_launch("conv12_s3");
      if (keep_f32)
        hipLaunchKernelGGL(s3::conv12_s3<true>, dim3(persistent_blocks(N)), dim3(s3::Conv12S::kT), s3::Conv12S::LDS_TOTAL, s, s_dev,
                           (const uint4*)d.W1d, (const float*)d.s1q, (const float*)d.b1q, (const uint4*)d.B2e, (const float*)d.b2,
                           rec2, a1, N);
      else
        hipLaunchKernelGGL(s3::conv12_s3<false>, dim3(persistent_blocks(N)), dim3(s3::Conv12S::kT), s3::Conv12S::LDS_TOTAL, s, s_dev,
                           (const uint4*)d.W1d, (const float*)d.s1q, (const float*)d.b1q, (const uint4*)d.B2e, (const float*)d.b2,
                           rec2, (float*)nullptr, N);
    }
    {
      ProfScope prof(names[2], s);
      note_launch("conv3_img_s3");
      s3::launch_conv3_img(rec2, d.B3e, d.b3, rec3, N, s, persistent_blocks(N));
    }
    {
      ProfScope prof(names[3], s);
      note_launch("gemm_s3<fc>");
      const s3::Plan pl = s3::plan<s3::ProbFc>(N, N < kFcSplitBelow && (int64_t)8 * N <= 8192);  // (the partial tiles' space)
      if (pl.slices > 1) {  // small batches: the contraction split over blocks, fc_reduce adds slices + bias + ReLU
        float* part = ha + kHA * N;
        part += (64 - ((part - static_cast<float*>(ws)) & 63)) & 63;
        s3::launch<s3::ProbFc, s3::kEpiRaw>(rec3, d.Bfe, d.bf, part, N, s, pl);
        note_launch("fc_reduce"); hipLaunchKernelGGL(fc_reduce, dim3(ceil_div(N * 128, 256)), dim3(256), 0, s, (const float*)part, pl.slices, N,
                           (const float*)d.bf, h);
      } else {
        s3::launch<s3::ProbFc, s3::kEpiRelu>(rec3, d.Bfe, d.bf, h, N, s);
      }
    }
    if (keep_f32) {
      note_launch("unsplit_s3");
      const int64_t p2 = (int64_t)N * 81, p3 = (int64_t)N * 49;
      hipLaunchKernelGGL(s3::unsplit_s3<64>, dim3((unsigned)ceil_div(p2 * 16, 256)), dim3(256), 0, s, (const uint8_t*)rec2, a2, p2);
      hipLaunchKernelGGL(s3::unsplit_s3<64>, dim3((unsigned)ceil_div(p3 * 16, 256)), dim3(256), 0, s, (const uint8_t*)rec3, a3, p3);
    }
  } else {
  if (!fast_trunk_only) {
  {
    ProfScope prof(names[0], s);
    note_launch("conv1_bf16x3"); hipLaunchKernelGGL(conv1_bf16x3, dim3(ceil_div(N, Conv1B::S)), dim3(kThreads), Conv1B::LDS_BYTES, s, s_dev,
                       d.B1, d.b1, a1, N);
  }
  {
    ProfScope prof(names[1], s);
    launch_conv<Conv2>(a1, d.B2, d.b2, a2, N, s);
  }
  {
    ProfScope prof(names[2], s);
    launch_conv<Conv3>(a2, d.B3, d.b3, a3, N, s);
  }
  }
  }
  if (fc_split_bf16 || emu_conv) {
    // (h is already there)
  } else if (N < kFcSplitBelow) {
    const int splits = fc_splits(N);
    float* part = ha + kHA * N;
    part += (64 - ((part - static_cast<float*>(ws)) & 63)) & 63;  // 256-byte aligned (float4 loads)
    if (n->bft_stale) {  // (see rela_ffnet::bft_stale)
      PackAllArgs a{};
      a.p[6] = n->bft_src, a.BfT = d.BfT;
      for (int jn = 0; jn < 20; ++jn) a.first[jn + 1] = a.first[jn] + (jn == 5 ? (int)ceil_div((int64_t)3136 * 512, 256) : 0);
      hipLaunchKernelGGL(pack_ffnet_all, dim3(a.first[20]), dim3(256), 0, s, a);
      n->bft_stale = false;
    }
    ProbFcFwd p{};
    p.M = N, p.N = 512, p.K = 3136;
    p.a3 = a3, p.wt = d.BfT, p.part = part;
    gemm::launch_gemm<TileFcSmall>(p, splits, s, names[3]);
    note_launch("fc_reduce"); hipLaunchKernelGGL(fc_reduce, dim3(ceil_div(N * 128, 256)), dim3(256), 0, s, (const float*)part, splits, N,
                       (const float*)d.bf, h);
  } else {
    ProfScope prof(names[3], s);
    note_launch("gemm_mfma<GemmFc> (f32)");
    if (prefer_bm112(N, GemmFc::CT / GemmFc::CTB, GemmFc::BM))
      hipLaunchKernelGGL(gemm_mfma<GemmFc112>, dim3(GemmFc112::CT / GemmFc112::CTB, ceil_div(N, GemmFc112::BM)),
                         dim3(kThreads), 0, s, (const float*)a3, (const float*)nullptr, (const float*)d.Bf,
                         (const float*)d.bf, h, (const float*)nullptr, (float*)nullptr, N);
    else
      hipLaunchKernelGGL(gemm_mfma<GemmFc>, dim3(GemmFc::CT / GemmFc::CTB, ceil_div(N, GemmFc::BM)), dim3(kThreads), 0,
                         s, (const float*)a3, (const float*)nullptr, (const float*)d.Bf, (const float*)d.bf, h,
                         (const float*)nullptr, (float*)nullptr, N);
  }
  }
  {
    ProfScope prof(names[4], s);
    note_launch("heads_duel"); hipLaunchKernelGGL(heads_duel, dim3(ceil_div(N, kHeadRows)), dim3(256), 0, s, (const float*)h, (const float*)d.Bhp,
                       (const float*)d.bh, legal_dev, ha, q_dev, N, n->num_action);
  }
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

// ---- the Ape-X learner's three forwards in split-bf16, one launch per layer (csrc/learner.hip) ----------------
namespace rela_amd {
namespace {
// split records -> f32, in place, for the rows the backward kernels read: a1 [rows][400] records of 32 channels
// (64 B hi | 64 B lo), a2 [rows][81] and a3 [rows][49] records of 64 channels (128 B hi | 128 B lo).  A record is read
// whole by the lanes that then overwrite it (one wave per 64-channel record, half a wave per 32-channel one).
__global__ void unsplit_trunk_rows(uint8_t* __restrict__ a1, uint8_t* __restrict__ a2, uint8_t* __restrict__ a3, int rows,
                                   int blocks1, int blocks2) {
  // a thread owns four channels of a record: 8 B of hi + 8 B of lo in, 16 B of f32 out; the 8 (a1) or 16 (a2, a3) lanes
  // of a record sit in one wave, which reads all its records before it overwrites any
  int b = blockIdx.x;
  const int t = threadIdx.x;
  int64_t rec;
  int q, C;
  uint8_t* base;
  bool ok;
  if (b < blocks1) {
    rec = (int64_t)b * 32 + (t >> 3), q = t & 7, C = 32, base = a1, ok = rec < (int64_t)rows * 400;
  } else {
    b -= blocks1;
    base = a2;
    int64_t pixels = (int64_t)rows * 81;
    if (b >= blocks2) b -= blocks2, base = a3, pixels = (int64_t)rows * 49;
    rec = (int64_t)b * 16 + (t >> 4), q = t & 15, C = 64, ok = rec < pixels;
  }
  uint8_t* r = base + (ok ? rec : 0) * (C * 4);
  const uint2 hi = *reinterpret_cast<const uint2*>(r + q * 8), lo = *reinterpret_cast<const uint2*>(r + C * 2 + q * 8);
  float4 v;
  v.x = __uint_as_float(hi.x << 16) + __uint_as_float(lo.x << 16);
  v.y = __uint_as_float(hi.x & 0xffff0000u) + __uint_as_float(lo.x & 0xffff0000u);
  v.z = __uint_as_float(hi.y << 16) + __uint_as_float(lo.y << 16);
  v.w = __uint_as_float(hi.y & 0xffff0000u) + __uint_as_float(lo.y & 0xffff0000u);
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ok) *reinterpret_cast<float4*>(r + q * 16) = v;
}
}  // namespace

bool ffnet_learner_forward_ok(const rela_ffnet* on, const rela_ffnet* tg, int B) {
  const auto packs_bf16_fc = [](const rela_ffnet* n) { return !(n->max_rows > 0 && n->max_rows < kFastTrunkMinN); };
  return on && tg && on->loaded && tg->loaded && B >= kFastTrunkMinN && 2 * B < kFcSplitBelow && packs_bf16_fc(on) &&
         packs_bf16_fc(tg);
}

// online over [s ; s'] (2 B rows: rows < B are s) and target over s' (B rows): conv1 -> conv2 of both nets in ONE
// launch (block ranges in proportion to the rows), conv3 likewise, fc as split-K launches of fc_bf16s straight from
// a3's records, the dueling heads per Q table.  ws_on has the ffnet_ws layout for 2 B rows, ws_tg for B rows; a1 (rows
// < B only), a2 and a3 hold split RECORDS: ffnet_learner_unsplit turns the rows < B into f32 for the backward pass.
int ffnet_learner_forward(const rela_ffnet* on, const rela_ffnet* tg, int B, const uint8_t* s_obs, const uint8_t* s_next,
                          const float* legal, const float* nlegal, float* q_on, float* q_no, float* q_nt, void* ws_on,
                          void* ws_tg, int64_t ws_bytes, hipStream_t s) {
  RELA_CHECK(ffnet_learner_forward_ok(on, tg, B), RELA_ESTATE, "ffnet_learner_forward: nets not loaded / batch %d unsupported", B);
  RELA_CHECK(ws_bytes >= rela_ffnet_workspace_bytes(on, 2 * B), RELA_EINVAL, "ffnet_learner_forward: workspace too small");
  RELA_CHECK(((uintptr_t)s_obs & 15) == 0 && ((uintptr_t)s_next & 15) == 0, RELA_EINVAL, "ffnet_learner_forward: frames must be 16-byte aligned");
  const FFNetWs w = ffnet_ws(ws_on, 2 * B), wt = ffnet_ws(ws_tg, B);
  TrunkJobs jobs{};
  jobs.n = 2;
  const int total = std::min(kNumCU, 3 * B);
  const int nb0 = std::max(1, std::min(total - 1, (int)((int64_t)total * 2 / 3)));
  TrunkJob& j0 = jobs.j[0];
  j0.in0 = s_obs, j0.in1 = s_next, j0.n_in0 = B;
  j0.B2 = (const uint4*)on->d.B2f, j0.B3 = (const uint4*)on->d.B3f;
  j0.b2 = on->d.b2, j0.b3 = on->d.b3;
  j0.W1d = on->d.W1d, j0.s1q = on->d.s1q, j0.b1q = on->d.b1q;
  j0.a1_out = reinterpret_cast<uint8_t*>(w.a1), j0.a1_lo = 0, j0.n_a1 = B;
  j0.a2 = reinterpret_cast<uint8_t*>(w.a2), j0.a3 = reinterpret_cast<uint8_t*>(w.a3);
  j0.N = 2 * B, j0.block0 = 0, j0.nblocks = nb0;
  TrunkJob& j1 = jobs.j[1];
  j1.in0 = s_next, j1.in1 = s_next, j1.n_in0 = B;
  j1.B2 = (const uint4*)tg->d.B2f, j1.B3 = (const uint4*)tg->d.B3f;
  j1.b2 = tg->d.b2, j1.b3 = tg->d.b3;
  j1.W1d = tg->d.W1d, j1.s1q = tg->d.s1q, j1.b1q = tg->d.b1q;
  j1.a1_out = nullptr, j1.a1_lo = 0, j1.n_a1 = 0;
  j1.a2 = reinterpret_cast<uint8_t*>(wt.a2), j1.a3 = reinterpret_cast<uint8_t*>(wt.a3);
  j1.N = B, j1.block0 = nb0, j1.nblocks = total - nb0;
  {
    ProfScope prof("learner_fwd_conv12", s);
    note_launch("conv12_i8_jobs");
    hipLaunchKernelGGL(conv12_i8_jobs, dim3(total), dim3(kThreads), Conv12I::LDS_TOTAL, s, jobs);
  }
  {
    // conv3 walks groups of Conv3F::S frames: the same block ranges serve (a job never has more blocks than groups
    // would keep busy only below 2 * 256 frames per job, where the surplus blocks leave at once)
    ProfScope prof("learner_fwd_conv3", s);
    note_launch("conv3_bf16s_jobs");
    hipLaunchKernelGGL(conv3_bf16s_jobs, dim3(total), dim3(kThreads), Conv3F::LDS_TOTAL, s, jobs);
  }
  auto fc = [&](const rela_ffnet* n, const FFNetWs& ww, int N, void* wsp) {
    const int rb = ceil_div(N, FcFast::BM);
    int slices = std::max(1, std::min(FcFast::NPOS, kNumCU / (4 * rb)));
    slices = std::min(slices, 8192 / N);
    const int per = ceil_div(FcFast::NPOS, slices);
    slices = ceil_div(FcFast::NPOS, per);
    float* part = ww.ha + kHA * N;
    part += (64 - ((part - static_cast<float*>(wsp)) & 63)) & 63;
    {
      ProfScope prof("learner_fwd_fc", s);
      note_launch("fc_bf16s (split-K)");
      hipLaunchKernelGGL((fc_bf16s<FcFast, true>), fc_grid_xcd(rb, slices), dim3(kThreads), FcFast::LDS_BYTES, s,
                         (const uint8_t*)ww.a3, (const uint4*)n->d.Bff, (const float*)n->d.bf, part, N, per);
    }
    note_launch("fc_reduce");
    hipLaunchKernelGGL(fc_reduce, dim3(ceil_div(N * 128, 256)), dim3(256), 0, s, (const float*)part, slices, N,
                       (const float*)n->d.bf, ww.h);
  };
  fc(on, w, 2 * B, ws_on);
  fc(tg, wt, B, ws_tg);
  {
    ProfScope prof("learner_fwd_heads", s);
    const int A = on->num_action, nb = ceil_div(B, kHeadRows);
    HeadJobs jb{};
    const float* hs[3] = {w.h, w.h + (size_t)B * kH, wt.h};
    float* has[3] = {w.ha, w.ha + (size_t)B * kHA, wt.ha};
    const rela_ffnet* nets[3] = {on, on, tg};
    const float* legals[3] = {legal, nlegal, nlegal};
    float* qs[3] = {q_on, q_no, q_nt};
    for (int k = 0; k < 3; ++k) {
      jb.h[k] = hs[k], jb.Bhp[k] = nets[k]->d.Bhp, jb.bias[k] = nets[k]->d.bh, jb.legal[k] = legals[k];
      jb.ha[k] = has[k], jb.q[k] = qs[k], jb.N[k] = B, jb.first[k] = k * nb;
    }
    jb.first[3] = 3 * nb;
    note_launch("heads_duel_jobs");
    hipLaunchKernelGGL(heads_duel_jobs, dim3(3 * nb), dim3(256), 0, s, jb, A);
  }
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

int trunk_unsplit_rows(float* a1, float* a2, float* a3, int rows, hipStream_t s) {
  const int b1 = (int)ceil_div((int64_t)rows * 400, 32), b2 = (int)ceil_div((int64_t)rows * 81, 16),
            b3 = (int)ceil_div((int64_t)rows * 49, 16);
  ProfScope prof("learner_unsplit", s);
  note_launch("unsplit_trunk_rows");
  hipLaunchKernelGGL(unsplit_trunk_rows, dim3(b1 + b2 + b3), dim3(256), 0, s, reinterpret_cast<uint8_t*>(a1),
                     reinterpret_cast<uint8_t*>(a2), reinterpret_cast<uint8_t*>(a3), rows, b1, b2);
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

int ffnet_learner_unsplit(int B, void* ws_on, hipStream_t s) {
  const FFNetWs w = ffnet_ws(ws_on, 2 * B);
  return trunk_unsplit_rows(w.a1, w.a2, w.a3, B, s);
}
}  // namespace rela_amd

// =====================================================================================
// AtariLSTMNet (pyrela/net.py:58-163): trunk -> LSTM gates + cell (one fused GEMM) -> heads
// =====================================================================================
struct rela_lstmnet {
  int device = 0;
  int num_action = 0;
  FFNetDev d;            // B1..b3 (trunk) and Bh/bh (heads); Bf/bf unused
  float* Bl = nullptr;   // lstm frags [128][912][64]
  float* bl = nullptr;   // b_ih + b_hh, permuted [2048]
  uint8_t* Wrec = nullptr;  // weight_ih_l0 as rec64 rows in the permuted column order [2048][49][64 hi | 64 lo]
  uint4* Wx3 = nullptr;     // ... as three-part bf16 fragments in the same column order (gemm_s3.h; f32x3 mode)
  bool loaded = false;
  uint64_t version = 0;  // bumped by every load
  // 0 = exact f32; 1 = split-bf16 conv trunk and, from kFastMinN rows up, the x part of the gate GEMM on split-bf16
  // MFMA too (the recurrent part and the cell stay f32); 2 = f32x3: from 512 rows up the conv trunk AND the x part of the
  // gate GEMM (3136 -> 2048: three quarters of the step's FLOPs) with three-part operands on the bf16 matrix cores
  // (conv12_s3.h, conv_img_s3.h, gemm_s3.h: f32 accuracy), the recurrent part, the cell and the heads as in mode 0
  int precision = 0;
};

namespace {
constexpr int64_t kLstmWsFloats = kA1 + kA2 + kA3 + kHA + 2048;  // + the x part of the gates (bf16x2 / f32x3 modes)
using ProbGateX3 = s3::ProbFcT<2048>;
// split3 records of a2 / a3 behind the f32 layout (f32x3 mode)
inline int64_t lstm_ws_records_offset(int batch) {
  return (((int64_t)sizeof(float) * kLstmWsFloats * (batch > 0 ? batch : 0) + 256) + 255) & ~(int64_t)255;
}
}

extern "C" int rela_lstmnet_create(rela_lstmnet** out, int num_action, int device) {
  RELA_CHECK(out && num_action >= 1 && num_action <= 31, RELA_EINVAL,
             "rela_lstmnet_create: num_action must be in 1..31 (got %d)", num_action);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_last_error("rela_lstmnet_create: HIP device %d not available (%d visible); there is no CPU path", device, ndev);
    return RELA_ENODEV;
  }
  DeviceGuard g(device);
  auto* n = new rela_lstmnet();
  n->device = device;
  n->num_action = num_action;
  FFNetDev& d = n->d;
  RELA_HIP(hipMalloc(&d.B1, sizeof(uint4) * Conv1B::FRAG_UINT4));
  RELA_HIP(hipMalloc(&d.b1, sizeof(float) * 32));
  RELA_HIP(hipMalloc(&d.B2, sizeof(float) * 4 * 128 * 64));
  RELA_HIP(hipMalloc(&d.b2, sizeof(float) * 64));
  RELA_HIP(hipMalloc(&d.B3, sizeof(float) * 4 * 144 * 64));
  RELA_HIP(hipMalloc(&d.b3, sizeof(float) * 64));
  RELA_HIP(hipMalloc(&d.Bh, sizeof(float) * 2 * 128 * 64));
  RELA_HIP(hipMalloc(&d.bh, sizeof(float) * 32));
  RELA_HIP(hipMalloc(&n->Bl, sizeof(float) * (size_t)GemmLstm::CT * GemmLstm::KS * 64));
  RELA_HIP(hipMalloc(&n->Wrec, (size_t)2048 * 49 * 256));
  RELA_HIP(hipMalloc(&n->Wx3, sizeof(uint4) * f32emu::packed_u4<f32emu::ProbGateX>()));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&s3::conv12_s3<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               s3::Conv12S::LDS_TOTAL));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&s3::conv12_s3<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               s3::Conv12S::LDS_TOTAL));
  RELA_HIP(hipMalloc(&n->bl, sizeof(float) * 2048));
  RELA_HIP(hipMalloc(&d.W1d, sizeof(uint4) * Conv12I::W1_UINT4));
  RELA_HIP(hipMalloc(&d.s1q, sizeof(float) * 32));
  RELA_HIP(hipMalloc(&d.b1q, sizeof(float) * 32));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_i8), hipFuncAttributeMaxDynamicSharedMemorySize,
                               Conv12I::LDS_TOTAL));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_i8_jobs), hipFuncAttributeMaxDynamicSharedMemorySize,
                               Conv12I::LDS_TOTAL));
  RELA_HIP(hipMalloc(&d.B2f, sizeof(uint4) * Conv2F::CT * Conv2F::KS * 2 * 64));
  RELA_HIP(hipMalloc(&d.B3f, sizeof(uint4) * Conv3F::CT * Conv3F::KS * 2 * 64));
  RELA_HIP(hipMalloc(&d.B2e, sizeof(uint4) * f32emu::packed_u4<f32emu::ProbConv2>()));
  RELA_HIP(hipMalloc(&d.B3e, sizeof(uint4) * f32emu::packed_u4<f32emu::ProbConv3>()));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16s<Conv3F>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Conv3F::LDS_TOTAL));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1_bf16x3),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Conv1B::LDS_BYTES));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma<Conv2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Conv2::LDS_BYTES));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma<Conv3>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Conv3::LDS_BYTES));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_bstat<Conv2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Conv2::LDS_BYTES));
  RELA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_bstat<Conv3>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Conv3::LDS_BYTES));
  *out = n;
  return RELA_OK;
}

extern "C" void rela_lstmnet_destroy(rela_lstmnet* n) {
  if (!n) return;
  DeviceGuard g(n->device);
  (void)hipDeviceSynchronize();
  void* ps[] = {n->d.B1, n->d.b1, n->d.B2, n->d.b2, n->d.B3, n->d.b3, n->d.Bh, n->d.bh, n->Bl, n->bl,
                n->d.B2f, n->d.B3f, n->Wrec, n->d.W1d, n->d.s1q, n->d.b1q, n->d.B2e, n->d.B3e, n->Wx3};
  for (void* p : ps) (void)hipFree(p);
  delete n;
}

extern "C" int rela_lstmnet_set_precision(rela_lstmnet* n, int mode) {
  RELA_CHECK(n && mode >= 0 && mode <= 2, RELA_EINVAL,
             "rela_lstmnet_set_precision: mode must be 0 (f32), 1 (bf16x2) or 2 (f32x3: conv2 / conv3 of the trunk with three-part operands)");
  if (n->precision != mode) n->version += 1;  // cached forwards of the other mode must not be reused
  n->precision = mode;
  return RELA_OK;
}
extern "C" int rela_lstmnet_precision(const rela_lstmnet* n) { return n ? n->precision : 0; }

extern "C" int rela_lstmnet_num_action(const rela_lstmnet* n) { return n ? n->num_action : 0; }
extern "C" uint64_t rela_lstmnet_version(const rela_lstmnet* n) { return n ? n->version : 0; }

extern "C" int64_t rela_lstmnet_workspace_bytes(const rela_lstmnet* n, int batch) {
  (void)n;
  return lstm_ws_records_offset(batch) + (int64_t)(batch > 0 ? batch : 0) * (kRec2Bytes + kRec3Bytes) + 256;
}

extern "C" int rela_lstmnet_load(rela_lstmnet* n, const rela_lstmnet_params* p, int on_device, void* stream_) {
  RELA_CHECK(n && p, RELA_EINVAL, "rela_lstmnet_load: bad arguments");
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(n->device);
  const int A = n->num_action;
  const size_t cnt[14] = {32 * 256, 32, 64 * 512, 64, 64 * 576, 64, (size_t)2048 * 3136, (size_t)2048 * 512, 2048, 2048,
                          512, 1, (size_t)A * 512, (size_t)A};
  const float* src[14] = {p->conv1_w, p->conv1_b, p->conv2_w, p->conv2_b, p->conv3_w, p->conv3_b, p->w_ih,
                          p->w_hh,    p->b_ih,    p->b_hh,    p->v_w,     p->v_b,     p->a_w,     p->a_b};
  const float* dv[14];
  float* tmp = nullptr;
  if (on_device) {
    for (int i = 0; i < 14; ++i) {
      RELA_CHECK(src[i], RELA_EINVAL, "rela_lstmnet_load: parameter %d is NULL", i);
      dv[i] = src[i];
    }
  } else {
    size_t total = 0;
    for (int i = 0; i < 14; ++i) total += cnt[i];
    RELA_HIP(hipMalloc(&tmp, sizeof(float) * total));
    size_t off = 0;
    for (int i = 0; i < 14; ++i) {
      RELA_CHECK(src[i], RELA_EINVAL, "rela_lstmnet_load: parameter %d is NULL", i);
      RELA_HIP(hipMemcpyAsync(tmp + off, src[i], sizeof(float) * cnt[i], hipMemcpyHostToDevice, s));
      dv[i] = tmp + off;
      off += cnt[i];
    }
  }
  auto pack = [&](int mode, const float* w, const float* w2, float* frag, int CT, int KS) {
    const int64_t total = (int64_t)CT * KS * 64;
    hipLaunchKernelGGL(pack_frags, dim3(ceil_div(total, 256)), dim3(256), 0, s, mode, w, w2, A, frag, CT, KS);
  };
  hipLaunchKernelGGL(pack_conv1_bf16x3, dim3(ceil_div(2 * 8 * 64 * 8, 256)), dim3(256), 0, s, dv[0],
                     reinterpret_cast<uint16_t*>(n->d.B1), 0);
  pack(kPackConv2, dv[2], nullptr, n->d.B2, 4, 128);
  pack(kPackConv3, dv[4], nullptr, n->d.B3, 4, 144);
  hipLaunchKernelGGL(pack_f32emu, dim3(ceil_div(f32emu::packed_u4<f32emu::ProbConv2>() * 8 / 3, 256)), dim3(256), 0, s, 1, dv[2],
                     reinterpret_cast<uint16_t*>(n->d.B2e), f32emu::ProbConv2::NCG, f32emu::ProbConv2::KS);
  hipLaunchKernelGGL(pack_f32emu, dim3(ceil_div(f32emu::packed_u4<f32emu::ProbConv3>() * 8 / 3, 256)), dim3(256), 0, s, 2, dv[4],
                     reinterpret_cast<uint16_t*>(n->d.B3e), f32emu::ProbConv3::NCG, f32emu::ProbConv3::KS);
  pack(kPackLstm, dv[6], dv[7], n->Bl, GemmLstm::CT, GemmLstm::KS);
  hipLaunchKernelGGL(pack_wih_rec64_perm, dim3(ceil_div((int64_t)2048 * 3136 / 8, 256)), dim3(256), 0, s, dv[6], n->Wrec);
  hipLaunchKernelGGL(pack_f32emu, dim3((unsigned)ceil_div(f32emu::packed_u4<f32emu::ProbGateX>() * 8 / 3, 256)), dim3(256), 0, s, 4, dv[6],
                     reinterpret_cast<uint16_t*>(n->Wx3), f32emu::ProbGateX::NCG, f32emu::ProbGateX::KS);
  pack(kPackHeads, dv[12], dv[10], n->d.Bh, 2, 128);
  hipLaunchKernelGGL(pack_conv1_i8, dim3(32), dim3(256), 0, s, dv[0], dv[1], reinterpret_cast<uint8_t*>(n->d.W1d), n->d.s1q,
                     n->d.b1q);
  hipLaunchKernelGGL(pack_frags_bf16s, dim3(ceil_div((int64_t)Conv2F::CT * Conv2F::KS * 64 * 8, 256)), dim3(256), 0, s, 1,
                     dv[2], reinterpret_cast<uint16_t*>(n->d.B2f), Conv2F::CT, Conv2F::KS);
  hipLaunchKernelGGL(pack_frags_bf16s, dim3(ceil_div((int64_t)Conv3F::CT * Conv3F::KS * 64 * 8, 256)), dim3(256), 0, s, 2,
                     dv[4], reinterpret_cast<uint16_t*>(n->d.B3f), Conv3F::CT, Conv3F::KS);
  RELA_HIP(hipMemcpyAsync(n->d.b1, dv[1], sizeof(float) * 32, hipMemcpyDeviceToDevice, s));
  RELA_HIP(hipMemcpyAsync(n->d.b2, dv[3], sizeof(float) * 64, hipMemcpyDeviceToDevice, s));
  RELA_HIP(hipMemcpyAsync(n->d.b3, dv[5], sizeof(float) * 64, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(pack_lstm_bias, dim3(8), dim3(256), 0, s, dv[8], dv[9], n->bl);
  hipLaunchKernelGGL(pack_head_bias, dim3(1), dim3(64), 0, s, dv[13], dv[11], A, n->d.bh);
  RELA_LAUNCH_CHECK();
  if (tmp) {
    RELA_HIP(hipStreamSynchronize(s));
    (void)hipFree(tmp);
  }
  n->loaded = true;
  n->version += 1;
  return RELA_OK;
}

namespace {
// conv trunk of an AtariLSTMNet: f32 kernels, or (fast && N >= kFastTrunkMinN) conv1 -> conv2 fused and conv3 on
// split-bf16 MFMA with a3 turned back into f32 in place (a1 is then NOT produced, a2 holds split records)
// records (with fast): a3 stays in split records (the rec64 operand of the learner's split-bf16 gate GEMM); returns
// whether it did
// rec (with emu, N >= kEmuConvMinN): N * (kRec2Bytes + kRec3Bytes) bytes of scratch -- the f32x3 trunk on split3 records
// (conv12_s3 -> conv3_img_s3); a3's records stay at rec + N * kRec2Bytes for the gate GEMM; keep_f32: a1 / a2 / a3 are
// ALSO written as channel-last f32 (the learner's online pass); returns whether the records were produced
bool lstm_trunk_launch(const FFNetDev& d, int N, const uint8_t* s_dev, float* a1, float* a2, float* a3, bool fast,
                       hipStream_t s, const char* const* names, bool records = false, bool emu = false, uint8_t* rec = nullptr,
                       bool keep_f32 = true) {
  if (emu && rec && N >= kEmuConvMinN && N <= kEmuMaxN) {
    uint8_t *rec2 = rec, *rec3 = rec + (int64_t)N * kRec2Bytes;
    {
      ProfScope prof(names[1], s);
      note_launch("conv12_s3");
      if (keep_f32)
        hipLaunchKernelGGL(s3::conv12_s3<true>, dim3(persistent_blocks(N)), dim3(s3::Conv12S::kT), s3::Conv12S::LDS_TOTAL, s, s_dev,
                           (const uint4*)d.W1d, (const float*)d.s1q, (const float*)d.b1q, (const uint4*)d.B2e, (const float*)d.b2,
                           rec2, a1, N);
      else
        hipLaunchKernelGGL(s3::conv12_s3<false>, dim3(persistent_blocks(N)), dim3(s3::Conv12S::kT), s3::Conv12S::LDS_TOTAL, s, s_dev,
                           (const uint4*)d.W1d, (const float*)d.s1q, (const float*)d.b1q, (const uint4*)d.B2e, (const float*)d.b2,
                           rec2, (float*)nullptr, N);
    }
    {
      ProfScope prof(names[2], s);
      note_launch("conv3_img_s3");
      s3::launch_conv3_img(rec2, d.B3e, d.b3, rec3, N, s, persistent_blocks(N));
    }
    if (keep_f32) {
      note_launch("unsplit_s3");
      const int64_t p2 = (int64_t)N * 81, p3 = (int64_t)N * 49;
      hipLaunchKernelGGL(s3::unsplit_s3<64>, dim3((unsigned)ceil_div(p2 * 16, 256)), dim3(256), 0, s, (const uint8_t*)rec2, a2, p2);
      hipLaunchKernelGGL(s3::unsplit_s3<64>, dim3((unsigned)ceil_div(p3 * 16, 256)), dim3(256), 0, s, (const uint8_t*)rec3, a3, p3);
    }
    return true;
  }
  if (fast && N >= kFastTrunkMinN) {
    uint8_t *r2 = reinterpret_cast<uint8_t*>(a2), *r3 = reinterpret_cast<uint8_t*>(a3);
    {
      ProfScope prof(names[1], s);
      note_launch("conv12_i8"); hipLaunchKernelGGL(conv12_i8, dim3(persistent_blocks(N)), dim3(kThreads), Conv12I::LDS_TOTAL, s, s_dev,
                         (const uint4*)d.W1d, (const float*)d.s1q, (const float*)d.b1q, (const uint4*)d.B2f, (const float*)d.b2, r2, N);
    }
    ProfScope prof(names[2], s);
    note_launch("conv_bf16s<Conv3F>"); hipLaunchKernelGGL(conv_bf16s<Conv3F>, dim3(persistent_blocks(ceil_div(N, Conv3F::S))), dim3(kThreads),
                       Conv3F::LDS_TOTAL, s, (const uint8_t*)r2, (const uint4*)d.B3f, (const float*)d.b3, r3, N);
    if (records) return true;
    note_launch("unsplit_records64"); hipLaunchKernelGGL(unsplit_records64, dim3(ceil_div((int64_t)N * 49, 4)), dim3(256), 0, s, r3, (int64_t)N * 49);
    return false;
  }
  {
    ProfScope prof(names[0], s);
    note_launch("conv1_bf16x3"); hipLaunchKernelGGL(conv1_bf16x3, dim3(ceil_div(N, Conv1B::S)), dim3(kThreads), Conv1B::LDS_BYTES, s, s_dev, d.B1,
                       d.b1, a1, N);
  }
  // (f32x3 without record scratch, or below its batch threshold: the exact f32 kernels -- same accuracy)
  {
    ProfScope prof(names[1], s);
    launch_conv<Conv2>(a1, d.B2, d.b2, a2, N, s);
  }
  {
    ProfScope prof(names[2], s);
    launch_conv<Conv3>(a2, d.B3, d.b3, a3, N, s);
  }
  return false;
}
const char* const kLstmActorNames[3] = {"conv1_bf16x3", "conv2_mfma", "conv3_mfma"};
}  // namespace

extern "C" int rela_lstmnet_step(const rela_lstmnet* n, int N, const uint8_t* s_dev, const float* legal_dev,
                                 const float* h_in, const float* c_in, float* h_out, float* c_out, float* q_dev,
                                 float* adv_dev, void* ws, int64_t ws_bytes, void* stream_) {
  RELA_CHECK(n && n->loaded, RELA_ESTATE, "rela_lstmnet_step: parameters were never loaded");
  RELA_CHECK(N >= 1 && s_dev && legal_dev && h_in && c_in && h_out && c_out && ws, RELA_EINVAL,
             "rela_lstmnet_step: bad arguments");
  RELA_CHECK(h_in != h_out && c_in != c_out, RELA_EINVAL, "rela_lstmnet_step: state may not be updated in place");
  RELA_CHECK(ws_bytes >= rela_lstmnet_workspace_bytes(n, N), RELA_EINVAL,
             "rela_lstmnet_step: workspace of %lld bytes is too small for batch %d", (long long)ws_bytes, N);
  RELA_CHECK(((uintptr_t)s_dev & 15) == 0 && ((uintptr_t)ws & 15) == 0 && ((uintptr_t)h_in & 15) == 0, RELA_EINVAL,
             "rela_lstmnet_step: s_dev, h_in and workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream_;
  float* a1 = static_cast<float*>(ws);
  float* a2 = a1 + kA1 * N;
  float* a3 = a2 + kA2 * N;
  float* ha = a3 + kA3 * N;
  float* gx = ha + kHA * N;
  const FFNetDev& d = n->d;
  // bf16x2 mode from kFastMinN rows up: a3 stays in split records, the x part of the gates is one split-bf16 GEMM
  // (gemm_bf16s.h: 41 GFLOP at 3,200 rows) and the f32 MFMA kernel only adds h x W_hh (K = 512) and runs the cell
  const bool fast_gates = n->precision == 1 && N >= kFastMinN;
  uint8_t* rec = static_cast<uint8_t*>(ws) + lstm_ws_records_offset(N);
  const bool recs = lstm_trunk_launch(d, N, s_dev, a1, a2, a3, n->precision == 1, s, kLstmActorNames, fast_gates,
                                      n->precision == 2, rec, /*keep_f32=*/false);
  if (n->precision == 2 && recs) {
    // f32x3: the x part of the gates (3136 -> 2048) as a three-part GEMM over a3's records -> gx (raw sums, permuted gate
    // columns); the f32 MFMA kernel adds h x W_hh (K = 512) and the bias and runs the cell
    {
      ProfScope prof("lstm_gates_x_f32x3", s);
      note_launch("gemm_s3<gates_x>");
      s3::launch<ProbGateX3, s3::kEpiRaw>(rec + (int64_t)N * kRec2Bytes, n->Wx3, nullptr, gx, N, s);
    }
    ProfScope prof("lstm_gates_mfma", s);
    note_launch("gemm_mfma<GemmLstmH> (f32)");
    if (prefer_bm112(N, GemmLstmH::CT / GemmLstmH::CTB, GemmLstmH::BM))
      hipLaunchKernelGGL(gemm_mfma<GemmLstmH112>, dim3(GemmLstmH112::CT / GemmLstmH112::CTB, ceil_div(N, GemmLstmH112::BM)),
                         dim3(kThreads), 0, s, h_in, (const float*)gx, (const float*)n->Bl, (const float*)n->bl, h_out,
                         c_in, c_out, N);
    else
      hipLaunchKernelGGL(gemm_mfma<GemmLstmH>, dim3(GemmLstmH::CT / GemmLstmH::CTB, ceil_div(N, GemmLstmH::BM)),
                         dim3(kThreads), 0, s, h_in, (const float*)gx, (const float*)n->Bl, (const float*)n->bl, h_out,
                         c_in, c_out, N);
  } else if (fast_gates && recs) {
    int rc = gemm16::launch_rec64_nt(reinterpret_cast<const uint8_t*>(a3), n->Wrec, N, 2048, 49, gemm16::EpiPlain{gx, 2048}, s,
                                     "lstm_gates_x_bf16");
    if (rc != RELA_OK) return rc;
    ProfScope prof("lstm_gates_mfma", s);
    note_launch("gemm_mfma<GemmLstmH> (f32)");
    if (prefer_bm112(N, GemmLstmH::CT / GemmLstmH::CTB, GemmLstmH::BM))
      hipLaunchKernelGGL(gemm_mfma<GemmLstmH112>, dim3(GemmLstmH112::CT / GemmLstmH112::CTB, ceil_div(N, GemmLstmH112::BM)),
                         dim3(kThreads), 0, s, h_in, (const float*)gx, (const float*)n->Bl, (const float*)n->bl, h_out,
                         c_in, c_out, N);
    else
      hipLaunchKernelGGL(gemm_mfma<GemmLstmH>, dim3(GemmLstmH::CT / GemmLstmH::CTB, ceil_div(N, GemmLstmH::BM)),
                         dim3(kThreads), 0, s, h_in, (const float*)gx, (const float*)n->Bl, (const float*)n->bl, h_out,
                         c_in, c_out, N);
  } else {
    ProfScope prof("lstm_gates_mfma", s);
    note_launch("gemm_mfma<GemmLstm> (f32)");
    if (prefer_bm112(N, GemmLstm::CT / GemmLstm::CTB, GemmLstm::BM))
      hipLaunchKernelGGL(gemm_mfma<GemmLstm112>, dim3(GemmLstm112::CT / GemmLstm112::CTB, ceil_div(N, GemmLstm112::BM)),
                         dim3(kThreads), 0, s, (const float*)a3, h_in, (const float*)n->Bl, (const float*)n->bl, h_out,
                         c_in, c_out, N);
    else
      hipLaunchKernelGGL(gemm_mfma<GemmLstm>, dim3(GemmLstm::CT / GemmLstm::CTB, ceil_div(N, GemmLstm::BM)),
                         dim3(kThreads), 0, s, (const float*)a3, h_in, (const float*)n->Bl, (const float*)n->bl, h_out,
                         c_in, c_out, N);
  }
  if (q_dev || adv_dev) {
    {
      ProfScope prof("heads_mfma", s);
      hipLaunchKernelGGL(gemm_mfma<GemmHeads>, dim3(GemmHeads::CT / GemmHeads::CTB, ceil_div(N, GemmHeads::BM)),
                         dim3(kThreads), 0, s, (const float*)h_out, (const float*)nullptr, (const float*)d.Bh,
                         (const float*)d.bh, ha, (const float*)nullptr, (float*)nullptr, N);
    }
    {
      ProfScope prof("dueling", s);
      hipLaunchKernelGGL(dueling_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, s, (const float*)ha, legal_dev, q_dev,
                         adv_dev, N, n->num_action);
    }
  }
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

// ---- internal entry points for the R2D2 learner (csrc/learner_r2d2.hip) -------------------------
// conv trunk only: frames u8[N][4][84][84] -> a1 / a2 / a3 (channel-last, ffnet_layout.h)
namespace rela_amd {
int lstmnet_trunk(const rela_lstmnet* n, int N, const uint8_t* s_dev, float* a1, float* a2, float* a3, hipStream_t s,
                  const char* const* names, bool fast, bool* a3_records, uint8_t* s3_scratch, bool keep_f32) {
  RELA_CHECK(n && n->loaded, RELA_ESTATE, "lstmnet_trunk: parameters were never loaded");
  RELA_CHECK(N >= 1 && s_dev && a1 && a2 && a3, RELA_EINVAL, "lstmnet_trunk: bad arguments");
  const bool rec = lstm_trunk_launch(n->d, N, s_dev, a1, a2, a3, fast, s, names, a3_records != nullptr,
                                     !fast && n->precision == 2, s3_scratch, keep_f32);
  if (a3_records) *a3_records = rec;
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

int64_t gate_x3_packed_bytes() { return (int64_t)sizeof(uint4) * f32emu::packed_u4<f32emu::ProbGateX>(); }
int pack_gate_x3(const float* w_ih_dev, void* dst, bool permuted, hipStream_t s) {
  RELA_CHECK(w_ih_dev && dst, RELA_EINVAL, "pack_gate_x3: bad arguments");
  hipLaunchKernelGGL(pack_f32emu, dim3((unsigned)ceil_div(f32emu::packed_u4<f32emu::ProbGateX>() * 8 / 3, 256)), dim3(256), 0, s,
                     permuted ? 4 : 3, w_ih_dev, reinterpret_cast<uint16_t*>(dst), f32emu::ProbGateX::NCG, f32emu::ProbGateX::KS);
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}
int gate_x3_gemm(const uint8_t* a3_records, const void* packed, const float* bias, float* gx, int M, hipStream_t s) {
  RELA_CHECK(a3_records && packed && bias && gx && M >= 1, RELA_EINVAL, "gate_x3_gemm: bad arguments");
  note_launch("gemm_s3<gates_x>");
  s3::launch<ProbGateX3, s3::kEpiBias>(a3_records, reinterpret_cast<const uint4*>(packed), bias, gx, M, s);
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

// The ONLINE net's conv trunk of the R2D2 learner on split-bf16 MFMA (r3): conv1 -> conv2 fused, with conv1's records
// copied out for the rows [a1_lo, N) -- the training frames, whose a1 the backward kernels read -- and conv3; a1 (those
// rows), a2 and a3 come out as split RECORDS in the f32 tensors' places.  The caller feeds a3's records to the gate GEMM
// and then turns the training rows back into f32 with trunk_unsplit_rows.
int lstmnet_trunk_records(const rela_lstmnet* n, int N, const uint8_t* s_dev, float* a1, float* a2, float* a3, int a1_lo,
                          hipStream_t s, const char* const* names) {
  RELA_CHECK(n && n->loaded, RELA_ESTATE, "lstmnet_trunk_records: parameters were never loaded");
  RELA_CHECK(N >= 1 && s_dev && a1 && a2 && a3 && a1_lo >= 0 && a1_lo <= N, RELA_EINVAL, "lstmnet_trunk_records: bad arguments");
  const FFNetDev& d = n->d;
  TrunkJobs jobs{};
  jobs.n = 1;
  TrunkJob& j0 = jobs.j[0];
  j0.in0 = s_dev, j0.in1 = s_dev, j0.n_in0 = N;
  j0.B2 = (const uint4*)d.B2f, j0.B3 = (const uint4*)d.B3f;
  j0.b2 = d.b2, j0.b3 = d.b3;
  j0.W1d = d.W1d, j0.s1q = d.s1q, j0.b1q = d.b1q;
  j0.a1_out = reinterpret_cast<uint8_t*>(a1), j0.a1_lo = a1_lo, j0.n_a1 = N - a1_lo;
  j0.a2 = reinterpret_cast<uint8_t*>(a2), j0.a3 = reinterpret_cast<uint8_t*>(a3);
  j0.N = N, j0.block0 = 0, j0.nblocks = std::min(kNumCU, N);
  {
    ProfScope prof(names[1], s);
    note_launch("conv12_i8_jobs");
    hipLaunchKernelGGL(conv12_i8_jobs, dim3(j0.nblocks), dim3(kThreads), Conv12I::LDS_TOTAL, s, jobs);
  }
  {
    ProfScope prof(names[2], s);
    note_launch("conv3_bf16s_jobs");
    hipLaunchKernelGGL(conv3_bf16s_jobs, dim3(j0.nblocks), dim3(kThreads), Conv3F::LDS_TOTAL, s, jobs);
  }
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

// dueling heads on rows of LSTM outputs: o f32[N][512] -> ha [N][32] (cols 0..A-1 = fc_a, col 31 = fc_v) and q [N][A]
int lstmnet_heads(const rela_lstmnet* n, int N, const float* o, const float* legal, float* ha, float* q, hipStream_t s,
                  const char* name) {
  RELA_CHECK(n && n->loaded, RELA_ESTATE, "lstmnet_heads: parameters were never loaded");
  const FFNetDev& d = n->d;
  ProfScope prof(name, s);
  hipLaunchKernelGGL(gemm_mfma<GemmHeads>, dim3(GemmHeads::CT / GemmHeads::CTB, ceil_div(N, GemmHeads::BM)),
                     dim3(kThreads), 0, s, o, (const float*)nullptr, (const float*)d.Bh, (const float*)d.bh, ha,
                     (const float*)nullptr, (float*)nullptr, N);
  hipLaunchKernelGGL(dueling_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, s, (const float*)ha, legal, q,
                     (float*)nullptr, N, n->num_action);
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}
}  // namespace rela_amd
