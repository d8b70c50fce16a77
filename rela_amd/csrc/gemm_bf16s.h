// gemm_bf16s.h -- f32-accurate GEMM on the bf16 matrix cores for the learners' large GEMMs (R2D2: the LSTM's
// input-side gate GEMM and its two gradients, [T*B, 3136] x [3136, 2048]: 100 GFLOP each).
//
//   C[m][n] = sum_k A[m][k] * B[n][k]          ("NT": both operands k-contiguous)
//
// Every operand is held as hi + lo bf16 (x = hi + lo to 2^-17 relative) and every product is three bf16 MFMAs
// (hi*hi + hi*lo + lo*hi, f32 accumulation) on v_mfma_f32_16x16x32_bf16: 16x the rate of v_mfma_f32_16x16x4_f32 for
// 3x the instructions, results within ~2e-6 relative of the f32 GEMM (the fast mode's stated tolerance).
// Operands arrive as "rec64" arrays -- per row, per chunk of 64 k: 64 hi (128 B) followed by 64 lo (128 B), the same
// 4 bytes per element as f32 and the same record the split-bf16 convolutions write (ffnet.hip) -- produced by the
// split kernels below (row-wise, or transposing for the weight gradient, whose k is the row index of both tensors).
//
// Kernel: 256 x 128 block tile, 8 waves of 64 x 64 (16 accumulator tiles), k-stages of 32 double-buffered in LDS
// (row = 64 B hi | 64 B lo | 32 B pad: 10 sixteen-byte units, conflict-free for the ds_read_b128 lane groups,
// tools/lds_conflicts.py), the global loads of the next TWO stages in two register sets during the MFMAs, ONE barrier
// per stage.  PMC at R2D2's shapes (tools/pmc_r2d2_learner.sh): MFMA pipe 36 % busy, LDS 21 %, no bank conflicts --
// what is left are the read burst after and the store burst before every barrier, on all eight waves at once.
// Blocks are mapped XCD-aware: block b runs on XCD b % 8, and within an XCD consecutive blocks walk the column
// blocks of one row block, so the co-resident blocks of an L2 share their A rows and all of B's current k-range.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "common.h"
#include "prof.h"

namespace rela_amd {
namespace gemm16 {
namespace {  // (included by both learners' translation units)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 512;                   // 8 waves
constexpr int BM = 256, BN = 128;         // block tile
constexpr int ROW = 160;                  // LDS bytes per tile row and stage
constexpr int A_BYTES = BM * ROW, B_BYTES = BN * ROW, STAGE = A_BYTES + B_BYTES;
constexpr int LDS_TOTAL = 2 * STAGE;      // 122,880 B
constexpr int REC = 256;                  // bytes per rec64 chunk

// hi / lo halves of 8 consecutive f32 -> two uint4 of packed bf16 (RNE both times)
__device__ __forceinline__ void split8(const float4 a, const float4 b, uint4& hi, uint4& lo) {
  const f32x2 v0 = {a.x, a.y}, v1 = {a.z, a.w}, v2 = {b.x, b.y}, v3 = {b.z, b.w};
  const bf16x2 h0 = __builtin_convertvector(v0, bf16x2), h1 = __builtin_convertvector(v1, bf16x2);
  const bf16x2 h2 = __builtin_convertvector(v2, bf16x2), h3 = __builtin_convertvector(v3, bf16x2);
  const bf16x2 l0 = __builtin_convertvector(v0 - __builtin_convertvector(h0, f32x2), bf16x2);
  const bf16x2 l1 = __builtin_convertvector(v1 - __builtin_convertvector(h1, f32x2), bf16x2);
  const bf16x2 l2 = __builtin_convertvector(v2 - __builtin_convertvector(h2, f32x2), bf16x2);
  const bf16x2 l3 = __builtin_convertvector(v3 - __builtin_convertvector(h3, f32x2), bf16x2);
  hi = make_uint4(__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1), __builtin_bit_cast(uint32_t, h2),
                  __builtin_bit_cast(uint32_t, h3));
  lo = make_uint4(__builtin_bit_cast(uint32_t, l0), __builtin_bit_cast(uint32_t, l1), __builtin_bit_cast(uint32_t, l2),
                  __builtin_bit_cast(uint32_t, l3));
}

// rows of an f32 matrix src[R][K] (K a multiple of 64, row stride K) -> rec64 dst[R][K / 64]
__global__ void split_rows_rec64(const float* __restrict__ src, int64_t R, int K, uint8_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 8 elements
  const int per_row = K >> 3;
  if (i >= R * per_row) return;
  const int64_t r = i / per_row;
  const int o = (int)(i - r * per_row);  // octet within the row
  const float* p = src + r * K + (int64_t)o * 8;
  uint4 hi, lo;
  split8(*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4), hi, lo);
  uint8_t* rec = dst + (r * (K >> 6) + (o >> 3)) * REC + (o & 7) * 16;
  *reinterpret_cast<uint4*>(rec) = hi;
  *reinterpret_cast<uint4*>(rec + 128) = lo;
}

// COLUMNS of an f32 matrix src[R][C] (row stride C, C a multiple of 64) -> rec64 dst[C][ceil(R / 64)]: record
// (c', chunk) holds rows 64 chunk .. 64 chunk + 63 of column c (zeros past row R), c' = c or, with a3_order, the
// state_dict column c49 = ch * 49 + pos of the channel-last column c = pos * 64 + ch (so that a weight gradient comes
// out in weight_ih_l0's own order).  One block per 64 x 64 tile.
__global__ __launch_bounds__(256) void split_cols_rec64(const float* __restrict__ src, int64_t R, int C, int a3_order,
                                                        uint8_t* __restrict__ dst) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64;
  const int64_t r0 = (int64_t)blockIdx.y * 64;
  const int tid = threadIdx.x;
  for (int i = tid; i < 64 * 16; i += 256) {
    const int r = i >> 4, q = i & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < R) v = *reinterpret_cast<const float4*>(src + (r0 + r) * C + c0 + 4 * q);
    tile[r][4 * q] = v.x, tile[r][4 * q + 1] = v.y, tile[r][4 * q + 2] = v.z, tile[r][4 * q + 3] = v.w;
  }
  __syncthreads();
  const int64_t chunks = (R + 63) >> 6;
  for (int i = tid; i < 64 * 8; i += 256) {
    const int c = i >> 3, o = i & 7;  // column of the tile, octet of rows
    float4 a, b;
    a.x = tile[8 * o][c], a.y = tile[8 * o + 1][c], a.z = tile[8 * o + 2][c], a.w = tile[8 * o + 3][c];
    b.x = tile[8 * o + 4][c], b.y = tile[8 * o + 5][c], b.z = tile[8 * o + 6][c], b.w = tile[8 * o + 7][c];
    uint4 hi, lo;
    split8(a, b, hi, lo);
    const int cc = c0 + c;
    const int64_t row = a3_order ? (int64_t)(cc & 63) * 49 + (cc >> 6) : cc;
    uint8_t* rec = dst + (row * chunks + blockIdx.y) * REC + o * 16;
    *reinterpret_cast<uint4*>(rec) = hi;
    *reinterpret_cast<uint4*>(rec + 128) = lo;
  }
}

// ---- epilogues: called with four consecutive columns n .. n + 3 of row m (m < M, n + 3 < N) ----
struct EpiBias {  // C[m][n] = v + bias[n]
  float* out;
  const float* bias;
  int ld;
  __device__ void operator()(int m, int n, f32x4 v) const {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias + n);
    *reinterpret_cast<f32x4*>(out + (size_t)m * ld + n) = v + b;
  }
};
struct EpiReluMask {  // C[m][n] = act[m][n] > 0 ? v : 0
  float* out;
  const float* act;
  int ld;
  __device__ void operator()(int m, int n, f32x4 v) const {
    const f32x4 a = *reinterpret_cast<const f32x4*>(act + (size_t)m * ld + n);
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = a[r] > 0.f ? v[r] : 0.f;
    *reinterpret_cast<f32x4*>(out + (size_t)m * ld + n) = o;
  }
};
struct EpiPlain {  // C[m][n] = v
  float* out;
  int ld;
  __device__ void operator()(int m, int n, f32x4 v) const { *reinterpret_cast<f32x4*>(out + (size_t)m * ld + n) = v; }
};

// A: rec64 [M][KC], B: rec64 [N][KC] (N a multiple of 4), KC chunks of 64 k.  grid = 8 * NCB * ceil(NRB / 8).
template <class Epi>
__global__ __launch_bounds__(kT) void gemm_rec64_nt(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B, int M,
                                                    int N, int KC, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int NRB = (M + BM - 1) / BM, NCB = (N + BN - 1) / BN;
  const int xcd = blockIdx.x & 7, sidx = blockIdx.x >> 3;
  const int cb = sidx % NCB, rb = (sidx / NCB) * 8 + xcd;
  if (rb >= NRB) return;
  const int m0 = rb * BM, n0 = cb * BN;
  const int wm = wave >> 1, wn = wave & 1;  // 4 x 2 waves of 64 x 64

  // staging: chunk u of a tile row = 16 B; u < 4 the stage's 32 hi, u >= 4 its 32 lo (clamped rows: a tail block
  // re-reads its last row instead of branching; those accumulators are never stored).  Six named registers, no
  // arrays and no branches around the loads: either pushes the staging data into scratch memory with a wait behind
  // every load.
  // two register sets, P and Q: a stage's loads are issued TWO stages before its LDS store (one stage of MFMAs,
  // 0.64 us, did not cover an L2 miss: MFMA pipe 36 % busy with one set)
  uint4 pa_0, pa_1, pa_2, pa_3, pb_0, pb_1, qa_0, qa_1, qa_2, qa_3, qb_0, qb_1;
  const int su = tid & 7, srow = tid >> 3;  // this thread's unit and first tile row (rows srow + 64 j)
  const int gofs = (su >> 2) * 128 + (su & 3) * 16;
  const size_t rstride = (size_t)KC * REC;
  const uint8_t* pa0 = A + (size_t)min(m0 + srow, M - 1) * rstride + gofs;
  const uint8_t* pa1 = A + (size_t)min(m0 + srow + 64, M - 1) * rstride + gofs;
  const uint8_t* pa2 = A + (size_t)min(m0 + srow + 128, M - 1) * rstride + gofs;
  const uint8_t* pa3 = A + (size_t)min(m0 + srow + 192, M - 1) * rstride + gofs;
  const uint8_t* pb0 = B + (size_t)min(n0 + srow, N - 1) * rstride + gofs;
  const uint8_t* pb1 = B + (size_t)min(n0 + srow + 64, N - 1) * rstride + gofs;
  const int sofs = srow * ROW + su * 16;
#define RELA_G_LOAD(S, ST)                                                       \
  do {                                                                           \
    const int st__ = min((ST), 2 * KC - 1);                                      \
    const int o__ = (st__ >> 1) * REC + (st__ & 1) * 64;                         \
    S##a_0 = *reinterpret_cast<const uint4*>(pa0 + o__);                         \
    S##a_1 = *reinterpret_cast<const uint4*>(pa1 + o__);                         \
    S##a_2 = *reinterpret_cast<const uint4*>(pa2 + o__);                         \
    S##a_3 = *reinterpret_cast<const uint4*>(pa3 + o__);                         \
    S##b_0 = *reinterpret_cast<const uint4*>(pb0 + o__);                         \
    S##b_1 = *reinterpret_cast<const uint4*>(pb1 + o__);                         \
  } while (0)
#define RELA_S_STORE(S, BUF)                                                     \
  do {                                                                           \
    uint8_t* ta__ = smem + (BUF) * STAGE + sofs;                                 \
    *reinterpret_cast<uint4*>(ta__) = S##a_0;                                    \
    *reinterpret_cast<uint4*>(ta__ + 64 * ROW) = S##a_1;                         \
    *reinterpret_cast<uint4*>(ta__ + 128 * ROW) = S##a_2;                        \
    *reinterpret_cast<uint4*>(ta__ + 192 * ROW) = S##a_3;                        \
    *reinterpret_cast<uint4*>(ta__ + A_BYTES) = S##b_0;                          \
    *reinterpret_cast<uint4*>(ta__ + A_BYTES + 64 * ROW) = S##b_1;               \
  } while (0)

  f32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int NS = 2 * KC;  // (even)
  RELA_G_LOAD(p, 0);
  RELA_S_STORE(p, 0);
  RELA_G_LOAD(p, 1);
  RELA_G_LOAD(q, 2);
  __syncthreads();
  const int aoff = (wm * 64 + li) * ROW + g * 16, boff = A_BYTES + (wn * 64 + li) * ROW + g * 16;
  auto mma_stage = [&](int buf) {
    const uint8_t* base = smem + buf * STAGE;
    uint4 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ah[t] = *reinterpret_cast<const uint4*>(base + aoff + t * 16 * ROW);
      al[t] = *reinterpret_cast<const uint4*>(base + aoff + t * 16 * ROW + 64);
      bh[t] = *reinterpret_cast<const uint4*>(base + boff + t * 16 * ROW);
      bl[t] = *reinterpret_cast<const uint4*>(base + boff + t * 16 * ROW + 64);
    }
    // operands swapped (B as the first operand): a lane ends up with C[m = li][n = 4 g .. 4 g + 3] of every tile
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[t]), xl = __builtin_bit_cast(bf16x8, al[t]);
        const bf16x8 wh = __builtin_bit_cast(bf16x8, bh[u]), wl = __builtin_bit_cast(bf16x8, bl[u]);
        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc[t][u], 0, 0, 0);
        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc[t][u], 0, 0, 0);
        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[t][u], 0, 0, 0);
      }
  };
  // stage st is in LDS buffer st & 1 when iteration st starts; set P holds stage st + 1 and set Q stage st + 2 at even st
  for (int st = 0; st < NS; st += 2) {
    mma_stage(0);
    RELA_S_STORE(p, 1);  // (buffer 1 was last read in stage st - 1: every wave is past that barrier)
    RELA_G_LOAD(p, st + 3);
    __syncthreads();
    mma_stage(1);
    RELA_S_STORE(q, 0);
    RELA_G_LOAD(q, st + 4);
    __syncthreads();
  }
#undef RELA_G_LOAD
#undef RELA_S_STORE
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int m = m0 + wm * 64 + t * 16 + li;
    if (m >= M) continue;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = n0 + wn * 64 + u * 16 + 4 * g;
      if (n < N) epi(m, n, acc[t][u]);
    }
  }
}

// The same GEMM with a 128 x 128 block tile: 80 KB of LDS and at most 128 registers, so that TWO blocks share a CU and
// one block's MFMAs fill the other's barrier, read and store bursts (8 waves of 64 x 32, one staging register set).
constexpr int SBM = 128, S_STAGE = (SBM + BN) * ROW, S_LDS = 2 * S_STAGE;  // 81,920 B
template <class Epi>
__global__ __launch_bounds__(kT, 2) void gemm_rec64_nt_s(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                         int M, int N, int KC, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int NRB = (M + SBM - 1) / SBM, NCB = (N + BN - 1) / BN;
  const int xcd = blockIdx.x & 7, sidx = blockIdx.x >> 3;
  const int cb = sidx % NCB, rb = (sidx / NCB) * 8 + xcd;
  if (rb >= NRB) return;
  const int m0 = rb * SBM, n0 = cb * BN;
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves of 64 x 32
  uint4 ra0, ra1, rb0, rb1;
  const int su = tid & 7, srow = tid >> 3;
  const int gofs = (su >> 2) * 128 + (su & 3) * 16;
  const size_t rstride = (size_t)KC * REC;
  const uint8_t* pa0 = A + (size_t)min(m0 + srow, M - 1) * rstride + gofs;
  const uint8_t* pa1 = A + (size_t)min(m0 + srow + 64, M - 1) * rstride + gofs;
  const uint8_t* pb0 = B + (size_t)min(n0 + srow, N - 1) * rstride + gofs;
  const uint8_t* pb1 = B + (size_t)min(n0 + srow + 64, N - 1) * rstride + gofs;
  const int sofs = srow * ROW + su * 16;
  constexpr int SA = SBM * ROW;
#define RELA_SG_LOAD(ST)                                      \
  do {                                                        \
    const int st__ = min((ST), 2 * KC - 1);                   \
    const int o__ = (st__ >> 1) * REC + (st__ & 1) * 64;      \
    ra0 = *reinterpret_cast<const uint4*>(pa0 + o__);         \
    ra1 = *reinterpret_cast<const uint4*>(pa1 + o__);         \
    rb0 = *reinterpret_cast<const uint4*>(pb0 + o__);         \
    rb1 = *reinterpret_cast<const uint4*>(pb1 + o__);         \
  } while (0)
#define RELA_SS_STORE(BUF)                                    \
  do {                                                        \
    uint8_t* ta__ = smem + (BUF) * S_STAGE + sofs;            \
    *reinterpret_cast<uint4*>(ta__) = ra0;                    \
    *reinterpret_cast<uint4*>(ta__ + 64 * ROW) = ra1;         \
    *reinterpret_cast<uint4*>(ta__ + SA) = rb0;               \
    *reinterpret_cast<uint4*>(ta__ + SA + 64 * ROW) = rb1;    \
  } while (0)
  f32x4 acc[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int NS = 2 * KC;
  RELA_SG_LOAD(0);
  RELA_SS_STORE(0);
  RELA_SG_LOAD(1);
  __syncthreads();
  const int aoff = (wm * 64 + li) * ROW + g * 16, boff = SA + (wn * 32 + li) * ROW + g * 16;
  for (int st = 0; st < NS; ++st) {
    const uint8_t* base = smem + (st & 1) * S_STAGE;
    uint4 ah[4], al[4], bh[2], bl[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ah[t] = *reinterpret_cast<const uint4*>(base + aoff + t * 16 * ROW);
      al[t] = *reinterpret_cast<const uint4*>(base + aoff + t * 16 * ROW + 64);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bh[u] = *reinterpret_cast<const uint4*>(base + boff + u * 16 * ROW);
      bl[u] = *reinterpret_cast<const uint4*>(base + boff + u * 16 * ROW + 64);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[t]), xl = __builtin_bit_cast(bf16x8, al[t]);
        const bf16x8 wh = __builtin_bit_cast(bf16x8, bh[u]), wl = __builtin_bit_cast(bf16x8, bl[u]);
        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, acc[t][u], 0, 0, 0);
        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc[t][u], 0, 0, 0);
        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[t][u], 0, 0, 0);
      }
    RELA_SS_STORE((st + 1) & 1);
    RELA_SG_LOAD(st + 2);
    __syncthreads();
  }
#undef RELA_SG_LOAD
#undef RELA_SS_STORE
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int m = m0 + wm * 64 + t * 16 + li;
    if (m >= M) continue;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int n = n0 + wn * 32 + u * 16 + 4 * g;
      if (n < N) epi(m, n, acc[t][u]);
    }
  }
}

template <class Epi>
inline int launch_rec64_nt(const uint8_t* A, const uint8_t* B, int M, int N, int KC, Epi epi, hipStream_t s,
                           const char* name) {
  // (initialised once, thread-safely: launches may come from several host threads)
  static const hipError_t attr_set =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rec64_nt<Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
  RELA_HIP(attr_set);
  // the small tile when the large one would leave most CUs without a block (dW_hh: 32 blocks -> 64 on twice the CUs,
  // 0.18 -> 0.11 ms); on the large GEMMs it loses to the large tile's operand reuse (gates_x 0.55 -> 0.59 ms) although
  // two blocks per CU fill each other's barriers: they are bound by operand traffic.
  const bool small_tile = ceil_div(M, BM) * ceil_div(N, BN) < 128;
  if (small_tile) {
    // (initialised once, thread-safely: launches may come from several host threads)
    static const hipError_t attr_s =
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rec64_nt_s<Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS);
    RELA_HIP(attr_s);
    const int NRB = ceil_div(M, SBM), NCB = ceil_div(N, BN);
    ProfScope prof(name, s);
    note_launch("gemm_rec64_nt");
    hipLaunchKernelGGL(gemm_rec64_nt_s<Epi>, dim3(8 * NCB * ceil_div(NRB, 8)), dim3(kT), S_LDS, s, A, B, M, N, KC, epi);
    return RELA_OK;
  }
  const int NRB = ceil_div(M, BM), NCB = ceil_div(N, BN);
  ProfScope prof(name, s);
  note_launch("gemm_rec64_nt");
  hipLaunchKernelGGL(gemm_rec64_nt<Epi>, dim3(8 * NCB * ceil_div(NRB, 8)), dim3(kT), LDS_TOTAL, s, A, B, M, N, KC, epi);
  return RELA_OK;
}

}  // namespace
}  // namespace gemm16
}  // namespace rela_amd
