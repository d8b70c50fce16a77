// gemm_f32emu.h -- f32 contractions carried out on the bf16 matrix cores WITHOUT giving up f32 accuracy.
//
// r5: the ARITHMETIC defined here (three exact bf16 parts per operand, six products, small terms in their own
// accumulator) and the weight packing (pack_f32emu_at) are what the library runs; the KERNEL of this file (activations
// as f32, split in every consumer's k-loop) is no longer launched by it -- csrc/gemm_s3.h, conv12_s3.h and
// conv_img_s3.h work on activations split once by their producer.  The kernel stays for tools/ubench/f32emu_probe.hip,
// the A/B that showed what bounded it.
//
//   out[m][n] = relu(bias[n] + sum_k X(m, k) * W[k][n])        X, W, bias, out: f32
//
// v_mfma_f32_16x16x4_f32 peaks at 157 TFLOP/s on gfx950, v_mfma_f32_16x16x32_bf16 at 2.5 PFLOP/s (16 x).  An f32 number
// has a 24-bit significand and a bf16 number an 8-bit one, so every f32 operand is the sum of THREE bf16 numbers,
//       x = x0 + x1 + x2,   x0 = bf16(x),  x1 = bf16(x - x0),  x2 = bf16(x - x0 - x1)      (round to nearest even),
// where both subtractions are exact in f32 and what is left after x2 is below 2^-26 |x| (signed digits carry an extra
// bit each).  A product of two bf16 numbers is exact in an f32 accumulator (16 significant bits), so
//       x * w = sum_{i, j} x_i * w_j
// can be accumulated term by term in the MFMA's f32 accumulators.  NPROD = 9 keeps all nine terms; NPROD = 6 (what the
// library runs) keeps the terms with i + j <= 2 and drops x1 w2 + x2 w1 + x2 w2 <= 2 * 2^-9 * 2^-18 |x w| = 2^-26 |x w|,
// a quarter of the 2^-24 rounding an f32 multiplier puts on the same product.  The five small terms have accumulators
// of their own (added once per pass), so the running sum takes ONE rounding per 32 k where an f32 FMA chain takes 32.
// Measured (tools/ubench/f32emu_probe.hip, random ReLU-like activations, mean |error| against f64): conv2 2.3e-8 with
// six or nine products against 7.1e-8 for a sequential f32 FMA chain, conv3 2.6e-8 / 7.9e-8, fc 6.6e-8 / 1.8e-7; whole
// network (tests/test_ffnet_gpu.py::test_ffnet_f32x3_is_f32_accurate): 1.50e-8 against 1.99e-8 for the exact f32 MFMA
// kernels and 1.49e-8 for torch CPU f32.  (The "split-bf16" fast mode of gemm_bf16s.h keeps TWO parts and three
// products: 2^-16 per product.  That is a different, narrower arithmetic; this one is not.)
//
// Tiling.  The bf16 MFMA is fast enough that operand delivery, not the matrix core, is what a kernel has to organise:
//   * a wave owns a 64-pixel x 64-channel output tile (TM x TN = 4 x 4 MFMA tiles, 2 x 64 accumulator registers): per
//     k-step of 32 it needs 4 x 3 activation fragments and 4 x 3 weight fragments for 96 MFMAs of 16 cycles;
//   * ACTIVATIONS never touch LDS.  All 64 channels of a pixel belong to the same wave, so no other wave wants the same
//     im2col row: lane (pixel li, k-group g) reads the 32 contiguous bytes X(m, 32 ks + 8 g .. + 7) straight from
//     global memory (the im2col gather is index arithmetic), two k-steps ahead of their use -- also across the end of
//     a pass, from the next pass's rows -- and splits them into the three bf16 fragments in registers (9 VALU ops per
//     pair of values) while the MFMAs of the tile before issue;
//   * the k-steps of conv2 walk the taps one input-parity class at a time and those of conv3 in boustrophedon order, so
//     that the pixels a wave fetches for one k-step are mostly those of the k-step before (L2 hits: 68 % for conv2);
//   * WEIGHTS are split and laid out in fragment order once, at load time (pack_f32emu_at), and the four waves of a
//     block share each k-step's 12 KB through a double-buffered LDS stage (one barrier per k-step): a lane's fragment
//     is one conflict-free ds_read_b128;
//   * operands are swapped (weights are the MFMA's A operand), so a lane ends up with FOUR CONSECUTIVE CHANNELS of one
//     pixel: bias is the accumulators' initial value and the epilogue is ReLU + one 16-byte store per tile;
//   * one block of 4 waves per CU (OCC = 1: 256 + ~190 registers, no spills; at two blocks per CU the register budget
//     of 256 leaves one activation set and spills); a block owns a contiguous range of 16-pixel row tiles, its waves
//     a quarter each, walked in passes of 4 or 3 tiles chosen so that every block of the launch gets the same work to
//     within one tile (a grid of whole 64-pixel tiles would quantise 33,180 row tiles to 5 rounds for 4.05).
// Where the time goes (r4, N = 6,554, probe at ~1.95 GHz under load: conv2 263 us, conv3 153, fc 150 against 312 / 229
// / 203 for the exact f32 MFMA kernels): SQ_VALU_MFMA_BUSY 40 %; MFMA time is ADDITIVE to the rest (nine products cost
// +43 us = the MFMA time of the three extra ones).  Ablations: without the activation fetches 159 us, without the
// weight stage + barrier 222, without the split 255, none of the three 127.  Tried without gain: activations coalesced
// through per-wave LDS sets (277 us: the texture addresser is not the limit), every tap an L2 hit (271), one register
// set (285), the staging instructions folded into the tiles' MFMA shadows with double-buffered weight fragments (277:
// ~200 v_accvgpr moves per k-step pair -- the allocator parks operands in AGPRs and the VALU, not the matrix core,
// sets the pace); two blocks per CU with passes of 2 / 1 tiles (OCC = 2: 284 us).  Kernel time scales with 1 / clock (215 us
// inside bench.py at 2.37 GHz): cycle-bound at ~3,900 cycles per k-step for 1,536 of MFMA.  SQ counters (conv2, per wave):
// 317 VALU + 97 MFMA instructions per k-step, all VALU time co-executing with the MFMA pipe, the wave issuing 40 % of
// its cycles, 32 % in s_waitcnt (LDS: 3 %) and ~25 % in other waits -- the ISA shows an s_waitcnt vmcnt(0) at the top
// of every k-step pair, where the allocator copies registers that loads are still in flight to into AGPRs (pressure:
// 256 architectural registers).  The same loads as opaque instructions straight into AGPRs with hand-counted vmcnt
// (4 TMv + 4 / 2 TMv) measured slower (conv3 180 us; 327 v_accvgpr moves per pair).  Next: the loop in ISA-level hands.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace rela_amd {
namespace f32emu {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct WRegs {
  u32x4 r0, r1, r2;  // this thread's three 16-byte pieces of one k-step of weights, on their way to LDS
};

constexpr int kT = 256;                       // 4 wavefronts per block
constexpr int TN = 4;                         // 64 output channels per wave (and per block)
constexpr int kStageU4 = TN * 3 * 64;         // uint4 per k-step of weights: [u][part][lane]
constexpr int kMaxBlocks = 512;               // 2 per CU

// ---- the three contractions of the AtariFFNet trunk (pyrela/net.py:18-31) as implicit GEMMs over channel-last f32 ----
// row_base(m): element offset of im2col row m's patch origin; koff(ks): element offset of k-step ks inside the patch
// (k = (kh, kw, c), c fastest, 32 k per step).
struct ProbConv2 {  // a1 [N][20][20][32] -> a2 [N][9][9][64], 4x4 stride 2
  static constexpr int KS = 16, NCG = 1, OC = 64;
  __device__ static int row_base(int m) {
    const int n = m / 81, pos = m - n * 81;
    const int oy = pos / 9, ox = pos - oy * 9;
    return ((n * 20 + 2 * oy) * 20 + 2 * ox) * 32;
  }
  // k-step ks -> tap (kh, kw).  An input pixel (y, x) serves the four taps with kh = y (mod 2), kw = x (mod 2), so
  // the taps are walked one parity class at a time, and inside a class in the order (0,0) (0,2) (2,2) (2,0): the
  // pixels a wave fetches for one k-step are, up to one row or one column, those of the k-step before, and all four
  // uses of a pixel fall within four consecutive k-steps -- they hit in L2 instead of leaving the XCD again.
  __host__ __device__ static int tap(int ks) {
    const int c = ks >> 2, j = ks & 3;
    const int dh = j >> 1, dw = (j ^ (j >> 1)) & 1;
    return (((c >> 1) + 2 * dh) << 2) | ((c & 1) + 2 * dw);
  }
  __device__ static int koff(int ks) {
    const int tp = tap(ks);
    return ((tp >> 2) * 20 + (tp & 3)) * 32;
  }
};
struct ProbConv3 {  // a2 [N][9][9][64] -> a3 [N][7][7][64], 3x3 stride 1
  static constexpr int KS = 18, NCG = 1, OC = 64;
  __device__ static int row_base(int m) {
    const int n = m / 49, pos = m - n * 49;
    const int oy = pos / 7, ox = pos - oy * 7;
    return ((n * 9 + oy) * 9 + ox) * 64;
  }
  // taps in boustrophedon order (0,0) (0,1) (0,2) (1,2) (1,1) (1,0) (2,0) (2,1) (2,2): each k-step's pixels are the
  // previous one's shifted by one column or one row
  __host__ __device__ static int tap(int ks) {
    const int i = ks >> 1, kh = i / 3, r = i - 3 * kh;
    return kh * 3 + ((kh & 1) ? 2 - r : r);
  }
  __device__ static int koff(int ks) {
    const int tp = tap(ks);
    return ((tp / 3) * 9 + tp % 3) * 64 + (ks & 1) * 32;
  }
};
struct ProbGateX {  // a3 [N][3136] -> the x part of the LSTM gates [N][2048] (pack mode 4; the kernel is gemm_s3.h's)
  static constexpr int KS = 98, NCG = 32, OC = 2048;
};
struct ProbFc {  // a3 [N][3136] (k = pos * 64 + c) -> h [N][512]
  static constexpr int KS = 98, NCG = 8, OC = 512;
  __device__ static int row_base(int m) { return m * 3136; }
  __device__ static int koff(int ks) { return ks * 32; }
};

// eight f32 (this lane's k-group of one pixel) -> the three bf16 fragments
__device__ __forceinline__ void split3(const f32x4& a, const f32x4& b, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
  const f32x2 v[4] = {{a[0], a[1]}, {a[2], a[3]}, {b[0], b[1]}, {b[2], b[3]}};
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16x2 x0 = __builtin_convertvector(v[i], bf16x2);
    const f32x2 r1 = v[i] - __builtin_convertvector(x0, f32x2);
    const bf16x2 x1 = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(x1, f32x2);
    const bf16x2 x2 = __builtin_convertvector(r2, bf16x2);
    h[i] = __builtin_bit_cast(uint32_t, x0);
    m[i] = __builtin_bit_cast(uint32_t, x1);
    l[i] = __builtin_bit_cast(uint32_t, x2);
  }
  p0 = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  p1 = __builtin_bit_cast(bf16x8, make_uint4(m[0], m[1], m[2], m[3]));
  p2 = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

// The k-loop of one pass (TMv row tiles x 64 channels) inside the wave's flattened (pass, k-step, row tile) pipeline:
//   * while the 4 * NPROD MFMAs of one tile issue (16 cycles each, 4 of them issue), the VALU splits the NEXT tile's
//     activations into xf[other]; the tile after the last one of a k-step is the first of the next k-step, and the one
//     after the last k-step the first of the NEXT PASS -- the pipeline never drains between passes;
//   * activations live in two register sets (k-step parity); a tile's registers are fetched again, for two k-steps
//     later, as soon as they have been split (two k-steps = 2 x 96 MFMAs ahead of their use), from the next pass's
//     rows (xoffn) once the k-step index runs past this pass;
//   * the weight fragments of a k-step are read from LDS once, right behind the barrier that published them.
// Invariant at the start of k-step ks (S = ks & 1): xa/xb[S][1..] hold k-step ks, [S][0] is in flight for ks + 2,
// [S ^ 1][*] hold ks + 1, xf[(S * TMv) & 1] the split tile 0 of ks, wf the fragments of ks, wreg the weights of ks + 1.
template <class P, int NPROD, int TMv, int TMX>
__device__ __forceinline__ void k_loop(const uint8_t* __restrict__ Xb, const uint4* __restrict__ wsrc, uint4* wl,
                                       const uint32_t (&xoff)[TMX], const uint32_t (&xoffn)[TMX], f32x4 (&xa)[2][TMX],
                                       f32x4 (&xb)[2][TMX], bf16x8 (&wf)[TN][3], bf16x8 (&xf)[2][3], f32x4 (&acc)[TMX][TN], f32x4 (&accs)[TMX][TN],
                                       WRegs& wreg, int& step0) {
  const int tid = threadIdx.x, lane = tid & 63;
  auto reload = [&](auto set, int t, int k) {
    constexpr int S = decltype(set)::value;
    const bool nextp = k >= P::KS;  // (wave-uniform)
    const uint32_t ko = (uint32_t)P::koff(nextp ? k - P::KS : k) * 4u;
    const f32x4* p = reinterpret_cast<const f32x4*>(Xb + ((nextp ? xoffn[t] : xoff[t]) + ko));
    xa[S][t] = p[0];
    xb[S][t] = p[1];
  };
  constexpr int kI[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};  // (weight part, activation part), smallest terms first
  constexpr int kJ[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
  auto step = [&](int ks, auto set, auto other) {
    constexpr int S = decltype(set)::value;
    const int sidx = step0 + ks;
#pragma unroll
    for (int t = 0; t < TMv; ++t) {
      const int cur = (S * TMv + t) & 1, nxt = cur ^ 1;
      if (t + 1 < TMv) {
        split3(xa[S][t + 1], xb[S][t + 1], xf[nxt][0], xf[nxt][1], xf[nxt][2]);
        reload(set, t + 1, ks + 2);
      } else {
        split3(xa[S ^ 1][0], xb[S ^ 1][0], xf[nxt][0], xf[nxt][1], xf[nxt][2]);  // first tile of the next k-step
        reload(other, 0, ks + 3);
      }
      // The NPROD - 1 small terms (each <= 2^-8 of the product) have accumulators of their own, added to the main ones
      // once per pass: their roundings happen at their own magnitude, and the main accumulator takes ONE rounding per
      // 32 k (the x0 w0 MFMA) where an f32 FMA chain takes 32 -- the result is closer to the exact sum than that chain's.
#pragma unroll
      for (int q = 9 - NPROD; q < 8; ++q)
#pragma unroll
        for (int u = 0; u < TN; ++u)
          accs[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][kI[q]], xf[cur][kJ[q]], accs[t][u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < TN; ++u)
        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][0], xf[cur][0], acc[t][u], 0, 0, 0);
    }
    // weights of k-step sidx + 1 (in registers since the step before) -> the other stage; then fetch k-step sidx + 2
    u32x4* nxts = reinterpret_cast<u32x4*>(wl + ((sidx + 1) & 1) * kStageU4) + tid;
    nxts[0] = wreg.r0;
    nxts[kT] = wreg.r1;
    nxts[2 * kT] = wreg.r2;
    {
      int k2 = ks + 2;
      if (k2 >= P::KS) k2 -= P::KS;  // (the next pass walks the same weights; past the last pass the fetch is unused)
      const u32x4* src = reinterpret_cast<const u32x4*>(wsrc + (size_t)k2 * kStageU4) + tid;
      wreg.r0 = src[0];
      wreg.r1 = src[kT];
      wreg.r2 = src[2 * kT];
    }
    __syncthreads();
    const uint4* stage = wl + ((sidx + 1) & 1) * kStageU4 + lane;
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int p = 0; p < 3; ++p) wf[u][p] = __builtin_bit_cast(bf16x8, stage[(u * 3 + p) * 64]);
  };
  static_assert(P::KS % 2 == 0 && P::KS >= 4, "k-steps are walked in pairs");
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  for (int ks = 0; ks < P::KS; ks += 2) {
    step(ks, I0{}, I1{});
    step(ks + 1, I1{}, I0{});
  }
  step0 += P::KS;
}

// Wp: [cg][ks][u][part][lane] x 8 bf16 (pack_f32emu_at).  gridDim.x = 8 * NCG * (row blocks / 8): consecutive block ids
// go to the 8 XCDs in turn, so the column groups of one row block (they read the same activations) share an L2.
template <class P, int NPROD, int OCC>
__global__ __launch_bounds__(kT, OCC) void gemm_f32emu(const float* __restrict__ X, const uint4* __restrict__ Wp,
                                                     const float* __restrict__ bias, float* __restrict__ out, int M) {
  __shared__ __attribute__((aligned(16))) uint4 wl[2 * kStageU4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int cg = bi % P::NCG;
  const int rb = (bi / P::NCG) * 8 + xcd;
  const int nrb = gridDim.x / P::NCG;
  const int rt_total = (M + 15) >> 4;
  const int r0 = (int)((int64_t)rt_total * rb / nrb), r1 = (int)((int64_t)rt_total * (rb + 1) / nrb);
  const int cnt = r1 - r0;
  if (cnt <= 0) return;  // (block-uniform)
  const int w0 = r0 + cnt * wave / 4, w1 = r0 + cnt * (wave + 1) / 4;
  const int myc = w1 - w0;
  // tiles per pass: TMX or TMX - 1.  One block per CU (OCC = 1, 512 registers per wave): 4 / 3; two blocks per CU
  // (256 registers): 2 / 1 -- twice the weight-stage traffic per MFMA, but a second wave per SIMD to fill the stalls
  constexpr int TMX = OCC == 1 ? 4 : 2;
  const int maxc = (cnt + 3) >> 2;                              // the largest share among the four waves
  const int passes = (maxc + TMX - 1) / TMX;
  const int n4 = max(maxc - (TMX - 1) * passes, 0);             // passes of TMX tiles (the rest take TMX - 1): capacity >= maxc
  // pass p of this wave: its first tile (clamped into the wave's range: a wave one tile short of the largest share
  // recomputes its last tile and stores nothing) and how many of the pass's tiles are its own
  auto first_of = [&](int p) { return p < n4 ? TMX * p : TMX * n4 + (TMX - 1) * (p - n4); };
  const uint8_t* Xb = reinterpret_cast<const uint8_t*>(X);
  const uint4* wsrc = Wp + (size_t)cg * P::KS * kStageU4;
  uint32_t xoff[TMX], xoffn[TMX];  // byte offsets of the tiles' patch origins, this pass and the next (operand < 4 GB)
  auto set_xoff = [&](uint32_t (&xo)[TMX], int p) {
    const int t0 = min(w0 + first_of(p), max(w1 - 1, w0));
#pragma unroll
    for (int t = 0; t < TMX; ++t) {
      const int row = min((t0 + t) * 16 + li, M - 1);
      xo[t] = (uint32_t)(P::row_base(row) + g * 8) * 4u;
    }
  };
  set_xoff(xoff, 0);
  set_xoff(xoffn, min(1, passes - 1));

  // weights of k-step 0 -> stage 0, k-step 1 -> registers
  WRegs wreg;
#pragma unroll
  for (int j = 0; j < 3; ++j) wl[tid + j * kT] = wsrc[tid + j * kT];
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(wsrc + kStageU4) + tid;
    wreg.r0 = src[0];
    wreg.r1 = src[kT];
    wreg.r2 = src[2 * kT];
  }
  f32x4 xa[2][TMX], xb[2][TMX];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int t = 0; t < TMX; ++t) {
      const f32x4* p = reinterpret_cast<const f32x4*>(Xb + (xoff[t] + (uint32_t)P::koff(s2) * 4u));
      xa[s2][t] = p[0];
      xb[s2][t] = p[1];
    }
  __syncthreads();
  bf16x8 wf[TN][3], xf[2][3];
#pragma unroll
  for (int u = 0; u < TN; ++u)
#pragma unroll
    for (int p = 0; p < 3; ++p) wf[u][p] = __builtin_bit_cast(bf16x8, wl[(u * 3 + p) * 64 + lane]);
  split3(xa[0][0], xb[0][0], xf[0][0], xf[0][1], xf[0][2]);
  {
    const f32x4* p = reinterpret_cast<const f32x4*>(Xb + (xoff[0] + (uint32_t)P::koff(2) * 4u));
    xa[0][0] = p[0];
    xb[0][0] = p[1];
  }
  f32x4 acc[TMX][TN];   // x0 w0 terms; start at the bias: this lane's channels are 16 u + 4 g .. + 3 of the column group
  f32x4 accs[TMX][TN];  // the small terms
  auto acc_init = [&]() {
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + cg * 64 + 16 * u + 4 * g);
#pragma unroll
      for (int t = 0; t < TMX; ++t) acc[t][u] = bv, accs[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  acc_init();

  int step0 = 0;
  for (int p = 0; p < passes; ++p) {
    const bool four = p < n4;  // (block-uniform)
    if (four)
      k_loop<P, NPROD, TMX, TMX>(Xb, wsrc, wl, xoff, xoffn, xa, xb, wf, xf, acc, accs, wreg, step0);
    else
      k_loop<P, NPROD, TMX - 1, TMX>(Xb, wsrc, wl, xoff, xoffn, xa, xb, wf, xf, acc, accs, wreg, step0);
    const int first = first_of(p);
    const int t0 = min(w0 + first, max(w1 - 1, w0));
    const int nvalid = min(four ? TMX : TMX - 1, myc - first);
#pragma unroll
    for (int t = 0; t < TMX; ++t) {
      const int row = (t0 + t) * 16 + li;
      if (t < nvalid && row < M) {
        float* o = out + (size_t)row * P::OC + cg * 64 + 4 * g;
#pragma unroll
        for (int u = 0; u < TN; ++u) {
          f32x4 v = acc[t][u] + accs[t][u];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
          *reinterpret_cast<f32x4*>(o + 16 * u) = v;
        }
      }
    }
    acc_init();
#pragma unroll
    for (int t = 0; t < TMX; ++t) xoff[t] = xoffn[t];
    set_xoff(xoffn, min(p + 2, passes - 1));
  }
}

// f32 weight (state_dict layouts of pyrela/net.py:18-31) -> fragment-ordered bf16 triples.  mode: 1 conv2 (64,32,4,4) |
// 2 conv3 (64,64,3,3) | 3 fc (OC,3136; k = pos * 64 + c <- torch's c * 49 + pos) | 4 the LSTM's input weights with
// permuted gate columns.  One thread per (cg, ks, u, lane, j).
__device__ __forceinline__ uint16_t bf16_rne_bits(float v) {
  const uint32_t x = __builtin_bit_cast(uint32_t, v);
  return (uint16_t)((x + 0x7fffu + ((x >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_bits_f32(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
__device__ __forceinline__ void pack_f32emu_at(int64_t idx, int mode, const float* __restrict__ w, uint16_t* __restrict__ frag,
                                               int NCG, int KS) {
  const int64_t total = (int64_t)NCG * KS * TN * 64 * 8;
  if (idx >= total) return;
  const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
  const int u = (int)((idx >> 9) & 3);
  const int ks = (int)((idx >> 11) % KS), cg = (int)((idx >> 11) / KS);
  const int k = ks * 32 + (lane >> 4) * 8 + j;
  const int oc = cg * 64 + u * 16 + (lane & 15);
  float v;
  if (mode == 1) {
    const int c = k & 31, tap = ProbConv2::tap(k >> 5);
    v = w[((oc * 32 + c) * 4 + (tap >> 2)) * 4 + (tap & 3)];
  } else if (mode == 2) {
    const int c = k & 63, tap = ProbConv3::tap(k >> 5);
    v = w[((oc * 64 + c) * 3 + tap / 3) * 3 + tap % 3];
  } else if (mode == 3) {
    const int c = k & 63, pos = k >> 6;
    v = w[(size_t)oc * 3136 + c * 49 + pos];
  } else {  // 4: weight_ih_l0 (2048, 3136) with the gate columns permuted to 4 * unit + gate (ffnet.hip: GemmLstmH)
    const int c = k & 63, pos = k >> 6;
    v = w[(size_t)((oc & 3) * 512 + (oc >> 2)) * 3136 + c * 49 + pos];
  }
  const uint16_t p0 = bf16_rne_bits(v);
  const float r1 = v - bf16_bits_f32(p0);
  const uint16_t p1 = bf16_rne_bits(r1);
  const float r2 = r1 - bf16_bits_f32(p1);
  const uint16_t p2 = bf16_rne_bits(r2);
  const size_t base = ((((size_t)cg * KS + ks) * TN + u) * 3) * 64 * 8 + (size_t)lane * 8 + j;
  frag[base] = p0;
  frag[base + 64 * 8] = p1;
  frag[base + 2 * 64 * 8] = p2;
}

template <class P>
constexpr int64_t packed_u4() { return (int64_t)P::NCG * P::KS * kStageU4; }

// rows M = pixels (conv) or samples (fc).  Element offsets are 32-bit: the caller keeps M * row size below 2^31.
template <class P, int NPROD, int OCC = 2>
inline void launch(const float* X, const uint4* Wp, const float* bias, float* out, int M, hipStream_t s) {
  const int rt_total = (M + 15) / 16;
  int nrb = std::min(kMaxBlocks / 2 * OCC / P::NCG, (rt_total + 3) / 4);  // at least one tile per wave
  nrb = (nrb + 7) / 8 * 8;
  hipLaunchKernelGGL((gemm_f32emu<P, NPROD, OCC>), dim3(nrb * P::NCG), dim3(kT), 0, s, X, Wp, bias, out, M);
}

}  // namespace f32emu
}  // namespace rela_amd
