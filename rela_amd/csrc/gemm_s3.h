// gemm_s3.h -- the f32x3 contractions of gemm_f32emu.h over PRE-SPLIT activations ("split3 records").
//
//   out[m][n] = relu(bias[n] + sum_k X(m, k) * W[k][n])        X, W, bias, out: f32 values
//
// Same arithmetic as gemm_f32emu.h (every f32 operand as three exact bf16 parts, the six products with i + j <= 2 on
// v_mfma_f32_16x16x32_bf16, f32 accumulation), but the split of an activation happens ONCE, in the epilogue of the
// kernel that produces it, instead of in every consumer's k-loop (conv2 reads each a1 value four times, conv3 each a2
// value nine times, and the split is 13 vector instructions per pair of values: two thirds of the k-loop's VALU work
// in gemm_f32emu.h, in a loop whose pace the VALU, not the matrix core, sets).
//
// split3 record of a pixel with C channels (C = 32 for a1, 64 for a2 / a3): 6 C bytes,
//       [ part 0: C x bf16 | part 1: C x bf16 | part 2: C x bf16 ],     x = p0 + p1 + p2 exactly
// (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1), round to nearest even: for a normal f32 the third residual
// has at most 8 significant bits, so the three parts carry the f32 value EXACTLY -- unsplit_s3 gives it back bit for
// bit, which is how the learner's backward pass and the f32 kernels below the batch thresholds read such a tensor).
// A lane's MFMA fragment of part p for k-group g is the 16 bytes at record + p * 2C + (k-step's channel half) * 64 + 16 g:
// no conversion, no VALU instruction between the load and the matrix core.
//
// This file: the record format, split / unsplit, the address maps of the contractions over records, and the GEMM
// kernel (fc 3136 -> 512, the recurrent net's input projection 3136 -> 2048).  The convolutions over records are
// conv12_s3.h (conv1 -> conv2 fused per frame) and conv_img_s3.h (conv3 from LDS images).
// History (r5, measured): the first form kept gemm_f32emu.h's loop (fragments straight from global
// memory into registers) minus the split -- slower in proportion to its 6 instead of 4 bytes per value; the second moved
// the fragments through LDS-DMA in 64-column blocks -- no faster (L2 delivery); the third is below.  Merging the five
// small products into the main accumulator was tried in the first form: 7.7e-8 mean error per layer against 2.3e-8.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "gemm_f32emu.h"

namespace rela_amd {
namespace s3 {

using f32emu::bf16x2;
using f32emu::bf16x8;
using f32emu::f32x2;
using f32emu::f32x4;
using f32emu::kMaxBlocks;
using f32emu::kStageU4;
using f32emu::kT;
using f32emu::TN;
using f32emu::u32x4;
using f32emu::WRegs;

// ---- the contractions over split3 records: byte offsets ----
// row_base(m): byte offset of im2col row m's patch origin (record of its first pixel); koff(ks): byte offset of k-step
// ks inside the patch for part 0, k-group 0; PS: byte distance between the parts of a record.
struct ProbConv2 {  // a1 records [N][20][20] x 192 B -> a2 [N][9][9][64]
  static constexpr int KS = 16, NCG = 1, OC = 64, PS = 64, REC = 192;
  __device__ static uint32_t row_base(int m) {
    const int n = m / 81, pos = m - n * 81;
    const int oy = pos / 9, ox = pos - oy * 9;
    return (uint32_t)(((n * 20 + 2 * oy) * 20 + 2 * ox) * REC);
  }
  __device__ static uint32_t koff(int ks) {
    const int tp = f32emu::ProbConv2::tap(ks);
    return (uint32_t)(((tp >> 2) * 20 + (tp & 3)) * REC);
  }
};
struct ProbConv3 {  // a2 records [N][9][9] x 384 B -> a3 [N][7][7][64]
  static constexpr int KS = 18, NCG = 1, OC = 64, PS = 128, REC = 384;
  __device__ static uint32_t row_base(int m) {
    const int n = m / 49, pos = m - n * 49;
    const int oy = pos / 7, ox = pos - oy * 7;
    return (uint32_t)(((n * 9 + oy) * 9 + ox) * REC);
  }
  __device__ static uint32_t koff(int ks) {
    const int tp = f32emu::ProbConv3::tap(ks);
    return (uint32_t)(((tp / 3) * 9 + tp % 3) * REC + (ks & 1) * 64);
  }
  __device__ static uint32_t pair_off(int kp) { return koff(2 * kp); }
};
template <int OC_>
struct ProbFcT {  // a3 records [N][49] x 384 B (k = pos * 64 + c) -> h [N][OC]
  static constexpr int KS = 98, NCG = OC_ / 64, OC = OC_, PS = 128, REC = 384;
  __device__ static uint32_t row_base(int m) { return (uint32_t)m * (49u * REC); }
  __device__ static uint32_t koff(int ks) { return (uint32_t)((ks >> 1) * REC + (ks & 1) * 64); }
  __device__ static uint32_t pair_off(int kp) { return (uint32_t)(kp * REC); }
};
using ProbFc = ProbFcT<512>;

// four values (consecutive channels of one pixel) -> their three bf16 parts, two packed words per part
__device__ __forceinline__ void split3_4(const f32x4& v, uint2& p0, uint2& p1, uint2& p2) {
  const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
  const bf16x2 a0 = __builtin_convertvector(a, bf16x2), b0 = __builtin_convertvector(b, bf16x2);
  const f32x2 ra = a - __builtin_convertvector(a0, f32x2), rb = b - __builtin_convertvector(b0, f32x2);
  const bf16x2 a1 = __builtin_convertvector(ra, bf16x2), b1 = __builtin_convertvector(rb, bf16x2);
  const f32x2 sa = ra - __builtin_convertvector(a1, f32x2), sb = rb - __builtin_convertvector(b1, f32x2);
  const bf16x2 a2 = __builtin_convertvector(sa, bf16x2), b2 = __builtin_convertvector(sb, bf16x2);
  p0 = make_uint2(__builtin_bit_cast(uint32_t, a0), __builtin_bit_cast(uint32_t, b0));
  p1 = make_uint2(__builtin_bit_cast(uint32_t, a1), __builtin_bit_cast(uint32_t, b1));
  p2 = make_uint2(__builtin_bit_cast(uint32_t, a2), __builtin_bit_cast(uint32_t, b2));
}

// ---------------------------------------------------------------------------------------------------------------------
// The GEMM kernel (fc 3136 -> 512, the recurrent net's input projection 3136 -> 2048).  What r4's gemm_f32emu.h measured
// as "activation fetches cost 100 us of 263" is the texture addresser: a fragment-shaped load (16 rows x 64 B per wave
// instruction) keeps it busy ~38 cycles where a whole-line load of the same 1 KB takes ~16
// (profiles/r04_f32emu_probe_counters.txt: TA_BUSY 45 % with 6,064 loads per CU), and the pre-split form of that kernel
// (same loop, 20 vector instructions left per 144 MFMAs) ran SLOWER in proportion to its 6 instead of 4 bytes per
// value.  So here
//   * ACTIVATIONS reach LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no staging instructions) in WHOLE
//     128-byte lines: one wave instruction fetches one part (both channel halves = the two k-steps of a "pair") of 8
//     consecutive rows.  The LDS image of such a 1 KB block is lane-linear, so the bank spreading is done on the SOURCE
//     side: lane l fetches (row r = (l & 15) >> 1, 16-byte unit u = 2 (l >> 4) + (l & 1)), which puts unit u of row r at
//     slot 16 (u >> 1) + 2 r + (u & 1) -- the 16 lanes a ds_read_b128 pass serves (rows 0-7 at unit u0, rows 0-7 at
//     u0 + 1) then hit the 16 bank groups once each (SQ_LDS_BANK_CONFLICT = 0);
//   * a block owns a range of 16-row tiles and 128 COLUMNS; it walks the range in passes of up to TMV tiles; the four
//     waves own 32 columns each and ALL tiles of the pass: an activation fragment is fetched from global memory once
//     per block and read from LDS by four waves.  (r5, first form: 64 columns and <= 6 tiles per block moved 1.6 GB
//     through L2 for fc at 6,400 rows and spent more issue slots on its 15 LDS-DMAs per pair -- each with its M0 and
//     address arithmetic -- than on its 52 MFMAs: 167 us.  Bytes per MFMA go with 1 / rows + 1 / columns.)
//   * WEIGHTS: a wave's own six fragments per k-step arrive by LDS-DMA too (fragment order = lane order: 1 KB
//     contiguous reads), into a private two-slot ring -- no register holds data in flight, so nothing the compiler does
//     to registers can touch an outstanding load;
//   * two activation buffers: pair P + 1 is issued when pair P starts; ONE raw s_barrier per pair (it publishes pair
//     P's lines and retires the reads of the buffer pair P + 1 lands in); counted vmcnt, never 0 in the loop; the
//     pipeline runs on across pass boundaries (next pass's rows, same weights);
//   * LDS reads are inline asm (the compiler would drain vmcnt before any LDS read it can see while an LDS-DMA is
//     pending) with counted lgkmcnt: a tile's three fragments are read while the tile before issues its 12 MFMAs.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TMV = 7;                       // row tiles per pass (7 x 32 accumulator registers per wave: 8 spilled)
constexpr int CT = 2;                        // column tiles per wave (32 columns; 128 per block)
constexpr int kABuf = TMV * 6144;            // one pair of k-steps: [tile][part][rows 0-7 | 8-15] x 1 KB
constexpr int kBSlot = CT * 3072;            // one k-step of one wave's weights: [column tile][part] x 1 KB
constexpr int kLdsB = 2 * kABuf;             // the four waves' weight rings (2 slots each) start here
constexpr int kLdsSpare = kLdsB + 4 * 2 * kBSlot;  // 1 KB landing zone of the padding loads
constexpr int kLdsTotal = kLdsSpare + 1024;
static_assert(kLdsTotal <= 160 * 1024, "LDS budget");
__host__ __device__ constexpr int glds_per_wave(int nt) { return (6 * nt + 3) / 4; }

typedef __attribute__((address_space(3))) uint8_t* lds_ptr_t;
typedef const __attribute__((address_space(1))) uint8_t* gbl_ptr_t;
__device__ __forceinline__ void glds16(const uint8_t* g, uint8_t* l) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}
// probe builds only (tools/ubench/s3_probe.hip): S3_GEMM_ABLATE 1 = no activation loads in the loop, 2 = no weight loads
// (the counted waits then return at once: timing only, the results are garbage)
#ifndef S3_GEMM_ABLATE
#define S3_GEMM_ABLATE 0
#endif
#ifndef S3_GEMM_AHEAD
#define S3_GEMM_AHEAD 1  // activation fragments read this many tiles ahead of the MFMAs (1: 125-132 us, 2: 150-160 us for fc
#endif                   // at 6,400 rows on the boxes of r5 -- the deeper ring costs more in registers than the LDS round trip)
constexpr int kAheadTiles = S3_GEMM_AHEAD;
template <int OFF>
__device__ __forceinline__ u32x4 lds_read128(uint32_t addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void wait_lgkm(u32x4& a, u32x4& b, u32x4& c) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int I>
using IC = std::integral_constant<int, I>;
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(IC<N - 1>{});
  }
}

struct Pipe {            // what the k-loop carries from pass to pass
  uint32_t g;            // global pair counter (buffer = g & 1)
  uint32_t s;            // global k-step counter (weight slot = s & 1)
};

// this wave's weights of k-step k -> ring slot `slot`: [column tile j][part] x 1 KB, lane-linear
template <class P>
__device__ __forceinline__ void issue_weights(const uint8_t* wsrc, uint8_t* lds, uint32_t bring, int k, uint32_t slot, uint32_t lane) {
  const uint8_t* src = wsrc + (size_t)k * (kStageU4 * 16) + lane * 16;
  uint8_t* dst = lds + bring + slot * kBSlot;
  if constexpr (S3_GEMM_ABLATE == 2) return;
#pragma unroll
  for (int q = 0; q < CT * 3; ++q) glds16(src + q * 1024, dst + q * 1024);
}

// One pass: NT row tiles x this wave's 32 columns over all KS k-steps.  J = LDS-DMA instructions per wave and pair for
// the activations (the block's larger pass size decides it, so that the counts in flight do not change at a pass
// boundary).  offc / offn: this wave's J source offsets (row base + part + this lane's unit) for this pass and the next.
// In flight at the top of pair P (issue order): A(P) | W(2P) | W(2P+1) -- then W(2P+1)... see the waits below.
template <class P, int NT, int J>
__device__ __forceinline__ void k_pass(const uint8_t* __restrict__ Xb, const uint8_t* __restrict__ wsrc, uint8_t* lds, uint32_t lds0,
                                       const uint32_t (&offc)[J], const uint32_t (&offn)[J], const uint32_t (&ldst)[J],
                                       uint32_t frag_base, uint32_t wave, f32x4 (&acc)[NT][CT], f32x4 (&accs)[NT][CT], Pipe& pp,
                                       int kp0, int kp1) {
  constexpr int WL = 3 * CT;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t bring = kLdsB + wave * 2 * kBSlot;  // this wave's weight ring
  for (int kp = kp0; kp < kp1; ++kp) {  // (split-K: this block's slice of the pairs)
    const uint32_t abuf = (pp.g & 1) * kABuf;
    // ---- pair kp's lines have landed (issued one pair ago; younger: the weights of k-step s, issued since); everybody
    // is done with the other buffer
    wait_vm<WL>();
    __builtin_amdgcn_s_barrier();
    {  // weights of k-step s + 1 (its slot was read last at k-step s - 1), then the next pair's lines
      int k1 = 2 * kp + 1;
      issue_weights<P>(wsrc, lds, bring, k1, (pp.s + 1) & 1, lane);
      int k2 = kp + 1;
      const bool nextp = k2 >= kp1;  // (uniform)
      if (nextp) k2 = kp0;
      const uint8_t* src = Xb + P::pair_off(k2);
      const uint32_t dst = ((pp.g + 1) & 1) * kABuf;
      if constexpr (S3_GEMM_ABLATE != 1) {
#pragma unroll
        for (int j = 0; j < J; ++j)
          glds16(src + (nextp ? offn[j] : offc[j]), lds + (ldst[j] >= (uint32_t)kLdsSpare ? ldst[j] : dst + ldst[j]));
      }
    }
    static_for<2>([&](auto hh) {
      constexpr int H = decltype(hh)::value;
      if constexpr (H == 1) {  // weights of k-step s + 1 = the next pair's first (slot read last at k-step s - 1)
        int k2 = 2 * kp + 2;
        if (k2 >= 2 * kp1) k2 = 2 * kp0;
        issue_weights<P>(wsrc, lds, bring, k2, (pp.s + 1) & 1, lane);
      }
      wait_vm<J + WL>();  // the weights of k-step s (younger: the next pair's lines and one k-step of weights)
      const uint32_t wa = lds0 + bring + (pp.s & 1) * kBSlot + lane * 16;
      u32x4 wq[CT][3];
      static_for<CT>([&](auto jj) {
        constexpr int JJ = decltype(jj)::value;
        wq[JJ][0] = lds_read128<JJ * 3072>(wa);
        wq[JJ][1] = lds_read128<JJ * 3072 + 1024>(wa);
        wq[JJ][2] = lds_read128<JJ * 3072 + 2048>(wa);
      });
      const uint32_t xa = lds0 + abuf + frag_base + H * 512;
      // activation fragments: a ring of kAheadTiles + 1 tiles -- tile T + kAheadTiles is read while tile T issues its MFMAs
      constexpr int RING = kAheadTiles + 1, PRE = NT < kAheadTiles ? NT : kAheadTiles;
      u32x4 x[RING][3];
      static_for<PRE>([&](auto tt) {
        constexpr int T = decltype(tt)::value;
        x[T][0] = lds_read128<T * 6144>(xa);
        x[T][1] = lds_read128<T * 6144 + 2048>(xa);
        x[T][2] = lds_read128<T * 6144 + 4096>(xa);
      });
      constexpr int AHEAD = PRE * 3;  // activation reads issued behind the weights' so far
      static_for<CT>([&](auto jj) {
        constexpr int JJ = decltype(jj)::value;
        wait_lgkm<AHEAD + 3 * (CT - 1 - JJ)>(wq[JJ][0], wq[JJ][1], wq[JJ][2]);
      });
      bf16x8 wf[CT][3];
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) wf[j][q] = __builtin_bit_cast(bf16x8, wq[j][q]);
      static_for<NT>([&](auto tt) {
        constexpr int T = decltype(tt)::value;
        constexpr int C = T % RING;
        if constexpr (T + kAheadTiles < NT) {
          constexpr int N2 = (T + kAheadTiles) % RING;
          x[N2][0] = lds_read128<(T + kAheadTiles) * 6144>(xa);
          x[N2][1] = lds_read128<(T + kAheadTiles) * 6144 + 2048>(xa);
          x[N2][2] = lds_read128<(T + kAheadTiles) * 6144 + 4096>(xa);
        }
        // younger reads still allowed in flight: those of the tiles after T that were issued
        constexpr int YOUNGER = 3 * ((T + kAheadTiles < NT ? T + kAheadTiles : NT - 1) - T);
        wait_lgkm<YOUNGER>(x[C][0], x[C][1], x[C][2]);
        const bf16x8 x0 = __builtin_bit_cast(bf16x8, x[C][0]), x1 = __builtin_bit_cast(bf16x8, x[C][1]),
                     x2 = __builtin_bit_cast(bf16x8, x[C][2]);
        // the five small products have an accumulator of their own (added once per pass): the main one takes ONE
        // rounding per 32 k at the magnitude of the running sum
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          accs[T][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][2], x0, accs[T][j], 0, 0, 0);
          accs[T][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], x2, accs[T][j], 0, 0, 0);
          accs[T][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], x1, accs[T][j], 0, 0, 0);
          accs[T][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], x0, accs[T][j], 0, 0, 0);
          accs[T][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], x1, accs[T][j], 0, 0, 0);
          acc[T][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], x0, acc[T][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      pp.s += 1;
    });
    pp.g += 1;
  }
}

// X: split3 records; Wp: pack_f32emu_at's fragment-ordered bf16 triples [cg64][ks][u][part][lane] x 8;
// EPI 0: out = relu(bias + sum) as f32 [M][OC] | 1: the same as split3 records (6 * OC bytes per row) | 2: the raw sums,
// f32 [gridDim.y][M][OC] -- gridDim.y > 1 splits the contraction (small batches: 512 rows x 512 columns are 128 blocks of
// one tile; fc_reduce adds the slices, the bias and the ReLU).
// gridDim.x = 8 * NCG * (row blocks / 8) with NCG = OC / 128: consecutive block ids go to the 8 XCDs in turn, so the
// column groups of one row block (they read the same activations) share an L2.
enum { kEpiRelu = 0, kEpiReluS3 = 1, kEpiRaw = 2, kEpiBias = 3 };  // 3: bias + sum, no ReLU, f32 [M][OC]
template <class P, int EPI>
__global__ __launch_bounds__(kT, 1) void gemm_s3(const uint8_t* __restrict__ Xb, const uint4* __restrict__ Wp,
                                                 const float* __restrict__ bias, void* __restrict__ out_, int M) {
  constexpr int NCG = P::OC / (64 * CT);
  constexpr bool OUT_S3 = EPI == kEpiReluS3;
  const int kp0 = (int)((P::KS / 2) * blockIdx.y / gridDim.y), kp1 = (int)((P::KS / 2) * (blockIdx.y + 1) / gridDim.y);
  static_assert(P::OC % (64 * CT) == 0, "a block owns 128 columns");
  __shared__ __attribute__((aligned(1024))) uint8_t lds[kLdsTotal];
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)lds;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (a scalar: LDS-DMA destinations are wave-uniform)
  const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int cg = bi % NCG;
  const int rb = (bi / NCG) * 8 + xcd;
  const int nrb = gridDim.x / NCG;
  const int rt_total = (M + 15) >> 4;
  const int r0 = (int)((int64_t)rt_total * rb / nrb), r1 = (int)((int64_t)rt_total * (rb + 1) / nrb);
  const int cnt = r1 - r0;
  if (cnt <= 0) return;  // (block-uniform)
  const int passes = (cnt + TMV - 1) / TMV;
  const int base = cnt / passes, rem = cnt - base * passes;  // `rem` passes of base + 1 tiles, then passes of base
  const int ntmax = base + (rem ? 1 : 0);
  auto first_of = [&](int p) { return p < rem ? (base + 1) * p : (base + 1) * rem + base * (p - rem); };
  auto size_of = [&](int p) { return p < rem ? base + 1 : base; };
  // this wave's columns: 128 cg + 32 wave .. + 31 = column tiles u = 2 (wave & 1), + 1 of the 64-column pack group
  const int cg64 = cg * 2 + (wave >> 1), u0 = 2 * (wave & 1);
  const uint8_t* wsrc = reinterpret_cast<const uint8_t*>(Wp) + ((size_t)cg64 * P::KS * kStageU4 + (size_t)u0 * 3 * 64) * 16;
  const int col0 = cg * 128 + 32 * wave + 4 * g;  // this lane's first channel (column tile j: + 16 j)
  // fragment read address of (tile 0, part 0, half 0): rows 8-15 are the second 1 KB block of a (tile, part)
  const uint32_t frag_base = (uint32_t)((li >> 3) * 1024 + (16 * (g >> 1) + 2 * (li & 7) + (g & 1)) * 16);
  // LDS-DMA source: this lane's row inside an 8-row block and its 16-byte unit inside the 128-byte line
  const int lr = (lane & 15) >> 1, lu = 2 * (lane >> 4) + (lane & 1);

  f32x4 bv[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    if constexpr (EPI == kEpiRaw)
      bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    else
      bv[j] = *reinterpret_cast<const f32x4*>(bias + col0 + 16 * j);
    asm volatile("" : "+v"(bv[j]));  // (returned before the first LDS-DMA is issued: see conv_img_s3.h)
  }
  Pipe pp{0u, 0u};

  auto run = [&](auto ntm) {
    constexpr int NTM = decltype(ntm)::value, J = glds_per_wave(NTM);  // the block's passes have NTM or NTM - 1 tiles
    uint32_t offc[J], offn[J], ldst[J];
    // LDS-DMA instruction i = wave + 4 j of a pair: i = (tile * 3 + part) * 2 + row half; an instruction past the pass's
    // tiles fetches instruction 0's line again (into the spare KB, or into a tile slot this pass does not read)
    auto set_off = [&](uint32_t (&off)[J], int p) {
      const int nt = size_of(p), t0 = r0 + first_of(p);
#pragma unroll
      for (int j = 0; j < J; ++j) {
        int i = wave + 4 * j;
        if (i >= 6 * nt) i = 0;
        const int part = (i >> 1) % 3, t = i / 6, b = i & 1;
        const int row = min((t0 + t) * 16 + b * 8 + lr, M - 1);
        off[j] = P::row_base(row) + (uint32_t)(part * 128 + lu * 16);
      }
    };
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int i = wave + 4 * j;
      ldst[j] = i < 6 * ntmax ? (uint32_t)i * 1024u : (uint32_t)kLdsSpare;
    }
    set_off(offc, 0);
    set_off(offn, min(1, passes - 1));
    // prologue: pair 0's lines, then the weights of k-step 0 (what the loop's first counted wait expects in flight)
    {
      const uint8_t* src = Xb + P::pair_off(kp0);
#pragma unroll
      for (int j = 0; j < J; ++j) glds16(src + offc[j], lds + ldst[j]);
      issue_weights<P>(wsrc, lds, kLdsB + wave * 2 * kBSlot, 2 * kp0, 0, lane);
    }
    // one pass of NT tiles: accumulators (NT x 2 column tiles x {main, small terms} x 4 registers -- sized by NT, not by
    // TMV: with all eight tiles' registers live the kernel spilled 1.4 KB per lane and wrote 100 MB of scratch per launch),
    // the k-loop, the epilogue
    auto one_pass = [&](auto ntag, int p) {
      constexpr int NT = decltype(ntag)::value;
      f32x4 acc[NT][CT], accs[NT][CT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[t][j] = bv[j], accs[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      k_pass<P, NT, J>(Xb, wsrc, lds, lds0, offc, offn, ldst, frag_base, wave, acc, accs, pp, kp0, kp1);
      // ---- epilogue: this lane's four channels (per column tile) of pixel li of every tile
      const int t0 = r0 + first_of(p);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int row = (t0 + t) * 16 + li;
        if (row < M) {
#pragma unroll
          for (int j = 0; j < CT; ++j) {
            f32x4 v = acc[t][j] + accs[t][j];
            if constexpr (EPI == kEpiRelu || EPI == kEpiReluS3) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
            }
            if constexpr (OUT_S3) {
              uint8_t* o = reinterpret_cast<uint8_t*>(out_) + (size_t)row * (6 * P::OC) + (col0 + 16 * j) * 2;
              uint2 p0, p1, p2;
              split3_4(v, p0, p1, p2);
              *reinterpret_cast<uint2*>(o) = p0;
              *reinterpret_cast<uint2*>(o + 2 * P::OC) = p1;
              *reinterpret_cast<uint2*>(o + 4 * P::OC) = p2;
            } else {
              *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out_) + ((size_t)blockIdx.y * M + row) * P::OC + col0 + 16 * j) = v;
            }
          }
        }
      }
    };
    for (int p = 0; p < passes; ++p) {
      if (size_of(p) == NTM)
        one_pass(IC<NTM>{}, p);
      else if constexpr (NTM > 1)
        one_pass(IC<NTM - 1>{}, p);
#pragma unroll
      for (int j = 0; j < J; ++j) offc[j] = offn[j];
      set_off(offn, min(p + 2, passes - 1));
    }
  };
  switch (ntmax) {
    case 1: run(IC<1>{}); break;
    case 2: run(IC<2>{}); break;
    case 3: run(IC<3>{}); break;
    case 4: run(IC<4>{}); break;
    case 5: run(IC<5>{}); break;
    case 6: run(IC<6>{}); break;
    default: run(IC<7>{}); break;
  }
  wait_vm<0>();  // no LDS-DMA may outlive the workgroup's LDS allocation
}

// rows M = samples (fc).  Byte offsets are 32-bit: the caller keeps M * record bytes below 2^32.
// How a launch covers M rows: `nrb` row blocks (a multiple of 8: the XCD map) x NCG column groups x `slices` of the
// contraction.  Large M: one slice, as many row blocks as the chip has CUs for.  Small M (the learner's 512 rows, a
// cohort's first batches): every row block streams ALL weights of its columns, so few row blocks of ~4 tiles each and the
// contraction split over up to 8 slices instead (512 rows x 512 columns: 8 x 4 x 8 = 256 blocks, 77 MB of weights
// through L2 where 32 x 4 x 2 took 307 MB); fc_reduce adds the slices.
struct Plan {
  int nrb, slices;
};
template <class P>
inline Plan plan(int M, bool may_split) {
  constexpr int NCG = P::OC / (64 * CT);
  const int rt_total = (M + 15) / 16;
  const int full = kMaxBlocks / 2;  // one block per CU
  Plan pl;
  pl.nrb = (std::min(full / NCG, rt_total) + 7) / 8 * 8;
  pl.slices = 1;
  if (may_split && pl.nrb * NCG * 4 <= full * 3) {  // fewer than 3/4 of the CUs would get a block
    pl.nrb = ((rt_total + 3) / 4 + 7) / 8 * 8;
    pl.slices = std::max(1, std::min(8, full / (pl.nrb * NCG)));
  }
  return pl;
}
// slices > 1 (EPI = kEpiRaw only): out = f32 [slices][M][OC] raw sums
template <class P, int EPI>
inline void launch(const void* X, const uint4* Wp, const float* bias, void* out, int M, hipStream_t s, Plan pl) {
  constexpr int NCG = P::OC / (64 * CT);
  hipLaunchKernelGGL((gemm_s3<P, EPI>), dim3(pl.nrb * NCG, EPI == kEpiRaw ? pl.slices : 1), dim3(kT), 0, s,
                     reinterpret_cast<const uint8_t*>(X), Wp, bias, out, M);
}
template <class P, int EPI>
inline void launch(const void* X, const uint4* Wp, const float* bias, void* out, int M, hipStream_t s) {
  launch<P, EPI>(X, Wp, bias, out, M, s, plan<P>(M, false));
}

// f32 channel-last [pixels][C] <-> split3 records (one thread per four channels)
template <int C>
__global__ void split_s3(const float* __restrict__ x, uint8_t* __restrict__ rec, int64_t pixels) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pixels * (C / 4)) return;
  const int64_t px = i / (C / 4);
  const int c4 = (int)(i - px * (C / 4));
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + px * C + c4 * 4);
  uint2 p0, p1, p2;
  split3_4(v, p0, p1, p2);
  uint8_t* o = rec + px * (6 * C) + c4 * 8;
  *reinterpret_cast<uint2*>(o) = p0;
  *reinterpret_cast<uint2*>(o + 2 * C) = p1;
  *reinterpret_cast<uint2*>(o + 4 * C) = p2;
}
template <int C>
__global__ void unsplit_s3(const uint8_t* __restrict__ rec, float* __restrict__ x, int64_t pixels) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pixels * (C / 4)) return;
  const int64_t px = i / (C / 4);
  const int c4 = (int)(i - px * (C / 4));
  const uint8_t* r = rec + px * (6 * C) + c4 * 8;
  const uint2 p0 = *reinterpret_cast<const uint2*>(r), p1 = *reinterpret_cast<const uint2*>(r + 2 * C),
              p2 = *reinterpret_cast<const uint2*>(r + 4 * C);
  auto f = [](uint32_t w, int hi) { return __builtin_bit_cast(float, hi ? (w & 0xffff0000u) : (w << 16)); };
  f32x4 v;
  // (p0 + p1) + p2: p0 + p1 is exact (24 significant bits at most: the parts do not overlap), then one exact add
  v[0] = (f(p0.x, 0) + f(p1.x, 0)) + f(p2.x, 0);
  v[1] = (f(p0.x, 1) + f(p1.x, 1)) + f(p2.x, 1);
  v[2] = (f(p0.y, 0) + f(p1.y, 0)) + f(p2.y, 0);
  v[3] = (f(p0.y, 1) + f(p1.y, 1)) + f(p2.y, 1);
  *reinterpret_cast<f32x4*>(x + px * C + c4 * 4) = v;
}

}  // namespace s3
}  // namespace rela_amd
