// conv12_s3.h -- conv1 -> conv2 of the AtariFFNet trunk (pyrela/net.py:20-25) FUSED per frame through LDS in the f32x3
// arithmetic: conv1's output a1 (400 pixels x 32 channels) never reaches HBM (r4: conv1_bf16x3 wrote 328 MB per 6,400
// frames and gemm_f32emu<conv2> read 395 MB back).
//
//   conv1  u8 frames x 24-bit fixed-point weights on the INT8 matrix cores, exactly as conv12_i8 (ffnet.hip): the pixels
//          x - 128 as int8, every weight as three balanced base-256 digits relative to its channel's largest, three
//          exact i32 sums per output, one f32 scale + bias (pack_conv1_i8).  Whole frame in LDS as [plane][4x4 cell].
//   a1     ReLU, split into the three bf16 parts (split3 record, 192 B per pixel) in conv1's epilogue, stored into the
//          LDS image T2: pixel (y, x) at 16-byte unit y * RQ + x * Q with Q = 13 (12 + 1 pad), RQ = 261, so that the 16
//          consecutive OUTPUT pixels of a conv2 tile (stride 2) advance by 10 units mod 16: conflict-free ds_read_b128.
//   conv2  six products per operand pair on v_mfma_f32_16x16x32_bf16 (gemm_f32emu.h's arithmetic, small terms in their own
//          accumulator); a wave owns 16 output channels, its 16 k-steps x 3 parts of weights are RESIDENT (192 registers);
//          tile by tile (16 pixels), fragments three k-steps ahead in a register ring.
//   a2     ReLU, split, staged in LDS as records and copied out whole (coalesced 16-byte stores) under the next frame's
//          conv1.
// One block of FOUR waves per CU (512 registers each: 240 hold weights), persistent over the frames b, b + grid, ...;
// two barriers per frame.  A1OUT (the learner's online(obs) pass): a1 is also written to HBM as f32 channel-last, what
// the backward kernels read.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_s3.h"

namespace rela_amd {
namespace s3 {

typedef int i32x4 __attribute__((ext_vector_type(4)));
// probe builds only (tools/ubench/s3_probe.hip): leave one part of the kernel out to see what it costs
#ifndef C12_ABLATE
#define C12_ABLATE 0
#endif

constexpr bool kNoCopyOut = C12_ABLATE == 1, kNoStage = C12_ABLATE == 2, kNoEpi1 = C12_ABLATE == 3, kNoEpi2 = C12_ABLATE == 4,
               kNoConv1 = C12_ABLATE == 5, kNoConv2 = C12_ABLATE == 6;

struct Conv12S {
  static constexpr int kT = 256;
  static constexpr int GW = 21, NPIX = GW * GW, PLANE_ELEMS = 84 * 84, IN_ELEMS = 4 * PLANE_ELEMS;
  static constexpr int PLANE1 = 7168;                 // 441 cells x 16 B, padded to a multiple of 256 B
  static constexpr int T1_BYTES = 4 * PLANE1;         // the frame as int8 cells
  static constexpr int Q = 13, RQ = 261;              // a1 image: pixel / row stride in 16-byte units
  static constexpr int T2_BYTES = 20 * RQ * 16;       // 83,520
  static constexpr int OROW = 400;                    // a2 record (384 B) + 16: the 16 pixels of a store spread over the banks
  static constexpr int O_BYTES = 81 * OROW;
  static constexpr int SPARE = T1_BYTES + T2_BYTES + O_BYTES;  // 256 B: rows past the last pixel land here
  static constexpr int LDS_TOTAL = SPARE + 256;
  static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
  static constexpr int CELLS = 4 * NPIX, IT = (CELLS + kT - 1) / kT;  // 7 cells per thread
  static constexpr int OV16 = 81 * 24, OIT = (OV16 + kT - 1) / kT;    // 16-byte chunks of an output tile: 8 per thread
  static constexpr int KS2 = 16;
  // k-step -> tap in pack_f32emu_at's mode-1 order (f32emu::ProbConv2::tap), as byte offset inside the a1 image
  static constexpr int tap2(int ks) {
    const int c = ks >> 2, j = ks & 3;
    const int dh = j >> 1, dw = (j ^ (j >> 1)) & 1;
    return (((c >> 1) + 2 * dh) << 2) | ((c & 1) + 2 * dw);
  }
  static constexpr int koff2(int ks) { return ((tap2(ks) >> 2) * RQ + (tap2(ks) & 3) * Q) * 16; }
};

// W1d / scale1 / bias1q: pack_conv1_i8 ([digit][ct 2][tap 4][lane] x 16 int8); B2: pack_f32emu_at mode 1
// ([ks][u][part][lane] x 8 bf16); out: a2 records [N][81] x 384 B; A1OUT: a1 also as f32 [N][400][32] (a1_out).
template <bool A1OUT>
__global__ __launch_bounds__(256, 1) void conv12_s3(const uint8_t* __restrict__ in, const uint4* __restrict__ W1d,
                                                    const float* __restrict__ scale1, const float* __restrict__ bias1q,
                                                    const uint4* __restrict__ B2, const float* __restrict__ bias2,
                                                    uint8_t* __restrict__ out, float* __restrict__ a1_out, int N) {
  using F = Conv12S;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem_c12[];
  uint8_t* t1 = smem_c12;
  uint8_t* t2 = t1 + F::T1_BYTES;
  uint8_t* otile = t2 + F::T2_BYTES;
  uint8_t* spare = smem_c12 + F::SPARE;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x, nblk = gridDim.x;
  int n = bid;
  if (n >= N) return;

  // ---- residents: conv2's weights (this wave's 16 channels), conv1's digits (this wave's column tile)
  const int ct1 = wave & 1, rg1 = wave >> 1;
  bf16x8 w2[F::KS2][3];
  {
    const uint4* bp = B2 + (size_t)wave * 3 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < F::KS2; ++ks)
#pragma unroll
      for (int p = 0; p < 3; ++p) w2[ks][p] = __builtin_bit_cast(bf16x8, bp[(size_t)(ks * TN * 3 + p) * 64]);
  }
  i32x4 wd[4][3];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int d = 0; d < 3; ++d) wd[ks][d] = __builtin_bit_cast(i32x4, W1d[((d * 2 + ct1) * 4 + ks) * 64 + lane]);
  const int ch1 = ct1 * 16 + 4 * g, ch2 = wave * 16 + 4 * g;
  const f32x4 sc1 = *reinterpret_cast<const f32x4*>(scale1 + ch1), bv1 = *reinterpret_cast<const f32x4*>(bias1q + ch1);
  const f32x4 bv2 = *reinterpret_cast<const f32x4*>(bias2 + ch2);

  // ---- staging of a frame's cells (clamped cell index; a cell = rows 4Y .. 4Y + 3 x 4 bytes of one plane)
  uint32_t st[F::IT][4];
  int cgoff[F::IT], cloff[F::IT];  // (frame-invariant: this thread's cells in the frame and in T1)
#pragma unroll
  for (int j = 0; j < F::IT; ++j) {
    const int c = min(tid + j * F::kT, F::CELLS - 1);
    const int pl = c / F::NPIX, P = c - pl * F::NPIX;
    const int Y = P / F::GW, X = P - Y * F::GW;
    cgoff[j] = pl * F::PLANE_ELEMS + 4 * Y * 84 + 4 * X;
    cloff[j] = pl * F::PLANE1 + P * 16;
  }
  auto g_load1 = [&](int fr, int j) __attribute__((always_inline)) {
    const uint8_t* src = in + (size_t)fr * F::IN_ELEMS + cgoff[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) st[j][r] = *reinterpret_cast<const uint32_t*>(src + r * 84);
  };
  auto s_store = [&](int j) __attribute__((always_inline)) {  // x -> x - 128 as int8: flip the sign bits
    *reinterpret_cast<uint4*>(t1 + cloff[j]) = make_uint4(st[j][0] ^ 0x80808080u, st[j][1] ^ 0x80808080u, st[j][2] ^ 0x80808080u,
                                                          st[j][3] ^ 0x80808080u);
  };

  // ---- the output tile of the previous frame -> HBM, in slices behind conv1's MFMAs
  int prev = -1;
  static_assert(F::OIT == 8, "eight named chunk registers (an array indexed inside the hooks stays in scratch)");
  uint4 oc0, oc1, oc2, oc3, oc4, oc5, oc6, oc7;
  auto oc_at = [&](auto jt) -> uint4& {
    constexpr int j = decltype(jt)::value;
    if constexpr (j == 0) return oc0;
    else if constexpr (j == 1) return oc1;
    else if constexpr (j == 2) return oc2;
    else if constexpr (j == 3) return oc3;
    else if constexpr (j == 4) return oc4;
    else if constexpr (j == 5) return oc5;
    else if constexpr (j == 6) return oc6;
    else return oc7;
  };
  auto o_read = [&](auto jt) __attribute__((always_inline)) {
    constexpr int j = decltype(jt)::value;
    const int i = min(tid + j * F::kT, F::OV16 - 1);
    const int px = i / 24, u = i - px * 24;
    oc_at(jt) = *reinterpret_cast<const uint4*>(otile + px * F::OROW + u * 16);
  };
  auto o_write = [&](auto jt) __attribute__((always_inline)) {
    constexpr int j = decltype(jt)::value;
    const int i = min(tid + j * F::kT, F::OV16 - 1);
    // (first frame: O holds nothing yet -- the bytes go to frame n's own rows, which its real tile overwrites later from
    // the same thread; no branch in conv1's instruction stream)
    reinterpret_cast<uint4*>(out + (size_t)(prev >= 0 ? prev : n) * (81 * 384))[i] = oc_at(jt);
  };

  // ---- conv1 over this wave's tiles [T0, T0 + NT): T1 -> split3 records in T2
  auto conv1_pass = [&](auto t0_tag, auto nt_tag, auto&& hook) {
    constexpr int T0 = decltype(t0_tag)::value, NT = decltype(nt_tag)::value;
    i32x4 s_hi[NT], s_mid[NT], s_lo[NT];
    int a1base[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s_hi[t] = s_mid[t] = s_lo[t] = i32x4{0, 0, 0, 0};
      const int rt = min(rg1 + 2 * (T0 + t), 24);
      const int m = rt * 16 + li;
      const int oy = m / 20, ox = m - oy * 20;
      a1base[t] = g * F::PLANE1 + (oy * F::GW + ox) * 16;
    }
    constexpr int TOT = 4 * NT, D = 4;
    uint4 x[D];
    auto a_issue = [&](auto idx_tag, int slot) {
      constexpr int IDX = decltype(idx_tag)::value, KS = IDX / NT, T = IDX - KS * NT;
      x[slot] = *reinterpret_cast<const uint4*>(t1 + a1base[T] + ((KS >> 1) * F::GW + (KS & 1)) * 16);
    };
    static_for<D>([&](auto i) { a_issue(i, decltype(i)::value); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<TOT>([&](auto it) {
      constexpr int IDX = decltype(it)::value, KS = IDX / NT, T = IDX - KS * NT, SLOT = IDX % D;
      const i32x4 xv = __builtin_bit_cast(i32x4, x[SLOT]);
      if constexpr (!kNoConv1) {
        s_hi[T] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd[KS][0], xv, s_hi[T], 0, 0, 0);
        s_mid[T] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd[KS][1], xv, s_mid[T], 0, 0, 0);
        s_lo[T] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wd[KS][2], xv, s_lo[T], 0, 0, 0);
      } else {
        asm volatile("" ::"v"(xv));
      }
      if constexpr (IDX + D < TOT) a_issue(IC<IDX + D>{}, SLOT);
      hook(IC<T0 * 4 + IDX>{});
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int t = 0; t < (kNoEpi1 ? 0 : NT); ++t) {
      // (the odd group's thirteenth tile is tile 24 again: the same values to the same addresses as the even group's)
      const int rt = min(rg1 + 2 * (T0 + t), 24);
      const int m = rt * 16 + li;
      const int y = m / 20, xx = m - y * 20;
      // the exact integer sum S_hi * 2^16 + S_mid * 2^8 + S_lo enters f32 in two halves (|S_mid * 256 + S_lo| < 2^31; the
      // product with 65536 is exact, so the fused form rounds once where mul + add rounded once too), then ONE rounding
      // for scale and bias (fused: the file is compiled -ffp-contract=off, so the fusion is spelled out)
      const i32x4 ml = s_mid[t] * 256 + s_lo[t];
      const f32x4 hf = __builtin_convertvector(s_hi[t], f32x4), lf = __builtin_convertvector(ml, f32x4);
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float u = __builtin_fmaf(hf[r], 65536.0f, lf[r]);
        const float y = __builtin_fmaf(u, sc1[r], bv1[r]);
        v[r] = y > 0.f ? y : 0.f;
      }
      uint2 p0, p1, p2;
      split3_4(v, p0, p1, p2);
      uint8_t* rec = t2 + (size_t)(y * F::RQ + xx * F::Q) * 16 + ch1 * 2;
      *reinterpret_cast<uint2*>(rec) = p0;
      *reinterpret_cast<uint2*>(rec + 64) = p1;
      *reinterpret_cast<uint2*>(rec + 128) = p2;
      if constexpr (A1OUT)  // the learner's copy: f32 channel-last
        *reinterpret_cast<f32x4*>(a1_out + ((size_t)n * 400 + m) * 32 + ch1) = v;
    }
  };

  // first frame into T1
#pragma unroll
  for (int j = 0; j < F::IT; ++j) g_load1(n, j);
#pragma unroll
  for (int j = 0; j < F::IT; ++j) s_store(j);
  __syncthreads();

  auto copy_hook = [&](auto idx_tag) {
    constexpr int IDX = decltype(idx_tag)::value;
    // copy-out slots: read chunk j from O behind conv1's item 2 j, store it behind item 2 j + 1 (first pass: 20 items)
    // (measured r5, same box: chunks spaced six items apart instead of one 302 us against 303; a conv1 ring of 8 instead of
    // 4 items 315)
    if constexpr (!kNoCopyOut && IDX < 2 * F::OIT) {
      if constexpr (IDX % 2 == 0) o_read(IC<IDX / 2>{});
      else o_write(IC<IDX / 2>{});
    }
  };
  auto no_hook = [](auto) {};
  static_assert(2 * F::OIT <= 20, "copy-out slots inside conv1's first pass");

  for (; n < N; n += nblk) {
    const int nn = (n + nblk < N) ? n + nblk : n;  // (the last round re-stages its own frame)
    conv1_pass(IC<0>{}, IC<5>{}, copy_hook);
    conv1_pass(IC<5>{}, IC<4>{}, no_hook);
    conv1_pass(IC<9>{}, IC<4>{}, no_hook);
    if constexpr (kNoEpi1) asm volatile("" ::"v"(sc1), "v"(bv1));
    __syncthreads();  // T2 complete, T1 and O free
    // ---- conv2 from T2, tile by tile; the next frame's cells go into T1 in the second half
    {
      constexpr int NTILE = 6, TOT = NTILE * F::KS2, D = 3;
      uint32_t xb[NTILE];
#pragma unroll
      for (int t = 0; t < NTILE; ++t) {
        const int m = t * 16 + li, mm = m < 81 ? m : 80;
        const int oy = mm / 9, ox = mm - oy * 9;
        xb[t] = (uint32_t)((2 * oy * F::RQ + 2 * ox * F::Q + g) * 16);
      }
      uint4 xr[D][3];
      auto a_issue = [&](auto idx_tag, int slot) {
        constexpr int IDX = decltype(idx_tag)::value, T = IDX / F::KS2, KS = IDX - T * F::KS2;
        const uint8_t* ap = t2 + xb[T] + F::koff2(KS);
        xr[slot][0] = *reinterpret_cast<const uint4*>(ap);
        xr[slot][1] = *reinterpret_cast<const uint4*>(ap + 64);
        xr[slot][2] = *reinterpret_cast<const uint4*>(ap + 128);
      };
      static_for<D>([&](auto i) { a_issue(i, decltype(i)::value); });
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc = bv2, accs = {0.f, 0.f, 0.f, 0.f};
      static_for<TOT>([&](auto it) {
        constexpr int IDX = decltype(it)::value, T = IDX / F::KS2, KS = IDX - T * F::KS2, SLOT = IDX % D;
        const bf16x8 x0 = __builtin_bit_cast(bf16x8, xr[SLOT][0]), x1 = __builtin_bit_cast(bf16x8, xr[SLOT][1]),
                     x2 = __builtin_bit_cast(bf16x8, xr[SLOT][2]);
        if constexpr (!kNoConv2) {
          accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[KS][2], x0, accs, 0, 0, 0);
          accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[KS][0], x2, accs, 0, 0, 0);
          accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[KS][1], x1, accs, 0, 0, 0);
          accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[KS][1], x0, accs, 0, 0, 0);
          accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[KS][0], x1, accs, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[KS][0], x0, acc, 0, 0, 0);
        } else {
          asm volatile("" ::"v"(x0), "v"(x1), "v"(x2));
        }
        if constexpr (IDX + D < TOT) a_issue(IC<IDX + D>{}, SLOT);
        // the next frame's cells: loaded behind items 1, 5, 9, ..., stored (sign bits flipped) in the second half
        static_for<F::IT>([&](auto jj) {
          constexpr int JJ = decltype(jj)::value;
          if constexpr (IDX == 1 + 4 * JJ && !kNoStage) g_load1(nn, JJ);
          if constexpr (IDX == TOT / 2 + 4 * JJ && !kNoStage) s_store(JJ);
        });
        if constexpr (KS == F::KS2 - 1 && kNoEpi2) asm volatile("" ::"v"(acc), "v"(accs));
        if constexpr (KS == F::KS2 - 1 && !kNoEpi2) {  // tile T complete: ReLU, split, its record slice into O
          f32x4 v = acc + accs;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
          uint2 p0, p1, p2;
          split3_4(v, p0, p1, p2);
          const int m = T * 16 + li;
          uint8_t* rec = (m < 81 ? otile + m * F::OROW : spare) + ch2 * 2;
          *reinterpret_cast<uint2*>(rec) = p0;
          *reinterpret_cast<uint2*>(rec + 128) = p1;
          *reinterpret_cast<uint2*>(rec + 256) = p2;
          acc = bv2, accs = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    __syncthreads();  // O complete, T1 ready, T2 free
    prev = n;
  }
  {  // the last frame's output tile
    uint4* dst = reinterpret_cast<uint4*>(out + (size_t)prev * (81 * 384));
    for (int i = tid; i < F::OV16; i += F::kT) {
      const int px = i / 24, u = i - px * 24;
      dst[i] = *reinterpret_cast<const uint4*>(otile + px * F::OROW + u * 16);
    }
  }
}

}  // namespace s3
}  // namespace rela_amd
