// conv_img_s3.h -- conv3 of the AtariFFNet trunk (pyrela/net.py:26-27: 64 -> 64 channels, 3x3, stride 1) in the f32x3
// arithmetic of gemm_f32emu.h, over split3 records (gemm_s3.h), as an IMAGE kernel:
//
//   * the chip's L2 delivers ~10 TB/s to LDS-DMA gathers (measured r5, gemm_s3: 1.9 GB in 197 us), and an im2col GEMM
//     with 64 output channels needs 32 B/clk/CU of activations alone at the bf16 MFMA rate -- so the input frames are
//     staged ONCE (31 KB per frame, read once from HBM, whole 128-byte lines, LDS-DMA) into a four-slot LDS ring and the
//     nine taps read them from there;
//   * the weights are RESIDENT IN REGISTERS: a wave owns 16 output channels, 18 k-steps x 3 parts x 4 registers = 216 of
//     its 512 (one block of four waves per CU); nothing but activations moves in the loop;
//   * rows are tiled across frame boundaries (16-row MFMA tiles over the block's 49 f rows: no padding of 49 to 64);
//     a tile touches at most two frames; frame F + 2 is issued when frame F is first needed (one barrier per frame);
//   * LDS image of a frame: pixel (y, x) at unit (16 B) y * RQ + x * Q, Q = 26 (24 units of record + 2), RQ = 246,
//     slot stride 2,250: consecutive OUTPUT pixels advance by 10 units mod 16 (also across output rows and frames),
//     so the 16 lanes a ds_read_b128 pass serves hit the 16 bank groups once each.  The LDS-DMA destination is
//     lane-linear, so each lane fetches the 16 bytes that belong at ITS unit of the image (pad units fetch byte 0 of
//     the frame: an L2 hit);
//   * per tile: 18 k-steps x (3 fragment reads + 6 MFMAs), fragments three k-steps ahead in a register ring, LDS
//     reads in inline asm with counted lgkmcnt (see gemm_s3.h on why).
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_s3.h"

namespace rela_amd {
namespace s3 {

struct Conv3Img {
  static constexpr int Q = 26, RQ = 246, SQ = 2250;  // pixel / row / slot stride in 16-byte units
  static constexpr int SLOT_BYTES = SQ * 16, NSLOT = 4;
  static constexpr int CHUNKS = (9 * RQ + 63) / 64;  // 1 KB LDS-DMA instructions per frame (35)
  static constexpr int G = (CHUNKS + 3) / 4;         // per wave (9; the last wave's ninth is a padding load)
  static constexpr int SPARE = NSLOT * SLOT_BYTES;
  static constexpr int LDS_BYTES = SPARE + 1024;
  static_assert(CHUNKS * 64 <= SQ && LDS_BYTES <= 160 * 1024, "LDS budget");
  static constexpr int KS = 18;
  // k-step ks -> tap in the order of pack_f32emu_at mode 2 (f32emu::ProbConv3::tap), as byte offset inside the image
  static constexpr int tap_of(int ks) {
    const int i = ks >> 1, kh = i / 3, r = i - 3 * kh;
    return kh * 3 + ((kh & 1) ? 2 - r : r);
  }
  static constexpr int koff(int ks) {
    const int tp = tap_of(ks);
    return ((tp / 3) * RQ + (tp % 3) * Q) * 16 + (ks & 1) * 64;
  }
};

// a2: records [N][81] x 384 B; Wp: pack_f32emu_at mode 2 ([ks][u][part][lane] x 8 bf16); out: records [N][49] x 384 B
__global__ __launch_bounds__(256, 1) void conv3_img_s3(const uint8_t* __restrict__ a2, const uint4* __restrict__ Wp,
                                                       const float* __restrict__ bias, uint8_t* __restrict__ out, int N) {
  using C = Conv3Img;
  __shared__ __attribute__((aligned(1024))) uint8_t lds[C::LDS_BYTES];
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)lds;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nb = gridDim.x, b = blockIdx.x;
  const int f0 = (int)((int64_t)N * b / nb), f1 = (int)((int64_t)N * (b + 1) / nb);
  const int nf = f1 - f0;
  if (nf <= 0) return;  // (block-uniform)
  const int rows = nf * 49, ntiles = (rows + 15) >> 4;
  const uint8_t* src0 = a2 + (size_t)f0 * (81 * 384);

  // ---- LDS-DMA: this lane's source offset inside a frame for each of its wave's chunks
  uint32_t goff[C::G], gdst[C::G];
#pragma unroll
  for (int j = 0; j < C::G; ++j) {
    const int c = wave + 4 * j;
    const int L = 64 * c + lane;
    const int y = L / C::RQ, rem = L - y * C::RQ;
    const int x = rem / C::Q, off = rem - x * C::Q;
    const bool valid = c < C::CHUNKS && y < 9 && x < 9 && off < 24;
    goff[j] = valid ? (uint32_t)((y * 9 + x) * 384 + off * 16) : 0u;
    gdst[j] = c < C::CHUNKS ? (uint32_t)c * 1024u : (uint32_t)C::SPARE;  // (uniform)
  }
  auto issue_frame = [&](int f) {  // frame f of the block (clamped: past the last one the loads are padding)
    const uint8_t* src = src0 + (size_t)min(f, nf - 1) * (81 * 384);
    const uint32_t slot = (uint32_t)(f & 3) * C::SLOT_BYTES;
#pragma unroll
    for (int j = 0; j < C::G; ++j)
      glds16(src + goff[j], lds + (f < nf && gdst[j] != (uint32_t)C::SPARE ? slot + gdst[j] : (uint32_t)C::SPARE));
  };

  // ---- resident weights and bias
  bf16x8 w[C::KS][3];
  {
    const uint4* wp = Wp + (size_t)wave * 3 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
      for (int p = 0; p < 3; ++p) w[ks][p] = __builtin_bit_cast(bf16x8, wp[(size_t)(ks * TN * 3 + p) * 64]);
  }
  f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 16 * wave + 4 * g);
  // (every ordinary load has returned before the first LDS-DMA is issued: the compiler drains vmcnt at the first use of
  // a load result while an LDS-DMA is in flight)
  asm volatile("" : "+v"(bv));
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
    for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(w[ks][p]));

  issue_frame(0);
  issue_frame(1);
  wait_vm<C::G>();
  __builtin_amdgcn_s_barrier();
  issue_frame(2);
  int cur = 0;  // newest frame whose lines are visible

  auto frag_addr = [&](int t) {  // this lane's pixel of tile t: address of its record's unit g in the LDS image
    const int m = min(16 * t + li, rows - 1);
    const int fr = m / 49, pos = m - 49 * fr;
    const int oy = pos / 7, ox = pos - 7 * oy;
    return lds0 + (uint32_t)(fr & 3) * C::SLOT_BYTES + (uint32_t)((oy * C::RQ + ox * C::Q + g) * 16);
  };
  // The fragment ring (three k-steps deep) runs on ACROSS tiles: the last three k-steps of a tile issue the first three
  // of the next one, whose reads then complete under the epilogue; they are waited for before the loop's back edge, so
  // no register holds an LDS read in flight where the compiler could touch it.
  uint32_t xa = frag_addr(0);
  u32x4 x[3][3];
  static_for<3>([&](auto kk) {
    constexpr int K = decltype(kk)::value;
    x[K][0] = lds_read128<C::koff(K)>(xa);
    x[K][1] = lds_read128<C::koff(K) + 128>(xa);
    x[K][2] = lds_read128<C::koff(K) + 256>(xa);
  });
  static_for<3>([&](auto kk) { wait_lgkm<0>(x[decltype(kk)::value][0], x[decltype(kk)::value][1], x[decltype(kk)::value][2]); });

  for (int t = 0; t < ntiles; ++t) {
    f32x4 acc = bv, accs = {0.f, 0.f, 0.f, 0.f}, acct = {0.f, 0.f, 0.f, 0.f};
    uint32_t xn = xa;
    static_for<C::KS>([&](auto kk) {
      constexpr int K = decltype(kk)::value, R = K % 3;
      if constexpr (K == C::KS - 6) {
        // the next tile's frame: its lines must be visible before that tile's first reads (issued at k-step KS - 3)
        const int tn = min(t + 1, ntiles - 1);
        const int last = min(16 * tn + 15, rows - 1) / 49;  // (uniform) newest frame the next tile reads
#if defined(S3_ABLATE_FRAMES)
        if (false) {
#else
        if (last > cur) {
#endif
          wait_vm<C::G>();               // frame cur + 1 has landed (frame cur + 2 may still be in flight)
          __builtin_amdgcn_s_barrier();  // ... for every wave; this tile reads frame cur only: cur - 1 is free
          cur = last;
          issue_frame(cur + 2);
        }
        xn = frag_addr(tn);
      }
      wait_lgkm<6>(x[R][0], x[R][1], x[R][2]);
      const bf16x8 x0 = __builtin_bit_cast(bf16x8, x[R][0]), x1 = __builtin_bit_cast(bf16x8, x[R][1]),
                   x2 = __builtin_bit_cast(bf16x8, x[R][2]);
      // the five small products have accumulators of their own (two chains), added once per tile
      accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[K][2], x0, accs, 0, 0, 0);
      acct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[K][0], x2, acct, 0, 0, 0);
      accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[K][1], x1, accs, 0, 0, 0);
      acct = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[K][1], x0, acct, 0, 0, 0);
      accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[K][0], x1, accs, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[K][0], x0, acc, 0, 0, 0);
      if constexpr (K + 3 < C::KS) {
        x[R][0] = lds_read128<C::koff(K + 3)>(xa);
        x[R][1] = lds_read128<C::koff(K + 3) + 128>(xa);
        x[R][2] = lds_read128<C::koff(K + 3) + 256>(xa);
      } else {
        x[R][0] = lds_read128<C::koff(K + 3 - C::KS)>(xn);
        x[R][1] = lds_read128<C::koff(K + 3 - C::KS) + 128>(xn);
        x[R][2] = lds_read128<C::koff(K + 3 - C::KS) + 256>(xn);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- epilogue: ReLU, split, this lane's four channels of its pixel
#if defined(S3_ABLATE_EPILOGUE)
    asm volatile("" ::"v"(acc), "v"(accs), "v"(acct));
    if (false) {
      f32x4 v = acc + (accs + acct);
#else
    if (16 * t + li < rows) {
      f32x4 v = acc + (accs + acct);
#endif
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      uint2 p0, p1, p2;
      split3_4(v, p0, p1, p2);
      uint8_t* o = out + ((size_t)f0 * 49 + (size_t)(16 * t + li)) * 384 + (16 * wave + 4 * g) * 2;
#if defined(S3_ABLATE_STORES)
      asm volatile("" ::"v"(p0), "v"(p1), "v"(p2), "v"(o));
#else
      *reinterpret_cast<uint2*>(o) = p0;
      *reinterpret_cast<uint2*>(o + 128) = p1;
      *reinterpret_cast<uint2*>(o + 256) = p2;
#endif
    }
    static_for<3>([&](auto kk) { wait_lgkm<0>(x[decltype(kk)::value][0], x[decltype(kk)::value][1], x[decltype(kk)::value][2]); });
    xa = xn;
  }
  wait_vm<0>();  // no LDS-DMA may outlive the workgroup's LDS allocation
}

inline void launch_conv3_img(const void* a2, const uint4* Wp, const float* bias, void* out, int N, hipStream_t s, int blocks = 256) {
  const int nb = std::max(1, std::min(blocks, N));
  hipLaunchKernelGGL(conv3_img_s3, dim3(nb), dim3(256), 0, s, reinterpret_cast<const uint8_t*>(a2), Wp, bias,
                     reinterpret_cast<uint8_t*>(out), N);
}

}  // namespace s3
}  // namespace rela_amd
