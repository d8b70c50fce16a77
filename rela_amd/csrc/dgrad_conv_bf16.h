// dgrad_conv_bf16.h -- data gradients of conv2 and conv3 as transposed convolutions on the bf16 matrix cores (the
// learners' bf16x2 mode), fused with the ReLU mask of the layer below.  They replace "GEMM into a column buffer +
// gather-form col2im" (csrc/learner_common.h), whose column buffers (B = 512: 85 + 58 MB; R2D2's 5,312 training
// frames: 881 + 600 MB) are written and read back through HBM.
//
// conv2 (4 x 4, stride 2):  d_a1[y][x][c] = [a1 > 0] * sum_{kh,kw,oc} d_a2[(y-kh)/2][(x-kw)/2][oc] * W2[oc][c][kh][kw].
//   By output parity (s, r) = (y & 1, x & 1), with y = 2u + s, x = 2v + r, kh = 2p + s, kw = 2q + r:
//     d_a1[2u+s][2v+r][c] = sum_{p,q in {0,1}} sum_oc d_a2[u - p][v - q][oc] * W2[oc][c][2p+s][2q+r]
//   -- four stride-1 correlations with 2 x 2 taps over the zero-padded 9 x 9 gradient: per class a GEMM of M = 100
//   pixels, K = 4 taps x 64 oc = 256, N = 32 channels.  The A operand is the gradient itself: [pixel][oc] with oc
//   contiguous is exactly an MFMA row fragment, so the LDS tile is the frame's split records and a tap is a record
//   offset (out-of-range taps point at a zero record).  Wave w: class w >> 1, channel tile w & 1, its 8 x (hi, lo)
//   weight fragments resident in registers for the whole launch.
// conv3 (3 x 3, stride 1):  d_a2[y][x][c] = [a2 > 0] * sum_{kh,kw,oc} d_a3[y - kh][x - kw][oc] * W3[oc][c][kh][kw]:
//   M = 81 pixels, K = 9 taps x 64 = 576, N = 64.  Wave w: channel tile w & 3, pixel tiles 3 (w >> 2) .. + 2, its 18
//   hi weight fragments resident in registers, the lo ones of all four channel tiles in LDS (72 KB).
// Both: hi + lo bf16 operands, three MFMAs per product, f32 accumulation; MFMA issued with the weights as the first
// operand, so a lane holds four consecutive channels of one output pixel: one 16-byte mask load, one 16-byte store.
// One persistent block per CU, gradient tiles double buffered in LDS (next frame's loads in registers during the
// MFMAs), one barrier per frame, A fragments three steps ahead in a register ring, the ReLU mask rows of a frame's
// outputs loaded before its MFMAs.  conv2 moves 654 MB of compulsory traffic in 0.138 ms at R2D2's shape (4.7 TB/s).
#pragma once
#include "prof.h"
#include <hip/hip_runtime.h>

#include "common.h"

namespace rela_amd {
namespace dgfast {
namespace {  // (included by both learners' translation units)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 512;
constexpr int REC = 288;  // LDS bytes per pixel record: 64 hi (128 B) | 64 lo (128 B) | 32 pad (18 units: conflict-free)

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
  const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
  const bf16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, f32x2), bf16x2);
  const bf16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, f32x2), bf16x2);
  hi = make_uint2(__builtin_bit_cast(uint32_t, ha), __builtin_bit_cast(uint32_t, hb));
  lo = make_uint2(__builtin_bit_cast(uint32_t, la), __builtin_bit_cast(uint32_t, lb));
}

// ---- weight fragments (MFMA 16x16x32 operand order, hi and lo): 8 bf16 per lane and k-step ----
// conv2: frag[cls 4][nt 2][ks 8][hl 2][lane 64]; k-step ks = tap (p, q) = ((ks >> 1) >> 1, (ks >> 1) & 1), oc half ks & 1;
//        lane (li, g) element j: W2[oc = 32 (ks & 1) + 8 g + j][c = 16 nt + li][kh = 2 p + s][kw = 2 q + r]
//        read from the permuted copy w2p[oc][(kh * 4 + kw) * 32 + c]
__global__ void pack_dgrad2_frags(const float* __restrict__ w2p, uint4* __restrict__ frag) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (cls, nt, ks, lane)
  if (idx >= 4 * 2 * 8 * 64) return;
  const int lane = idx & 63, ks = (idx >> 6) & 7, nt = (idx >> 9) & 1, cls = idx >> 10;
  const int li = lane & 15, g = lane >> 4;
  const int s = cls >> 1, r = cls & 1, tap = ks >> 1, p = tap >> 1, q = tap & 1;
  const int kh = 2 * p + s, kw = 2 * q + r, c = 16 * nt + li;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = w2p[(size_t)(32 * (ks & 1) + 8 * g + j) * 512 + (kh * 4 + kw) * 32 + c];
  uint2 h0, l0, h1, l1;
  split4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
  split4(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
  uint4* dst = frag + ((size_t)((cls * 2 + nt) * 8 + ks) * 2) * 64 + lane;
  dst[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
  dst[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
}
// conv3: frag[nt 4][ks 18][hl 2][lane 64]; k-step ks = tap ks >> 1 = kh * 3 + kw, oc half ks & 1;
//        element j: W3[oc = 32 (ks & 1) + 8 g + j][c = 16 nt + li][kh][kw] from w3p[oc][(kh * 3 + kw) * 64 + c]
__global__ void pack_dgrad3_frags(const float* __restrict__ w3p, uint4* __restrict__ frag) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (nt, ks, lane)
  if (idx >= 4 * 18 * 64) return;
  const int lane = idx & 63, ks = (idx >> 6) % 18, nt = (idx >> 6) / 18;
  const int li = lane & 15, g = lane >> 4;
  const int tap = ks >> 1, c = 16 * nt + li;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = w3p[(size_t)(32 * (ks & 1) + 8 * g + j) * 576 + tap * 64 + c];
  uint2 h0, l0, h1, l1;
  split4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
  split4(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
  uint4* dst = frag + ((size_t)(nt * 18 + ks) * 2) * 64 + lane;
  dst[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
  dst[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

// ---- conv2 ----
struct D2 {
  static constexpr int PIX = 81, TILE = (PIX + 1) * REC, LDS = 2 * TILE;  // + the zero record; double buffered
  static constexpr int QUADS = PIX * 16;                                   // float4 per frame of d_a2
};
__global__ __launch_bounds__(kT) void dgrad_conv2_bf16(const float* __restrict__ d_a2, const uint4* __restrict__ frag,
                                                       const float* __restrict__ a1, float* __restrict__ d_a1,
                                                       int frames) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int cls = wave >> 1, nt = wave & 1;
  const int s = cls >> 1, r = cls & 1;
  for (int i = tid; i < D2::LDS / 16; i += kT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

  bf16x8 wh[8], wl[8];
  {
    const uint4* fp = frag + (size_t)((cls * 2 + nt) * 8) * 2 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      wh[ks] = __builtin_bit_cast(bf16x8, fp[(size_t)(ks * 2) * 64]);
      wl[ks] = __builtin_bit_cast(bf16x8, fp[(size_t)(ks * 2 + 1) * 64]);
    }
  }
  // record offsets of this lane's output pixels (7 tiles of 16: m = 16 mt + li = 10 u + v) for the four taps
  int aoff[7][4];
#pragma unroll
  for (int mt = 0; mt < 7; ++mt) {
    const int m = mt * 16 + li, u = m / 10, v = m - u * 10;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int oy = u - (t >> 1), ox = v - (t & 1);
      const bool ok = m < 100 && oy >= 0 && oy < 9 && ox >= 0 && ox < 9;
      aoff[mt][t] = (ok ? oy * 9 + ox : D2::PIX) * REC + g * 16;
    }
  }
  float4 x0, x1, x2;
#define DG2_I(J) min(tid + (J) * kT, D2::QUADS - 1)
#define DG2_LOAD(F)                                                                   \
  do {                                                                                \
    const float* fd__ = d_a2 + (size_t)(F) * (81 * 64);                               \
    x0 = *reinterpret_cast<const float4*>(fd__ + (size_t)DG2_I(0) * 4);               \
    x1 = *reinterpret_cast<const float4*>(fd__ + (size_t)DG2_I(1) * 4);               \
    x2 = *reinterpret_cast<const float4*>(fd__ + (size_t)DG2_I(2) * 4);               \
  } while (0)
  auto put = [&](uint8_t* tile, int i, float4 v) {
    uint2 hi, lo;
    split4(v, hi, lo);
    uint8_t* rec = tile + (i >> 4) * REC + (i & 15) * 8;
    *reinterpret_cast<uint2*>(rec) = hi;
    *reinterpret_cast<uint2*>(rec + 128) = lo;
  };
  int f = blockIdx.x;
  DG2_LOAD(f);  // (blocks <= frames)
  __syncthreads();  // zero fill done
  put(smem, DG2_I(0), x0), put(smem, DG2_I(1), x1), put(smem, DG2_I(2), x2);
  {
    const int fn = (f + (int)gridDim.x < frames) ? f + (int)gridDim.x : f;
    DG2_LOAD(fn);
  }
  __syncthreads();
  int buf = 0;
  for (; f < frames; f += gridDim.x) {
    const uint8_t* tile = smem + buf * D2::TILE;
    // the ReLU mask rows of this frame's outputs, issued before the MFMAs (loaded in the epilogue their latency was
    // exposed once per frame: 8.5 us per frame against 2.3 us of MFMAs)
    f32x4 am[7];
#pragma unroll
    for (int mt = 0; mt < 7; ++mt) {
      const int m = min(mt * 16 + li, 99), u = m / 10, v = m - u * 10;
      am[mt] = *reinterpret_cast<const f32x4*>(a1 + ((size_t)f * 400 + (2 * u + s) * 20 + (2 * v + r)) * 32 + 16 * nt + 4 * g);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[7];
#pragma unroll
    for (int mt = 0; mt < 7; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      // A fragments three (k-step, tile) pairs ahead of their MFMAs in a register ring; the sched_barriers keep the
      // scheduler from sinking each LDS read to its use (every MFMA triple would wait for its own read)
      constexpr int TOT = 8 * 7, D = 3;
      uint4 rh[D], rl[D];
      auto a_issue = [&](int idx, int slot) {
        const int ks = idx / 7, mt = idx - ks * 7;
        const uint8_t* ap = tile + aoff[mt][ks >> 1] + (ks & 1) * 64;
        rh[slot] = *reinterpret_cast<const uint4*>(ap);
        rl[slot] = *reinterpret_cast<const uint4*>(ap + 128);
      };
#pragma unroll
      for (int i = 0; i < D; ++i) a_issue(i, i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int idx = 0; idx < TOT; ++idx) {
        const int ks = idx / 7, mt = idx - ks * 7, slot = idx % D;
        const bf16x8 ah = __builtin_bit_cast(bf16x8, rh[slot]), al = __builtin_bit_cast(bf16x8, rl[slot]);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks], ah, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], al, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], ah, acc[mt], 0, 0, 0);
        if (idx + D < TOT) a_issue(idx + D, slot);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the next frame's records go into the other buffer (last read two frames ago: every wave is past that barrier)
    uint8_t* ntile = smem + (buf ^ 1) * D2::TILE;
    put(ntile, DG2_I(0), x0), put(ntile, DG2_I(1), x1), put(ntile, DG2_I(2), x2);
    {
      const int f2 = f + 2 * (int)gridDim.x;
      DG2_LOAD(f2 < frames ? f2 : f);
    }
    // lane: channels 16 nt + 4 g .. + 3 of output pixel m = 16 mt + li -> (y, x) = (2 u + s, 2 v + r)
#pragma unroll
    for (int mt = 0; mt < 7; ++mt) {
      const int m = mt * 16 + li;
      if (m < 100) {
        const int u = m / 10, v = m - u * 10;
        const size_t o = ((size_t)f * 400 + (2 * u + s) * 20 + (2 * v + r)) * 32 + 16 * nt + 4 * g;
        const f32x4 a = am[mt];
        f32x4 d;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) d[rr] = a[rr] > 0.f ? acc[mt][rr] : 0.f;
        *reinterpret_cast<f32x4*>(d_a1 + o) = d;
      }
    }
    __syncthreads();
    buf ^= 1;
  }
#undef DG2_LOAD
#undef DG2_I
}

// ---- conv3 ----
struct D3 {
  static constexpr int PIX = 49, TILE = (PIX + 1) * REC;
  static constexpr int WL = 4 * 18 * 64 * 16;  // the lo weight fragments of all four channel tiles: 73,728 B
  static constexpr int LDS = 2 * TILE + WL;
  static constexpr int QUADS = PIX * 16;
};
__global__ __launch_bounds__(kT) void dgrad_conv3_bf16(const float* __restrict__ d_a3, const uint4* __restrict__ frag,
                                                       const float* __restrict__ a2, float* __restrict__ d_a2,
                                                       int frames) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int nt = wave & 3, mh = wave >> 2;
  for (int i = tid; i < 2 * D3::TILE / 16; i += kT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  // hi weight fragments in registers (72), lo ones in LDS (they would be another 72 registers: with them the mask
  // rows below could not be prefetched)
  uint4* wls = reinterpret_cast<uint4*>(smem + 2 * D3::TILE);
  for (int i = tid; i < 4 * 18 * 64; i += kT) wls[i] = frag[(size_t)((i >> 6) * 2 + 1) * 64 + (i & 63)];
  const uint4* wlp = wls + (size_t)nt * 18 * 64 + lane;
  bf16x8 wh[18];
  {
    const uint4* fp = frag + (size_t)(nt * 18) * 2 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) wh[ks] = __builtin_bit_cast(bf16x8, fp[(size_t)(ks * 2) * 64]);
  }
  // output pixels m = 16 (3 mh + mt) + li = 9 y + x (81 of 96), taps (kh, kw): source (y - kh, x - kw) in the 7 x 7 grid
  int aoff[3][9];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) {
    const int m = (3 * mh + mt) * 16 + li, y = m / 9, x = m - y * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int oy = y - t / 3, ox = x - t % 3;
      const bool ok = m < 81 && oy >= 0 && oy < 7 && ox >= 0 && ox < 7;
      aoff[mt][t] = (ok ? oy * 7 + ox : D3::PIX) * REC + g * 16;
    }
  }
  float4 x0, x1;
#define DG3_I(J) min(tid + (J) * kT, D3::QUADS - 1)
#define DG3_LOAD(F)                                                                   \
  do {                                                                                \
    const float* fd__ = d_a3 + (size_t)(F) * (49 * 64);                               \
    x0 = *reinterpret_cast<const float4*>(fd__ + (size_t)DG3_I(0) * 4);               \
    x1 = *reinterpret_cast<const float4*>(fd__ + (size_t)DG3_I(1) * 4);               \
  } while (0)
  auto put = [&](uint8_t* tile, int i, float4 v) {
    uint2 hi, lo;
    split4(v, hi, lo);
    uint8_t* rec = tile + (i >> 4) * REC + (i & 15) * 8;
    *reinterpret_cast<uint2*>(rec) = hi;
    *reinterpret_cast<uint2*>(rec + 128) = lo;
  };
  int f = blockIdx.x;
  DG3_LOAD(f);  // (blocks <= frames)
  __syncthreads();  // zero fill done
  put(smem, DG3_I(0), x0), put(smem, DG3_I(1), x1);
  {
    const int fn = (f + (int)gridDim.x < frames) ? f + (int)gridDim.x : f;
    DG3_LOAD(fn);
  }
  __syncthreads();
  int buf = 0;
  for (; f < frames; f += gridDim.x) {
    const uint8_t* tile = smem + buf * D3::TILE;
    f32x4 am[3];  // ReLU mask rows, issued before the MFMAs
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      const int m = min((3 * mh + mt) * 16 + li, 80);
      am[mt] = *reinterpret_cast<const f32x4*>(a2 + ((size_t)f * 81 + m) * 64 + 16 * nt + 4 * g);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      constexpr int TOT = 18 * 3, D = 3;  // (register ring as in dgrad_conv2_bf16; the lo weights one k-step ahead)
      uint4 rh[D], rl[D], wr[2];
      auto a_issue = [&](int idx, int slot) {
        const int ks = idx / 3, mt = idx - ks * 3;
        const uint8_t* ap = tile + aoff[mt][ks >> 1] + (ks & 1) * 64;
        rh[slot] = *reinterpret_cast<const uint4*>(ap);
        rl[slot] = *reinterpret_cast<const uint4*>(ap + 128);
      };
      wr[0] = wlp[0];
#pragma unroll
      for (int i = 0; i < D; ++i) a_issue(i, i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int idx = 0; idx < TOT; ++idx) {
        const int ks = idx / 3, mt = idx - ks * 3, slot = idx % D;
        if (mt == 0 && ks + 1 < 18) wr[(ks + 1) & 1] = wlp[(ks + 1) * 64];
        const bf16x8 wlk = __builtin_bit_cast(bf16x8, wr[ks & 1]);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, rh[slot]), al = __builtin_bit_cast(bf16x8, rl[slot]);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlk, ah, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], al, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], ah, acc[mt], 0, 0, 0);
        if (idx + D < TOT) a_issue(idx + D, slot);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    uint8_t* ntile = smem + (buf ^ 1) * D3::TILE;
    put(ntile, DG3_I(0), x0), put(ntile, DG3_I(1), x1);
    {
      const int f2 = f + 2 * (int)gridDim.x;
      DG3_LOAD(f2 < frames ? f2 : f);
    }
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      const int m = (3 * mh + mt) * 16 + li;
      if (m < 81) {
        const size_t o = ((size_t)f * 81 + m) * 64 + 16 * nt + 4 * g;
        const f32x4 a = am[mt];
        f32x4 d;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) d[rr] = a[rr] > 0.f ? acc[mt][rr] : 0.f;
        *reinterpret_cast<f32x4*>(d_a2 + o) = d;
      }
    }
    __syncthreads();
    buf ^= 1;
  }
#undef DG3_LOAD
#undef DG3_I
}

constexpr size_t kFrag2Bytes = (size_t)4 * 2 * 8 * 2 * 64 * 16, kFrag3Bytes = (size_t)4 * 18 * 2 * 64 * 16;

// The weight fragments alone (a caller that keeps them across steps packs them when the weights change and passes
// them as `frags` below)
inline void pack_frags(const float* w2p, const float* w3p, void* frag2, void* frag3, hipStream_t s) {
  hipLaunchKernelGGL(pack_dgrad2_frags, dim3(ceil_div(4 * 2 * 8 * 64, 256)), dim3(256), 0, s, w2p, (uint4*)frag2);
  hipLaunchKernelGGL(pack_dgrad3_frags, dim3(ceil_div(4 * 18 * 64, 256)), dim3(256), 0, s, w3p, (uint4*)frag3);
}

// d_a1 = [a1 > 0] * conv2^T(d_a2); `scratch` (>= kFrag2Bytes) receives the weight fragments unless `frags` has them
inline int launch_conv2(const float* d_a2, const float* w2p, const float* a1, float* d_a1, int frames, void* scratch,
                        hipStream_t s, const void* frags = nullptr) {
  // (initialised once, thread-safely: launches may come from several host threads)
  static const hipError_t attr_set =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad_conv2_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, D2::LDS);
  RELA_HIP(attr_set);
  if (!frags) hipLaunchKernelGGL(pack_dgrad2_frags, dim3(ceil_div(4 * 2 * 8 * 64, 256)), dim3(256), 0, s, w2p, (uint4*)scratch);
  note_launch("dgrad_conv2_bf16");
  hipLaunchKernelGGL(dgrad_conv2_bf16, dim3(frames < 256 ? frames : 256), dim3(kT), D2::LDS, s, d_a2,
                     (const uint4*)(frags ? frags : scratch), a1, d_a1, frames);
  return RELA_OK;
}
inline int launch_conv3(const float* d_a3, const float* w3p, const float* a2, float* d_a2, int frames, void* scratch,
                        hipStream_t s, const void* frags = nullptr) {
  // (initialised once, thread-safely: launches may come from several host threads)
  static const hipError_t attr_set =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad_conv3_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, D3::LDS);
  RELA_HIP(attr_set);
  if (!frags) hipLaunchKernelGGL(pack_dgrad3_frags, dim3(ceil_div(4 * 18 * 64, 256)), dim3(256), 0, s, w3p, (uint4*)scratch);
  note_launch("dgrad_conv3_bf16");
  hipLaunchKernelGGL(dgrad_conv3_bf16, dim3(frames < 256 ? frames : 256), dim3(kT), D3::LDS, s, d_a3,
                     (const uint4*)(frags ? frags : scratch), a2, d_a2, frames);
  return RELA_OK;
}

}  // namespace
}  // namespace dgfast
}  // namespace rela_amd
