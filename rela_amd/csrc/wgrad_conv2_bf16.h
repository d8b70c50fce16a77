// wgrad_conv2_bf16.h -- conv2's weight gradient on the bf16 matrix cores (the learners' bf16x2 mode).
//
//   dW2[oc][c][kh][kw] = sum_{b, oy, ox} d_a2[b][oy][ox][oc] * a1[b][2 oy + kh][2 ox + kw][c]        (4 x 4, stride 2)
//
// Same construction as wgrad_conv1_bf16.h: the contraction runs over pixels; with kh = 2 p + s, kw = 2 q + r the
// input splits into four sub-images I_{s,r}[c][y'][x'] = a1[2 y' + s][2 x' + r][c] (10 x 10) and
//   dW2[oc][c][2p+s][2q+r] = sum_{oy,ox} d[oy][ox][oc] * I_{s,r}[c][oy + p][ox + q]:
// the row tap p is an address offset of the image operand (rows are 16 elements = 32 B apart), the column tap q a
// second copy of the gradient operand shifted by one pixel.  Both operands are f32, so both are hi + lo bf16 and a
// product is three MFMAs (hi*hi + lo*hi + hi*lo).
//   k index of a frame: k = 16 oy + ox', ox' = ox + q (zeros where ox is out of range), 9 rows = 144 -> 160 = 5 k-steps.
//   Wave w owns the 16 channels cg = w >> 2 of sub-image sr = w & 3 and its four (p, q) taps: 4 of the 32 column
//   tiles for all 64 output channels (16 accumulator tiles); no two waves share an output element.
//   LDS: sub-images [hi, lo][s r][c][11 rows][16] bf16 (368 B apart) 94 KB | d^T of HALF the output channels
//        [q][hi, lo][32 oc][160 k] bf16 (rows 352 B apart) 45 KB: the two halves are staged and multiplied in turn.
// One persistent block per CU, frames strided over the blocks, the next frame's raw data in registers during the
// MFMAs.  Output: part[block][oc][(kh * 4 + kw) * 32 + c] for reduce_splits (kRedConv2).
#pragma once
#include "prof.h"
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"

namespace rela_amd {
namespace w2fast {
namespace {  // (included by both learners' translation units)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 512;
constexpr int SUB = 368;                     // bytes per sub-image (one channel): 11 rows x 16 el x 2 B + 16
constexpr int IMG_HALF = 4 * 32 * SUB;       // one of hi / lo: 47,104
constexpr int IMG_BYTES = 2 * IMG_HALF;      // 94,208
constexpr int DT_PITCH = 352;                // bytes per (copy, oc) row: 160 k x 2 B + 32
constexpr int DT_BYTES = 4 * 32 * DT_PITCH;  // [q][hi, lo][32 oc] = 45,056
constexpr int LDS_TOTAL = IMG_BYTES + DT_BYTES;
constexpr int kMaxBlocks = 256;
constexpr int kA1Quads = 400 * 8, kDQuads = 81 * 8;  // float4 per frame of a1; per frame and oc half of d_a2

__device__ __forceinline__ void split2(float x, uint16_t& hi, uint16_t& lo) {
  const bf16x2 h = __builtin_convertvector(f32x2{x, 0.f}, bf16x2);
  hi = (uint16_t)(__builtin_bit_cast(uint32_t, h) & 0xffffu);
  const float r = x - __uint_as_float((uint32_t)hi << 16);
  const bf16x2 l = __builtin_convertvector(f32x2{r, 0.f}, bf16x2);
  lo = (uint16_t)(__builtin_bit_cast(uint32_t, l) & 0xffffu);
}

__global__ __launch_bounds__(kT) void wgrad_conv2_bf16(const float* __restrict__ a1, const float* __restrict__ d_a2,
                                                       int frames, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* img = smem;
  uint8_t* dt = smem + IMG_BYTES;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;

  for (int i = tid; i < LDS_TOTAL / 16; i += kT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

  f32x4 acc[4][4];  // [m tile (16 oc)][(p, q)]
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging registers: the next frame.  a1's float4 are numbered (channel quad, parity class, cell of the 10 x 10
  // sub-image): a wave's lanes then store consecutive cells of ONE sub-image (conflict-free 2-byte stores; with
  // consecutive PIXELS neighbouring lanes alternate between two sub-images 11,776 B = 0 banks apart: 3.7 x the store
  // time, tools/lds_conflicts.py).  d_a2's are numbered (channel quad, pixel) as in wgrad_conv1_bf16.h. ----
  float4 x0, x1, x2, x3, x4, x5, x6;  // a1: 3,200 float4 = 8 quads x 400 pixels, seven per thread (clamped)
  float4 e0, e1, e2, e3;              // d_a2: 2 x 648 float4 = per oc half 8 quads x 81 pixels, two per thread and half
#define W2_AI(J) min(tid + (J) * kT, kA1Quads - 1)
#define W2_DI(J) min(tid + (J) * kT, kDQuads - 1)
#define W2_APX(I) ((2 * (((I) % 100) / 10) + (((I) % 400) / 200)) * 20 + 2 * (((I) % 100) % 10) + ((((I) % 400) / 100) & 1))
#define W2_ALOAD(J) (*reinterpret_cast<const float4*>(fa__ + W2_APX(W2_AI(J)) * 32 + (W2_AI(J) / 400) * 4))
#define W2_DLOAD(J, H) (*reinterpret_cast<const float4*>(fd__ + (W2_DI(J) % 81) * 64 + (H) * 32 + (W2_DI(J) / 81) * 4))
#define W2_LOAD(F)                                            \
  do {                                                        \
    const float* fa__ = a1 + (size_t)(F) * (400 * 32);        \
    const float* fd__ = d_a2 + (size_t)(F) * (81 * 64);       \
    x0 = W2_ALOAD(0), x1 = W2_ALOAD(1), x2 = W2_ALOAD(2), x3 = W2_ALOAD(3);  \
    x4 = W2_ALOAD(4), x5 = W2_ALOAD(5), x6 = W2_ALOAD(6);     \
    e0 = W2_DLOAD(0, 0), e1 = W2_DLOAD(1, 0), e2 = W2_DLOAD(0, 1), e3 = W2_DLOAD(1, 1);  \
  } while (0)

  auto put_a = [&](int ai, float4 v) {
    const int c0 = (ai / 400) * 4, rest = ai % 400, sr = rest / 100, cl = rest % 100;  // (= W2_APX's decomposition)
    uint8_t* cell = img + (sr * 32 + c0) * SUB + ((cl / 10) * 16 + cl % 10) * 2;
    const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint16_t hi, lo;
      split2(f[t], hi, lo);
      *reinterpret_cast<uint16_t*>(cell + t * SUB) = hi;
      *reinterpret_cast<uint16_t*>(cell + t * SUB + IMG_HALF) = lo;
    }
  };
  auto put_d = [&](int di, float4 v) {
    const int px = di % 81, oc0 = (di / 81) * 4;  // oc within the half
    const int oy = px / 9, ox = px - oy * 9;
    const int k0 = oy * 16 + ox;
    const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint16_t hi, lo;
      split2(f[t], hi, lo);
      uint8_t* r0 = dt + (oc0 + t) * DT_PITCH + k0 * 2;
      *reinterpret_cast<uint16_t*>(r0) = hi;                      // q = 0, hi
      *reinterpret_cast<uint16_t*>(r0 + 32 * DT_PITCH) = lo;      // q = 0, lo
      *reinterpret_cast<uint16_t*>(r0 + 64 * DT_PITCH + 2) = hi;  // q = 1 (shifted by one pixel), hi
      *reinterpret_cast<uint16_t*>(r0 + 96 * DT_PITCH + 2) = lo;  // q = 1, lo
    }
  };

  const int cg = wave >> 2, sr = wave & 3;
  const int b_base = (sr * 32 + cg * 16 + li) * SUB;
  const int a_base = li * DT_PITCH;

  // MFMAs of one oc half (m tiles 2 H, 2 H + 1) over the five k-steps
  auto mma_half = [&](auto half) {
    constexpr int H = decltype(half)::value;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const int G = ks * 4 + g;                       // k-group of 8: row G >> 1, columns 8 (G & 1) ..
      const int bo = b_base + ((G >> 1) * 16 + (G & 1) * 8) * 2;
      const int ao = a_base + (32 * ks + 8 * g) * 2;
      uint4 af[2][2][2];  // [q][hi, lo][m]
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
          for (int m = 0; m < 2; ++m)
            af[q][hl][m] = *reinterpret_cast<const uint4*>(dt + ((q * 2 + hl) * 32 + m * 16) * DT_PITCH + ao);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(img + bo + p * 32));
        const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(img + IMG_HALF + bo + p * 32));
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            f32x4& c = acc[2 * H + m][p * 2 + q];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[q][1][m]), bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[q][0][m]), bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[q][0][m]), bh, c, 0, 0, 0);
          }
      }
    }
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;

  int f = blockIdx.x;
  W2_LOAD(f);  // (blocks <= frames: f is a frame)
  __syncthreads();  // zero fill done
  for (; f < frames; f += gridDim.x) {
    put_a(W2_AI(0), x0), put_a(W2_AI(1), x1), put_a(W2_AI(2), x2), put_a(W2_AI(3), x3);
    put_a(W2_AI(4), x4), put_a(W2_AI(5), x5), put_a(W2_AI(6), x6);
    put_d(W2_DI(0), e0), put_d(W2_DI(1), e1);
    const float4 k2 = e2, k3 = e3;  // the second oc half waits here while the next frame's loads are issued
    {
      const int fn = (f + (int)gridDim.x < frames) ? f + (int)gridDim.x : f;  // (the last round re-reads its own frame)
      W2_LOAD(fn);
    }
    __syncthreads();
    mma_half(H0{});
    __syncthreads();  // d^T of half 0 read
    put_d(W2_DI(0), k2), put_d(W2_DI(1), k3);
    __syncthreads();
    mma_half(H1{});
    __syncthreads();  // tiles read: the next frame may overwrite them
  }
#undef W2_LOAD
#undef W2_ALOAD
#undef W2_APX
#undef W2_DLOAD
  // every wave owns its columns: straight to this block's partial tile, n = (kh * 4 + kw) * 32 + c
  float* out = part + (size_t)blockIdx.x * (64 * 512);
  const int s = sr >> 1, r = sr & 1, c = cg * 16 + li;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int p = n >> 1, q = n & 1;
      const int col = ((2 * p + s) * 4 + (2 * q + r)) * 32 + c;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) out[(size_t)(m * 16 + 4 * g + rr) * 512 + col] = acc[m][n][rr];
    }
#undef W2_AI
#undef W2_DI
}

inline int launch(const float* a1, const float* d_a2, int frames, float* part, hipStream_t s, int* blocks_out) {
  // (initialised once, thread-safely: launches may come from several host threads)
  static const hipError_t attr_set =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_conv2_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
  RELA_HIP(attr_set);
  const int blocks = frames < kMaxBlocks ? frames : kMaxBlocks;
  note_launch("wgrad_conv2_bf16");
  hipLaunchKernelGGL(wgrad_conv2_bf16, dim3(blocks), dim3(kT), LDS_TOTAL, s, a1, d_a2, frames, part);
  *blocks_out = blocks;
  return RELA_OK;
}

}  // namespace
}  // namespace w2fast
}  // namespace rela_amd
