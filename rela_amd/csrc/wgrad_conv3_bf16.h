// wgrad_conv3_bf16.h -- conv3's weight gradient on the bf16 matrix cores (the learners' bf16x2 mode).
//
//   dW3[oc][c][kh][kw] = sum_{b, oy, ox} d_a3[b][oy][ox][oc] * a2[b][oy + kh][ox + kw][c]             (3 x 3, stride 1)
//
// As wgrad_conv1_bf16.h / wgrad_conv2_bf16.h without the space-to-depth step (stride 1): the contraction runs over
// pixels, the row tap kh is an address offset of the image operand (rows 16 elements = 32 B apart) and the column tap
// kw selects one of THREE copies of the gradient operand, shifted by 0, 1 and 2 pixels.  Both operands hi + lo bf16,
// three MFMAs per product.
//   k index of a frame: k = 16 oy + ox', ox' = ox + kw (zeros elsewhere), 7 rows = 112 -> 128 = 4 k-steps.
//   Wave w owns the 16 input channels cg = w & 3 and the output channels 32 (w >> 2) .. + 31 with all nine taps:
//   2 x 9 accumulator tiles; no two waves share an output element.
//   LDS: image [hi, lo][c 64][10 rows][16] bf16 (336 B apart) 43 KB | d^T [kw 3][hi, lo][oc 64][128 k] bf16 (rows
//        288 B apart) 111 KB.
// Output: part[block][oc][(kh * 3 + kw) * 64 + c] for reduce_splits (kRedConv3).
#pragma once
#include "prof.h"
#include <hip/hip_runtime.h>

#include "common.h"

namespace rela_amd {
namespace w3fast {
namespace {  // (included by both learners' translation units)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 512;
constexpr int SUB = 336;                     // bytes per channel image: 10 rows x 16 el x 2 B + 16
constexpr int IMG_HALF = 64 * SUB;           // one of hi / lo: 21,504
constexpr int IMG_BYTES = 2 * IMG_HALF;      // 43,008
constexpr int DT_PITCH = 288;                // bytes per (copy, oc) row: 128 k x 2 B + 32
constexpr int DT_BYTES = 6 * 64 * DT_PITCH;  // [kw][hi, lo][64 oc] = 110,592
constexpr int LDS_TOTAL = IMG_BYTES + DT_BYTES;
constexpr int kMaxBlocks = 256;
constexpr int kAQuads = 81 * 16, kDQuads = 49 * 16;  // float4 per frame of a2 / of d_a3

__device__ __forceinline__ void split2(float x, uint16_t& hi, uint16_t& lo) {
  const bf16x2 h = __builtin_convertvector(f32x2{x, 0.f}, bf16x2);
  hi = (uint16_t)(__builtin_bit_cast(uint32_t, h) & 0xffffu);
  const float r = x - __uint_as_float((uint32_t)hi << 16);
  const bf16x2 l = __builtin_convertvector(f32x2{r, 0.f}, bf16x2);
  lo = (uint16_t)(__builtin_bit_cast(uint32_t, l) & 0xffffu);
}

__global__ __launch_bounds__(kT) void wgrad_conv3_bf16(const float* __restrict__ a2, const float* __restrict__ d_a3,
                                                       int frames, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* img = smem;
  uint8_t* dt = smem + IMG_BYTES;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;

  for (int i = tid; i < LDS_TOTAL / 16; i += kT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

  f32x4 acc[2][9];  // [m tile][kh * 3 + kw]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 9; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging registers: the next frame, quad-major numbering (see wgrad_conv1_bf16.h)
  float4 x0, x1, x2, e0, e1;
#define W3_AI(J) min(tid + (J) * kT, kAQuads - 1)
#define W3_DI(J) min(tid + (J) * kT, kDQuads - 1)
#define W3_ALOAD(J) (*reinterpret_cast<const float4*>(fa__ + (W3_AI(J) % 81) * 64 + (W3_AI(J) / 81) * 4))
#define W3_DLOAD(J) (*reinterpret_cast<const float4*>(fd__ + (W3_DI(J) % 49) * 64 + (W3_DI(J) / 49) * 4))
#define W3_LOAD(F)                                         \
  do {                                                     \
    const float* fa__ = a2 + (size_t)(F) * (81 * 64);      \
    const float* fd__ = d_a3 + (size_t)(F) * (49 * 64);    \
    x0 = W3_ALOAD(0), x1 = W3_ALOAD(1), x2 = W3_ALOAD(2);  \
    e0 = W3_DLOAD(0), e1 = W3_DLOAD(1);                    \
  } while (0)

  auto put_a = [&](int ai, float4 v) {
    const int px = ai % 81, c0 = (ai / 81) * 4;
    const int y = px / 9, x = px - y * 9;
    uint8_t* cell = img + c0 * SUB + (y * 16 + x) * 2;
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
    const int px = di % 49, oc0 = (di / 49) * 4;
    const int oy = px / 7, ox = px - oy * 7;
    const int k0 = oy * 16 + ox;
    const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint16_t hi, lo;
      split2(f[t], hi, lo);
      uint8_t* r0 = dt + (oc0 + t) * DT_PITCH + k0 * 2;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {  // copy kw = the gradient shifted by kw pixels
        *reinterpret_cast<uint16_t*>(r0 + (kw * 2) * 64 * DT_PITCH + kw * 2) = hi;
        *reinterpret_cast<uint16_t*>(r0 + (kw * 2 + 1) * 64 * DT_PITCH + kw * 2) = lo;
      }
    }
  };

  const int cg = wave & 3, mh = wave >> 2;
  const int b_base = (cg * 16 + li) * SUB;
  const int a_base = (mh * 32 + li) * DT_PITCH;

  int f = blockIdx.x;
  W3_LOAD(f);  // (blocks <= frames: f is a frame)
  __syncthreads();  // zero fill done
  for (; f < frames; f += gridDim.x) {
    put_a(W3_AI(0), x0), put_a(W3_AI(1), x1), put_a(W3_AI(2), x2);
    put_d(W3_DI(0), e0), put_d(W3_DI(1), e1);
    {
      const int fn = (f + (int)gridDim.x < frames) ? f + (int)gridDim.x : f;  // (the last round re-reads its own frame)
      W3_LOAD(fn);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int G = ks * 4 + g;  // k-group of 8: row G >> 1, columns 8 (G & 1) ..
      const int bo = b_base + ((G >> 1) * 16 + (G & 1) * 8) * 2;
      const int ao = a_base + (32 * ks + 8 * g) * 2;
      uint4 af[3][2][2];  // [kw][hi, lo][m]
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
          for (int m = 0; m < 2; ++m)
            af[kw][hl][m] = *reinterpret_cast<const uint4*>(dt + ((kw * 2 + hl) * 64 + m * 16) * DT_PITCH + ao);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(img + bo + kh * 32));
        const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(img + IMG_HALF + bo + kh * 32));
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            f32x4& c = acc[m][kh * 3 + kw];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[kw][1][m]), bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[kw][0][m]), bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[kw][0][m]), bh, c, 0, 0, 0);
          }
      }
    }
    __syncthreads();  // tiles read: the next frame may overwrite them
  }
#undef W3_LOAD
#undef W3_ALOAD
#undef W3_DLOAD
  float* out = part + (size_t)blockIdx.x * (64 * 576);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 9; ++n)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        out[(size_t)(mh * 32 + m * 16 + 4 * g + rr) * 576 + n * 64 + cg * 16 + li] = acc[m][n][rr];
#undef W3_AI
#undef W3_DI
}

inline int launch(const float* a2, const float* d_a3, int frames, float* part, hipStream_t s, int* blocks_out) {
  // (initialised once, thread-safely: launches may come from several host threads)
  static const hipError_t attr_set =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_conv3_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
  RELA_HIP(attr_set);
  const int blocks = frames < kMaxBlocks ? frames : kMaxBlocks;
  note_launch("wgrad_conv3_bf16");
  hipLaunchKernelGGL(wgrad_conv3_bf16, dim3(blocks), dim3(kT), LDS_TOTAL, s, a2, d_a3, frames, part);
  *blocks_out = blocks;
  return RELA_OK;
}

}  // namespace
}  // namespace w3fast
}  // namespace rela_amd
