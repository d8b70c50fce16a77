// wgrad_conv1_bf16.h -- conv1's weight gradient on the bf16 matrix cores (the learners' bf16x2 mode).
//
//   dW1[oc][c][kh][kw] = (1/255) sum_{b, oy, ox} d_a1[b][oy][ox][oc] * u8[b][c][4 oy + kh][4 ox + kw]
//
// The contraction runs over PIXELS, so an MFMA operand needs 8 consecutive pixels of one patch element j = (c, kh, kw)
// per lane -- a stride-4 gather in the image.  Space-to-depth removes it: with kh = 4 p + s, kw = 4 q + r the frame
// splits into 16 sub-images I_{s,r}[c][y'][x'] = u8[c][4 y' + s][4 x' + r] (21 x 21 each) and
//   dW1[oc][c][4p+s][4q+r] = sum_{oy,ox} d[oy][ox][oc] * I_{s,r}[c][oy + p][ox + q]:
// a 2 x 2-tap correlation per sub-image whose taps are plain row (p) and column (q) offsets.  The row offset is an
// address offset of the B operand; the column offset of ONE element (2 bytes: no aligned 16-byte read) is moved to
// the A operand instead, which is kept twice: d and d shifted by one pixel.  u8 is exact in bf16 and d = hi + lo, so
// a product costs two MFMAs and is accurate to 2^-17.
//   k index of half a frame (10 output rows): k = 24 oy_rel + ox', ox' = ox + q in 0..23 (zeros where ox is out of
//   range), 256 k per half = 8 k-steps of 32, ONE per wave; the wave keeps all 2 x 16 accumulator tiles
//   (32 oc x 256 j) and the eight waves' sums are folded in a fixed order at the end of the launch.
//   LDS: sub-images [c][s r][12 rows][24] bf16 (592 B apart: 16 lanes on 16 different 16-byte bank groups) |
//        d^T [q][hi, lo][oc][256 k] bf16 (rows 544 B apart).  Cells nobody writes are zeroed once and stay zero.
//   One persistent block per CU, frames strided over the blocks; the next half's raw data (2 x 16 B of u8, 4 x 16 B
//   of d) waits in registers during the MFMAs.
// Output: part[block][oc][j] partial sums in state_dict order for reduce_splits (which applies the 1/255).
#pragma once
#include "prof.h"
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace rela_amd {
namespace w1fast {
namespace {  // (included by both learners' translation units)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 512;
constexpr int SUB = 592;                  // bytes per sub-image: 12 rows x 24 el x 2 B + 16
constexpr int IMG_BYTES = 64 * SUB;       // 37,888
constexpr int DT_PITCH = 544;             // bytes per (copy, oc) row: 256 k x 2 B + 32
constexpr int DT_BYTES = 4 * 32 * DT_PITCH;  // 69,632
constexpr int LDS_TOTAL = IMG_BYTES + DT_BYTES;
constexpr int kMaxBlocks = 256;
static_assert(DT_BYTES >= 32 * 256 * 4, "the final reduction reuses the d^T area");

__device__ __forceinline__ uint16_t bf16_bits(float x) {  // RNE; exact for the integers 0..255
  const bf16x2 h = __builtin_convertvector(f32x2{x, 0.f}, bf16x2);
  return (uint16_t)(__builtin_bit_cast(uint32_t, h) & 0xffffu);
}

__global__ __launch_bounds__(kT) void wgrad_conv1_bf16(const uint8_t* __restrict__ obs, const float* __restrict__ d_a1,
                                                       int frames, float* __restrict__ part,
                                                       unsigned long long* __restrict__ stamps = nullptr) {
  // diagnostic (RELA_W1_STAMPS=1): shader clocks of block 0 / thread 0 -> stamps[0..15]
  int stamp_i = 0;
  auto stamp = [&]() {
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0 && stamp_i < 16) stamps[stamp_i++] = __builtin_amdgcn_s_memtime();
  };
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* img = smem;
  uint8_t* dt = smem + IMG_BYTES;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;

  for (int i = tid; i < LDS_TOTAL / 16; i += kT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

  f32x4 acc[2][16];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 16; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging registers: the half frame after the one being multiplied ----
  // image: a thread takes 4 x 4 space-to-depth CELLS -- rows 4 y' .. 4 y' + 3, columns 4 x' .. 4 x' + 3 of one plane, four
  // dword loads -- whose 16 pixels go to the SAME element (y', x') of the plane's 16 sub-images: sixteen 2-byte stores
  // at constant offsets from one address (before late r3 a thread took 16 consecutive bytes of a row and derived
  // sub-image, row and column per byte; the image part of the staging went from ~970 to ~830 cycles per half frame,
  // but it is the 64 two-byte stores of d^T per thread that dominate a half's 7.3 k cycles: RELA_W1_STAMPS, ~2.5 k for
  // d^T + 1.7 k waiting for the slowest wave, against 1.4 k of MFMAs).  924 cells per half (4 planes x 11 x 21), two per thread (clamped: the last
  // threads repeat cell 923, writing the same values twice).  d: 1,600 float4 (8 channel quads x 200 pixels), four per
  // thread (threads past 1,600 repeat number 1,599).  Quad-major numbering: the 64 lanes of a wave hold 64 consecutive
  // pixels of ONE channel quad, so their 2-byte LDS stores fall on 32 different banks (pixel-major numbering put a
  // wave's eight quads, whose rows are 2,176 B apart, on the same four banks: 8-way conflicts on every store).
  uint32_t im[2][4];
  float4 d0, d1, d2, d3;
  int cell_g[2], cell_l[2];  // a cell's byte offset inside a half frame (plane, row 4 y', column 4 x') and in LDS
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = min(tid + j * kT, 923);
    const int pl = c / 231, rem = c - pl * 231;
    const int yq = rem / 21, xq = rem - yq * 21;
    cell_g[j] = pl * 7056 + 4 * yq * 84 + 4 * xq;
    cell_l[j] = pl * 16 * SUB + (yq * 24 + xq) * 2;
  }
  const int di0 = tid, di1 = tid + kT, di2 = tid + 2 * kT, di3 = min(tid + 3 * kT, 1599);
  int d_l[4];  // (channel quad, pixel) -> byte offset of its hi element in the q = 0 copy of d^T
  {
    const int dis[4] = {di0, di1, di2, di3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int px = dis[j] % 200, oc0 = (dis[j] / 200) * 4;
      const int oy = px / 20, ox = px - oy * 20;
      d_l[j] = oc0 * DT_PITCH + (oy * 24 + ox) * 2;
    }
  }
#define W1_LOAD(F, H)                                                                                        \
  do {                                                                                                       \
    const uint8_t* fo__ = obs + (size_t)(F) * 28224 + (H) * (40 * 84);                                        \
    _Pragma("unroll") for (int j__ = 0; j__ < 2; ++j__)                                                       \
      _Pragma("unroll") for (int r__ = 0; r__ < 4; ++r__)                                                     \
        im[j__][r__] = *reinterpret_cast<const uint32_t*>(fo__ + cell_g[j__] + r__ * 84);                      \
    const float* fd__ = d_a1 + ((size_t)(F) * 400 + (H) * 200) * 32;                                          \
    d0 = *reinterpret_cast<const float4*>(fd__ + (di0 % 200) * 32 + (di0 / 200) * 4);                         \
    d1 = *reinterpret_cast<const float4*>(fd__ + (di1 % 200) * 32 + (di1 / 200) * 4);                         \
    d2 = *reinterpret_cast<const float4*>(fd__ + (di2 % 200) * 32 + (di2 / 200) * 4);                         \
    d3 = *reinterpret_cast<const float4*>(fd__ + (di3 % 200) * 32 + (di3 / 200) * 4);                         \
  } while (0)

  auto put_img = [&](int j) {
    uint8_t* base = img + cell_l[j];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float v = (float)((im[j][r] >> (8 * c)) & 0xffu);  // exact in bf16: its bits are the float's upper half
        *reinterpret_cast<uint16_t*>(base + (r * 4 + c) * SUB) = (uint16_t)(__float_as_uint(v) >> 16);
      }
  };
  auto put_d = [&](int j, float4 v) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    uint8_t* r0 = dt + d_l[j];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint16_t hi = bf16_bits(x[t]);
      const uint16_t lo = bf16_bits(x[t] - __uint_as_float((uint32_t)hi << 16));
      uint8_t* rt = r0 + t * DT_PITCH;                                // copy q = 0, hi
      *reinterpret_cast<uint16_t*>(rt) = hi;
      *reinterpret_cast<uint16_t*>(rt + 32 * DT_PITCH) = lo;          // q = 0, lo
      *reinterpret_cast<uint16_t*>(rt + 64 * DT_PITCH + 2) = hi;      // q = 1 (shifted by one pixel), hi
      *reinterpret_cast<uint16_t*>(rt + 96 * DT_PITCH + 2) = lo;      // q = 1, lo
    }
  };

  // fragment addresses of this wave's k-step (k = 32 wave + 8 g .. + 7)
  const int G = wave * 4 + g;
  const int oyr = G / 3, ox0 = (G - oyr * 3) * 8;
  const int a_off = li * DT_PITCH + (32 * wave + 8 * g) * 2;
  const int b_off = li * SUB + (oyr * 24 + ox0) * 2;

  int f = blockIdx.x;
  stamp();  // 0: kernel start (zero fill issued)
  W1_LOAD(f, 0);  // (blocks <= frames: f is a frame)
  __syncthreads();  // zero fill done
  stamp();  // 1
  for (; f < frames; f += gridDim.x) {
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      stamp();  // half start
      put_img(0);
      put_img(1);
      stamp();  // image staged
      put_d(0, d0);
      put_d(1, d1);
      put_d(2, d2);
      put_d(3, d3);
      {  // the half after this one (the last one re-reads itself)
        const int fn = (h == 0) ? f : ((f + (int)gridDim.x < frames) ? f + (int)gridDim.x : f);
        const int hn = (h == 0) ? 1 : ((f + (int)gridDim.x < frames) ? 0 : 1);
        W1_LOAD(fn, hn);
      }
      stamp();  // d staged, loads issued
      __syncthreads();
      stamp();  // barrier
      uint4 af[2][2][2];  // [q][hi, lo][m tile]
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
          for (int m = 0; m < 2; ++m)
            af[q][hl][m] = *reinterpret_cast<const uint4*>(dt + ((q * 2 + hl) * 32 + m * 16) * DT_PITCH + a_off);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const bf16x8 bf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(img + c * 16 * SUB + p * 48 + b_off));
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              const int n = (c * 2 + p) * 2 + q;
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[q][1][m]), bf, acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[q][0][m]), bf, acc[m][n], 0, 0, 0);
            }
        }
      stamp();  // MFMAs issued
      __syncthreads();  // fragments read: the tiles may be overwritten
    }
  }
  stamp();  // loop done
#undef W1_LOAD
  // ---- fold the eight waves' sums in wave order (deterministic), then one coalesced store per block ----
  float* red = reinterpret_cast<float*>(dt);
  for (int w = 0; w < 8; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 16; ++n) {
          const int c = n >> 2, p = (n >> 1) & 1, q = n & 1;
          const int j = c * 64 + (4 * p + (li >> 2)) * 8 + 4 * q + (li & 3);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* dst = red + (m * 16 + 4 * g + r) * 256 + j;
            *dst = (w == 0) ? acc[m][n][r] : *dst + acc[m][n][r];
          }
        }
    }
    __syncthreads();
  }
  float* out = part + (size_t)blockIdx.x * (32 * 256);
  for (int i = tid; i < 32 * 256 / 4; i += kT)
    reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(red)[i];
}

// max_blocks < kMaxBlocks leaves CUs to kernels of another stream (the Ape-X learner's two-lane backward: with 128 of
// the 256 CUs this kernel takes 66 us instead of 52 alone, but the side lane's conv2 weight gradient next to it 33
// instead of 49, and the step 0.485 -> 0.458 ms; flat between 96 and 160 blocks)
inline int launch(const uint8_t* obs, const float* d_a1, int frames, float* part, hipStream_t s, int* blocks_out,
                  int max_blocks = kMaxBlocks) {
  // (initialised once, thread-safely: launches may come from several host threads)
  static const hipError_t attr_set =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_conv1_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
  RELA_HIP(attr_set);
  const int cap = std::max(1, std::min(kMaxBlocks, max_blocks));
  const int blocks = frames < cap ? frames : cap;
  note_launch("wgrad_conv1_bf16");
  hipLaunchKernelGGL(wgrad_conv1_bf16, dim3(blocks), dim3(kT), LDS_TOTAL, s, obs, d_a1, frames, part, nullptr);
  *blocks_out = blocks;
  return RELA_OK;
}

}  // namespace
}  // namespace w1fast
}  // namespace rela_amd
