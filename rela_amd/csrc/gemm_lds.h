// gemm_lds.h -- generic LDS-tiled MFMA GEMM (v_mfma_f32_16x16x4_f32, f32 throughout), both
// operands staged through LDS by loader functors, optional split-K over blockIdx.z.
// Used by the learner's backward pass (learner.hip) and the small-batch fc forward (ffnet.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"
#include "prof.h"

namespace rela_amd {
namespace gemm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kLT = 512;  // 8 wavefronts
constexpr int BK = 32;    // K chunk staged per barrier (8 MFMA k-steps)

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- generic LDS-tiled MFMA GEMM:  C[M][N] = sum_k A(m,k) * B(k,n) ---------------------------
// A arrives either k-contiguous (AMC = false: loadA(m, k) -> A[m][k..k+3]) or m-contiguous
// (AMC = true: loadA(k, m) -> A[m..m+3][k], the transposed operand of a weight gradient);
// B is always n-contiguous: loadB(k, n) -> B[k][n..n+3].  Loaders return zeros out of range.
template <int BM_, int BN_, int WM_, int WN_, bool AMC_>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr bool AMC = AMC_;
  static_assert(WM * WN == 8, "8 wavefronts per block");
  static constexpr int TM = BM / 16 / WM, TN = BN / 16 / WN;  // 16x16 tiles per wave
  static_assert(TM >= 1 && TN >= 1, "tile too small for the wave grid");
  // leading dimensions = 16 (mod 32) floats, or k+2: a fragment read (16 rows x 4 k) touches
  // every bank exactly twice, the minimum for 64 lanes
  static constexpr int LDA = AMC ? BM + 16 : BK + 2;
  static constexpr int A_FLOATS = AMC ? BK * LDA : BM * LDA;
  static constexpr int LDB = BN + 16;
  static constexpr int B_FLOATS = BK * LDB;
  static constexpr int A_V4 = BM * BK / 4, B_V4 = BN * BK / 4;
  static constexpr int A_IT = (A_V4 + kLT - 1) / kLT, B_IT = (B_V4 + kLT - 1) / kLT;
};

template <class T, class P>
__global__ __launch_bounds__(kLT) void gemm_lds(const P p) {
  __shared__ __attribute__((aligned(16))) float sA[2][T::A_FLOATS];
  __shared__ __attribute__((aligned(16))) float sB[2][T::B_FLOATS];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, kk = lane >> 4;
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int m0 = blockIdx.y * T::BM, n0 = blockIdx.x * T::BN;
  const int nch = (p.K + BK - 1) / BK;
  const int c0 = blockIdx.z * p.kslice;
  const int c1 = min(nch, c0 + p.kslice);

  float4 ra[T::A_IT], rb[T::B_IT];
  auto gload = [&](int ch) {
    const int k0 = ch * BK;
#pragma unroll
    for (int j = 0; j < T::A_IT; ++j) {
      const int idx = tid + j * kLT;
      if (idx < T::A_V4) {
        if constexpr (T::AMC) {
          const int kr = idx / (T::BM / 4), q = idx % (T::BM / 4);
          ra[j] = p.loadA(k0 + kr, m0 + 4 * q);
        } else {
          const int r = idx >> 3, q = idx & 7;
          ra[j] = p.loadA(m0 + r, k0 + 4 * q);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < T::B_IT; ++j) {
      const int idx = tid + j * kLT;
      if (idx < T::B_V4) {
        const int kr = idx / (T::BN / 4), q = idx % (T::BN / 4);
        rb[j] = p.loadB(k0 + kr, n0 + 4 * q);
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < T::A_IT; ++j) {
      const int idx = tid + j * kLT;
      if (idx < T::A_V4) {
        if constexpr (T::AMC) {
          const int kr = idx / (T::BM / 4), q = idx % (T::BM / 4);
          *reinterpret_cast<float4*>(&sA[buf][kr * T::LDA + 4 * q]) = ra[j];
        } else {
          const int r = idx >> 3, q = idx & 7;
          float* d = &sA[buf][r * T::LDA + 4 * q];
          *reinterpret_cast<float2*>(d) = make_float2(ra[j].x, ra[j].y);
          *reinterpret_cast<float2*>(d + 2) = make_float2(ra[j].z, ra[j].w);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < T::B_IT; ++j) {
      const int idx = tid + j * kLT;
      if (idx < T::B_V4) {
        const int kr = idx / (T::BN / 4), q = idx % (T::BN / 4);
        *reinterpret_cast<float4*>(&sB[buf][kr * T::LDB + 4 * q]) = rb[j];
      }
    }
  };

  f32x4 acc[T::TM][T::TN];
#pragma unroll
  for (int t = 0; t < T::TM; ++t)
#pragma unroll
    for (int u = 0; u < T::TN; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (c0 < c1) {
    gload(c0);
    sstore(0);
  }
  __syncthreads();
  for (int ch = c0; ch < c1; ++ch) {
    const int buf = (ch - c0) & 1;
    if (ch + 1 < c1) gload(ch + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      float a[T::TM], b[T::TN];
#pragma unroll
      for (int t = 0; t < T::TM; ++t) {
        const int row = (wm * T::TM + t) * 16 + li;
        a[t] = T::AMC ? sA[buf][(4 * ks + kk) * T::LDA + row] : sA[buf][row * T::LDA + 4 * ks + kk];
      }
#pragma unroll
      for (int u = 0; u < T::TN; ++u) b[u] = sB[buf][(4 * ks + kk) * T::LDB + (wn * T::TN + u) * 16 + li];
#pragma unroll
      for (int t = 0; t < T::TM; ++t)
#pragma unroll
        for (int u = 0; u < T::TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[u], acc[t][u], 0, 0, 0);
    }
    if (ch + 1 < c1) sstore(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int t = 0; t < T::TM; ++t)
#pragma unroll
    for (int u = 0; u < T::TN; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + (wm * T::TM + t) * 16 + kk * 4 + r;
        const int n = n0 + (wn * T::TN + u) * 16 + li;
        if (m < p.M && n < p.N) p.store(blockIdx.z, m, n, acc[t][u][r]);
      }
}

struct ProbBase {
  int M, N, K, kslice;
};

template <class T, class P>
void launch_gemm(P p, int splits, hipStream_t s, const char* name) {
  const int nch = ceil_div(p.K, BK);
  p.kslice = ceil_div(nch, splits);
  ProfScope prof(name, s);
  note_launch("gemm_lds (f32)");
  hipLaunchKernelGGL((gemm_lds<T, P>), dim3(ceil_div(p.N, T::BN), ceil_div(p.M, T::BM), splits), dim3(kLT), 0, s, p);
}

}  // namespace gemm
}  // namespace rela_amd
