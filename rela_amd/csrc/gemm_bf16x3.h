// gemm_bf16x3.h -- the learner's small f32 GEMMs on the bf16 matrix cores, with the SAME loader functors as gemm_lds.h.
//
//   C[M][N] = sum_k A(m,k) * B(k,n),  A and B delivered as f32 by the Problem's loadA / loadB (im2col, transposition and
//   u8 -> f32 conversion stay index arithmetic in the loader), optional split-K over blockIdx.z -> Problem::store(z, ..).
//
// gemm_lds.h multiplies on v_mfma_f32_16x16x4_f32 (157 TFLOP/s peak); at the learner's sizes (batch 512: K = 512 for
// the fc layer, 25 k - 41 k for the conv weight gradients, 64 - 512 output rows) its blocks also wait out every K-chunk's
// load latency.  Here every operand is split on the way INTO LDS into bf16 hi = bf16(x) and lo = bf16(x - hi) (16
// significant bits together) and a product is  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  on v_mfma_f32_16x16x32_bf16 with f32
// accumulation: ~2^-16 relative per product, 16 x the MFMA rate per instruction and 8 x the k per instruction.
//
// LDS images are K-MAJOR for an operand that arrives m- (or n-) contiguous -- the transposed operand of a weight
// gradient, and every B -- : rows = k, columns = m, exactly what the loader's float4 fills with one 8-byte store per
// half, and the MFMA fragment (8 consecutive k of one row m per lane) comes out of TWO ds_read_b64_tr_b16, the
// hardware's transposing read (cdna_hip_programming.md T10): per 16-lane group a block of 4 k-rows x 16 columns is
// delivered column-major.  The 32 k of a step are stored in the order  row(8g + j) = 4g + j (j < 4), 16 + 4g + (j - 4)
// so that the two lane groups of a 32-lane half read 8 CONSECUTIVE image rows; with a row stride of 8 (mod 16) x odd
// banks (BM * 2 + 32 bytes, BM a multiple of 32) those 8 rows x 8 banks cover the 64 banks once: conflict-free.
// An A that arrives k-contiguous (the data-gradient GEMMs) gets an [m][k] image and plain ds_read_b128 fragments.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"
#include "gemm_lds.h"
#include "prof.h"

namespace rela_amd {
namespace gemm3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

constexpr int kT = 512;  // 8 wavefronts
constexpr int BK = 32;   // one MFMA k-step per staged chunk

// image row of logical k (0..31) of a chunk
__device__ __forceinline__ int krow(int k) {
  const int g = k >> 3, j = k & 7;
  return j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4);
}

// four floats -> 4 bf16 hi (8 bytes) and 4 bf16 lo
__device__ __forceinline__ void split4(float4 v, uint2& hi, uint2& lo) {
  const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
  const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
  const bf16x2 la = __builtin_convertvector(a - __builtin_convertvector(ha, f32x2), bf16x2);
  const bf16x2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, f32x2), bf16x2);
  hi = make_uint2(__builtin_bit_cast(uint32_t, ha), __builtin_bit_cast(uint32_t, hb));
  lo = make_uint2(__builtin_bit_cast(uint32_t, la), __builtin_bit_cast(uint32_t, lb));
}

// four floats -> hi, mid, lo (each 4 bf16): x = hi + mid + lo up to 2^-26 |x| (both subtractions are exact in f32)
__device__ __forceinline__ void split4x3(float4 v, uint2& hi, uint2& mid, uint2& lo) {
  const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
  const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
  const f32x2 ra = a - __builtin_convertvector(ha, f32x2), rb = b - __builtin_convertvector(hb, f32x2);
  const bf16x2 ma = __builtin_convertvector(ra, bf16x2), mb = __builtin_convertvector(rb, bf16x2);
  const bf16x2 la = __builtin_convertvector(ra - __builtin_convertvector(ma, f32x2), bf16x2);
  const bf16x2 lb = __builtin_convertvector(rb - __builtin_convertvector(mb, f32x2), bf16x2);
  hi = make_uint2(__builtin_bit_cast(uint32_t, ha), __builtin_bit_cast(uint32_t, hb));
  mid = make_uint2(__builtin_bit_cast(uint32_t, ma), __builtin_bit_cast(uint32_t, mb));
  lo = make_uint2(__builtin_bit_cast(uint32_t, la), __builtin_bit_cast(uint32_t, lb));
}

// AMC as in gemm_lds.h: true = loadA(k, m) returns A[m..m+3][k] (m-contiguous, k-major image, transposed reads);
// false = loadA(m, k) returns A[m][k..k+3] (k-contiguous, [m][k] image, row reads).  B: loadB(k, n) -> B[k][n..n+3].
// PARTS = 2: the split-bf16 fast arithmetic above.  PARTS = 3: every f32 operand as THREE bf16 parts (hi, mid, lo: an exact
// split, see gemm_f32emu.h) and the six products with i + j <= 2, the five small ones in accumulators of their own:
// f32 accuracy at 16 / 6 the f32 MFMA rate -- the learner's GEMMs of the f32x3 mode.
template <int BM_, int BN_, int WM_, int WN_, bool AMC_, int PARTS_ = 2>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, PARTS = PARTS_;
  static constexpr bool AMC = AMC_;
  static_assert(PARTS == 2 || PARTS == 3, "two or three bf16 parts per operand");
  static_assert(WM * WN == 8, "8 wavefronts per block");
  static_assert(BM % 32 == 0 && BN % 32 == 0, "k-major row strides need multiples of 32 columns");
  static constexpr int TM = BM / 16 / WM, TN = BN / 16 / WN;  // 16x16 tiles per wave
  static_assert(TM >= 1 && TN >= 1, "tile too small for the wave grid");
  static constexpr int LDA = AMC ? BM * 2 + 32 : BK * 2 + 16;  // bytes per image row ([k][m] or [m][k])
  static constexpr int A_HALF = AMC ? BK * LDA : BM * LDA;     // bytes of one of hi / lo
  static constexpr int LDB = BN * 2 + 32;
  static constexpr int B_HALF = BK * LDB;
  static constexpr int BUF = PARTS * (A_HALF + B_HALF);        // one stage: A parts, then B parts
  static constexpr int LDS_BYTES = 2 * BUF;
  static constexpr int A_V4 = BM * BK / 4, B_V4 = BN * BK / 4;
  static constexpr int A_IT = (A_V4 + kT - 1) / kT, B_IT = (B_V4 + kT - 1) / kT;
};

// Block -> (N tile, M tile, K split).  Workgroups are dealt round-robin over the 8 XCDs by linear id, so on the plain 3-D
// grid the blocks that read the SAME operand bytes -- the tiles of one K split (they share the split's slice of A and
// B), or, without splits, the M tiles of one N tile (they share B's column block) -- sit on eight different L2s and each
// fetches those bytes from the Infinity Cache / HBM again.  The 1-D maps give such blocks ids that are equal mod 8 and
// consecutive in id / 8 (for speed only: any placement computes the same thing).  Ids past the last tile leave at once.
struct GridMap {
  int gx, gy, gz;
  int mode;  // 0: plain (blockIdx.x / y / z) | 1: splits grouped (gz >= 8) | 2: M tiles of an N tile grouped
};
__device__ __forceinline__ bool grid_map(const GridMap& gm, int& bx, int& by, int& bz) {
  if (gm.mode == 0) {
    bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    return true;
  }
  const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
  if (gm.mode == 1) {
    const int tiles = gm.gx * gm.gy;
    bz = (i / tiles) * 8 + xcd;
    const int t = i % tiles;
    bx = t % gm.gx, by = t / gm.gx;
    return bz < gm.gz;
  }
  bx = (i / gm.gy) * 8 + xcd, by = i % gm.gy, bz = 0;
  return bx < gm.gx;
}

template <class T, class P>
__global__ __launch_bounds__(kT) void gemm_bf16x3(const P p, const GridMap gm) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int bx, by, bz;
  if (!grid_map(gm, bx, by, bz)) return;  // (block-uniform, before any barrier)
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int m0 = by * T::BM, n0 = bx * T::BN;
  const int nch = (p.K + BK - 1) / BK;
  const int c0 = bz * p.kslice;
  const int c1 = min(nch, c0 + p.kslice);

  // Staging registers: D chunks in flight.  A chunk is 24 (or fewer) MFMAs per wave -- far less than one load's
  // latency -- and these GEMMs have only a handful of chunks per block (fc at batch 512: 16), so with ONE chunk
  // ahead (gemm_lds.h) every chunk waited out its loads: 29 us for the fc weight gradient, 16 x ~1.5 us.  Chunk c of
  // the block lives in register set (c - c0) % D; the chunk loop is unrolled by D so that the sets are named statically.
  constexpr int D = 4;
  float4 ra[D][T::A_IT], rb[D][T::B_IT];
  auto gload = [&](int ch, auto set) {
    constexpr int S = decltype(set)::value;
    const int k0 = ch * BK;
#pragma unroll
    for (int j = 0; j < T::A_IT; ++j) {
      const int idx = min(tid + j * kT, T::A_V4 - 1);  // (clamped, not predicated: a surplus thread repeats a neighbour)
      if constexpr (T::AMC) {
        const int kr = idx / (T::BM / 4), q = idx % (T::BM / 4);
        ra[S][j] = p.loadA(k0 + kr, m0 + 4 * q);
      } else {
        const int r = idx >> 3, q = idx & 7;
        ra[S][j] = p.loadA(m0 + r, k0 + 4 * q);
      }
    }
#pragma unroll
    for (int j = 0; j < T::B_IT; ++j) {
      const int idx = min(tid + j * kT, T::B_V4 - 1);
      const int kr = idx / (T::BN / 4), q = idx % (T::BN / 4);
      rb[S][j] = p.loadB(k0 + kr, n0 + 4 * q);
    }
  };
  auto sstore = [&](int buf, auto set) {
    constexpr int S = decltype(set)::value;
    uint8_t* base = smem + buf * T::BUF;
#pragma unroll
    for (int j = 0; j < T::A_IT; ++j) {
      const int idx = min(tid + j * kT, T::A_V4 - 1);
      uint2 hi, mid, lo;
      if constexpr (T::PARTS == 3) split4x3(ra[S][j], hi, mid, lo);
      else split4(ra[S][j], hi, lo);
      int off;
      if constexpr (T::AMC) {
        const int kr = idx / (T::BM / 4), q = idx % (T::BM / 4);
        off = krow(kr) * T::LDA + q * 8;
      } else {
        const int r = idx >> 3, q = idx & 7;
        off = r * T::LDA + q * 8;
      }
      *reinterpret_cast<uint2*>(base + off) = hi;
      if constexpr (T::PARTS == 3) *reinterpret_cast<uint2*>(base + T::A_HALF + off) = mid;
      *reinterpret_cast<uint2*>(base + (T::PARTS - 1) * T::A_HALF + off) = lo;
    }
    uint8_t* bb = base + T::PARTS * T::A_HALF;
#pragma unroll
    for (int j = 0; j < T::B_IT; ++j) {
      const int idx = min(tid + j * kT, T::B_V4 - 1);
      const int kr = idx / (T::BN / 4), q = idx % (T::BN / 4);
      uint2 hi, mid, lo;
      if constexpr (T::PARTS == 3) split4x3(rb[S][j], hi, mid, lo);
      else split4(rb[S][j], hi, lo);
      const int off = krow(kr) * T::LDB + q * 8;
      *reinterpret_cast<uint2*>(bb + off) = hi;
      if constexpr (T::PARTS == 3) *reinterpret_cast<uint2*>(bb + T::B_HALF + off) = mid;
      *reinterpret_cast<uint2*>(bb + (T::PARTS - 1) * T::B_HALF + off) = lo;
    }
  };

  // transposed fragment of a k-major image: rows 4g + q and 16 + 4g + q, columns col0 + 4p .. + 3 (lane = 16g + 4q + p)
  const int tr_off = (4 * g + (li >> 2));  // image row of the first read; + 16 rows for the second
  auto frag_tr = [&](const uint8_t* img, int ld, int col0) -> bf16x8 {
    const uint8_t* a0 = img + tr_off * ld + (col0 + 4 * (li & 3)) * 2;
    const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
    const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0 + 16 * ld));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 z = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    return __builtin_bit_cast(bf16x8, z);
  };
  // row fragment of an [m][k] image: 8 consecutive k (16 bytes) of row `row`
  auto frag_row = [&](const uint8_t* img, int row) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(img + row * T::LDA + g * 16));
  };

  f32x4 acc[T::TM][T::TN];
  f32x4 accs[T::PARTS == 3 ? T::TM : 1][T::PARTS == 3 ? T::TN : 1];  // (three parts: the five small terms)
#pragma unroll
  for (int t = 0; t < T::TM; ++t)
#pragma unroll
    for (int u = 0; u < T::TN; ++u) {
      acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (T::PARTS == 3) accs[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  if (c0 < c1) gload(c0, S0{});
  if (c0 + 1 < c1) gload(c0 + 1, S1{});
  if (c0 + 2 < c1) gload(c0 + 2, S2{});
  if (c0 + 3 < c1) gload(c0 + 3, S3{});
  if (c0 < c1) sstore(0, S0{});
  __syncthreads();
  // one chunk: LDS[buf] holds chunk `cur` (stored a step ago from set `cs`, which is free now): refill that set with
  // chunk cur + D, multiply chunk cur, store chunk cur + 1 (set `ns`, loaded D - 1 steps ago) into the other buffer
  auto step = [&](int cur, auto cs, auto ns) {
    const int buf = (cur - c0) & 1;
    if (cur + D < c1) gload(cur + D, cs);
    const uint8_t* base = smem + buf * T::BUF;
    const uint8_t* bb = base + T::PARTS * T::A_HALF;
    bf16x8 bp[T::TN][T::PARTS];  // [..][0] = hi ... [..][PARTS - 1] = lo
#pragma unroll
    for (int u = 0; u < T::TN; ++u) {
      const int col0 = (wn * T::TN + u) * 16;
#pragma unroll
      for (int q = 0; q < T::PARTS; ++q) bp[u][q] = frag_tr(bb + q * T::B_HALF, T::LDB, col0);
    }
#pragma unroll
    for (int t = 0; t < T::TM; ++t) {
      const int row0 = (wm * T::TM + t) * 16;
      bf16x8 ap[T::PARTS];
#pragma unroll
      for (int q = 0; q < T::PARTS; ++q) {
        if constexpr (T::AMC) ap[q] = frag_tr(base + q * T::A_HALF, T::LDA, row0);
        else ap[q] = frag_row(base + q * T::A_HALF, row0 + li);
      }
#pragma unroll
      for (int u = 0; u < T::TN; ++u) {
        if constexpr (T::PARTS == 2) {
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[u][0], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[u][1], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[u][0], acc[t][u], 0, 0, 0);
        } else {  // (a part, b part) with i + j <= 2, smallest first; hi * hi alone in the main accumulator
          accs[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[2], bp[u][0], accs[t][u], 0, 0, 0);
          accs[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[u][2], accs[t][u], 0, 0, 0);
          accs[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[u][1], accs[t][u], 0, 0, 0);
          accs[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], bp[u][0], accs[t][u], 0, 0, 0);
          accs[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[u][1], accs[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], bp[u][0], acc[t][u], 0, 0, 0);
        }
      }
    }
    if (cur + 1 < c1) sstore(buf ^ 1, ns);
    __syncthreads();
  };
  for (int base = c0; base < c1; base += D) {  // (the conditions below are block-uniform)
    step(base, S0{}, S1{});
    if (base + 1 < c1) step(base + 1, S1{}, S2{});
    if (base + 2 < c1) step(base + 2, S2{}, S3{});
    if (base + 3 < c1) step(base + 3, S3{}, S0{});
  }

#pragma unroll
  for (int t = 0; t < T::TM; ++t)
#pragma unroll
    for (int u = 0; u < T::TN; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + (wm * T::TM + t) * 16 + g * 4 + r;
        const int n = n0 + (wn * T::TN + u) * 16 + li;
        if (m < p.M && n < p.N) {
          if constexpr (T::PARTS == 3) p.store(bz, m, n, acc[t][u][r] + accs[t][u][r]);
          else p.store(bz, m, n, acc[t][u][r]);
        }
      }
}

// same contract as gemm::launch_gemm (the Problem's kslice is counted in chunks of BK = 32 here as well)
template <class T, class P>
int launch_gemm(P p, int splits, hipStream_t s, const char* name) {
  static const hipError_t attr_set = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3<T, P>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
  RELA_HIP(attr_set);
  const int nch = ceil_div(p.K, BK);
  p.kslice = ceil_div(nch, splits);
  ProfScope prof(name, s);
  note_launch("gemm_bf16x3");
  constexpr bool xmap = true;  // XCD-aware 1-D block maps (bit-identical to the plain grid, -10 % on the weight gradients)
  GridMap gm{ceil_div(p.N, T::BN), ceil_div(p.M, T::BM), splits, 0};
  dim3 grid(gm.gx, gm.gy, gm.gz);
  if (xmap && gm.gz >= 8) {
    gm.mode = 1;
    grid = dim3(8 * gm.gx * gm.gy * ceil_div(gm.gz, 8));
  } else if (xmap && gm.gz == 1 && gm.gy > 1 && gm.gx >= 8) {
    gm.mode = 2;
    grid = dim3(8 * gm.gy * ceil_div(gm.gx, 8));
  }
  hipLaunchKernelGGL((gemm_bf16x3<T, P>), grid, dim3(kT), T::LDS_BYTES, s, p, gm);
  return RELA_OK;
}

}  // namespace gemm3
}  // namespace rela_amd
