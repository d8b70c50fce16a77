// Micro-benchmark: v_mfma_f32_16x16x32_bf16 fed the way conv_bf16s feeds it -- A fragments (hi, lo) from LDS through a
// register ring, B fragments resident in registers -- with and without the LDS reads, for 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds_feed mfma_lds_feed.hip && ./mfma_lds_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: operands loop-invariant registers (pure MFMA)   1: A (hi, lo) from LDS, ring depth D, 3 MFMAs per pair
// MODE 2: as 1 but ONE b128 read per 3 MFMAs (half the LDS traffic)   3: as 1 but 6 MFMAs per pair (two column tiles)
template <int MODE, int D>
__global__ __launch_bounds__(512) void k(float* out, int iters, const uint4* wsrc) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  for (int i = tid; i < 59200 / 16; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = make_uint4(i, i * 3, i * 7, 0x3f803f80u);
  __syncthreads();
  constexpr int KS = 16, RT = 3;
  bf16x8 bh[KS], bl[KS];
  for (int ks = 0; ks < KS; ++ks) {
    bh[ks] = __builtin_bit_cast(bf16x8, wsrc[(ks * 2) * 64 + lane]);
    bl[ks] = __builtin_bit_cast(bf16x8, wsrc[(ks * 2 + 1) * 64 + lane]);
  }
  int abase[RT];
  for (int t = 0; t < RT; ++t) {
    const int m = (t * 2) * 16 + li, mm = m < 81 ? m : 0, oy = mm / 9, ox = mm % 9;
    abase[t] = (oy * 2 * 185 + ox * 2 * 9 + g) * 16;
  }
  f32x4 acc[RT][2];
  for (int t = 0; t < RT; ++t) acc[t][0] = acc[t][1] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    constexpr int TOT = KS * RT;
    uint4 ah[D], al[D];
    auto a_issue = [&](int idx, int slot) {
      const int ks = idx / RT, t = idx - ks * RT, kh = ks / 4, kw = ks - kh * 4;
      const uint8_t* ap = smem + abase[t] + (kh * 185 + kw * 9) * 16;
      ah[slot] = *reinterpret_cast<const uint4*>(ap);
      if (MODE != 2) al[slot] = *reinterpret_cast<const uint4*>(ap + 64);
    };
    if (MODE != 0) {
#pragma unroll
      for (int i = 0; i < D; ++i) a_issue(i, i);
    } else {
      for (int i = 0; i < D; ++i) ah[i] = make_uint4(it, 1, 2, 3), al[i] = make_uint4(4, 5, 6, it);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int idx = 0; idx < TOT; ++idx) {
      const int ks = idx / RT, t = idx - ks * RT, slot = idx % D;
      const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[slot]);
      const bf16x8 xl = __builtin_bit_cast(bf16x8, MODE == 2 ? ah[slot] : al[slot]);
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[ks], xl, acc[t][0], 0, 0, 0);
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[ks], xh, acc[t][0], 0, 0, 0);
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[ks], xh, acc[t][0], 0, 0, 0);
      if (MODE == 3) {
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[ks], xl, acc[t][1], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[(ks + 1) % KS], xh, acc[t][1], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[(ks + 1) % KS], xh, acc[t][1], 0, 0, 0);
      }
      if (MODE != 0 && idx + D < TOT) a_issue(idx + D, slot);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int t = 0; t < RT; ++t) s += acc[t][0][0] + acc[t][0][3] + acc[t][1][1];
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <int MODE, int D>
void run(int threads, const char* name) {
  float* out;
  uint4* w;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&w, 32 * 64 * 16);
  (void)hipMemset(w, 0x3f, 32 * 64 * 16);
  const int iters = 400;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, D>), dim3(256), dim3(threads), 65536, 0, out, iters, w);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, D>), dim3(256), dim3(threads), 65536, 0, out, iters, w);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = (MODE == 3 ? 6.0 : 3.0) * 48 * iters;
  const double per_simd = per_wave * (threads / 64) / 4.0;
  printf("%-44s waves/SIMD %d: %7.1f us  %.2f ns per MFMA per SIMD\n", name, threads / 256, ms * 1e3, ms * 1e6 / per_simd);
}

int main() {
  for (int threads : {256, 512}) {
    run<0, 4>(threads, "registers only");
    run<1, 2>(threads, "A hi+lo from LDS, ring 2");
    run<1, 4>(threads, "A hi+lo from LDS, ring 4");
    run<1, 8>(threads, "A hi+lo from LDS, ring 8");
    run<2, 4>(threads, "A one b128 per triple, ring 4");
    run<3, 4>(threads, "A hi+lo from LDS, 6 MFMAs per pair, ring 4");
  }
  return 0;
}
