// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 under the accumulation patterns the forward kernels use.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
  bf16x8 a, b0, b1;
  for (int i = 0; i < 8; ++i) a[i] = (__bf16)(threadIdx.x * 0.001f + i), b0[i] = (__bf16)(i * 0.5f), b1[i] = (__bf16)(i * 0.25f);
  f32x4 acc[6];
  for (int t = 0; t < 6; ++t) acc[t] = f32x4{0, 0, 0, 0};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // three dependent MFMAs per accumulator, accumulators in sequence (conv_bf16s order)
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[t], 0, 0, 0);
      }
    } else if (MODE == 1) {  // the same 18 MFMAs, round-robin over the accumulators (no back-to-back dependence)
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, p == 1 ? b1 : b0, acc[t], 0, 0, 0);
    } else {  // pairs interleaved (conv1 order)
#pragma unroll
      for (int t = 0; t < 6; t += 2)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, p == 1 ? b1 : b0, acc[t], 0, 0, 0);
          acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, p == 1 ? b1 : b0, acc[t + 1], 0, 0, 0);
        }
    }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int t = 0; t < 6; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(int threads, const char* name) {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mfma_per_simd = 18.0 * iters * (threads / 64) / 4.0;
  printf("%-28s waves/SIMD %d: %.1f us, %.2f shader-clock ticks per MFMA per SIMD (clock64), %.1f ns per MFMA per SIMD, %.0f TFLOP/s\n",
         name, threads / 256, ms * 1e3, (double)c / mfma_per_simd, ms * 1e6 / mfma_per_simd,
         mfma_per_simd * 1024 * 16384.0 / (ms * 1e-3) / 1e12);
}

int main() {
  for (int threads : {256, 512}) {
    run<0>(threads, "3 dependent per acc");
    run<1>(threads, "round-robin 6 acc");
    run<2>(threads, "pairs interleaved");
  }
  return 0;
}
