// Micro-benchmark of the conv1 inner loop of the fused kernel: A fragments as two ds_read_b64 from the bf16 image,
// weight fragments (2 pieces) from LDS per k-step or resident in registers, RPW row tiles per wave.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_conv1_feed mfma_conv1_feed.hip && ./mfma_conv1_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PLANE = 7552, KS = 8;

template <int RPW, int BREG, int D>
__global__ __launch_bounds__(512) void k(float* out, int iters, const uint4* wsrc) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4, wave = tid >> 6;
  uint4* b1s = reinterpret_cast<uint4*>(smem + 4 * PLANE);
  for (int i = tid; i < 4 * PLANE / 16; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3f803f80u, i, 0x40004000u, 7);
  for (int i = tid; i < 2 * 2 * KS * 64; i += blockDim.x) b1s[i] = wsrc[i];
  __syncthreads();
  const int ct = wave & 1, rg = wave >> 1;
  int abase[RPW], abase2[RPW];
  for (int t = 0; t < RPW; ++t) {
    const int m = min((rg + t * 4) * 16 + li, 199), oy = m / 20, ox = m % 20;
    abase[t] = g * PLANE + (4 * oy * 84 + 4 * ox) * 2;
    abase2[t] = abase[t] + 8;
    asm volatile("" : "+v"(abase2[t]));
  }
  bf16x8 breg[2][KS];
  if (BREG)
    for (int p = 0; p < 2; ++p)
      for (int ks = 0; ks < KS; ++ks) breg[p][ks] = __builtin_bit_cast(bf16x8, wsrc[((p * 2 + ct) * KS + ks) * 64 + lane]);
  f32x4 acc[RPW];
  for (int t = 0; t < RPW; ++t) acc[t] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    constexpr int TOT = KS * RPW;
    uint2 a0[D], a1[D];
    auto a_issue = [&](int idx, int slot) {
      const int ks = idx / RPW, t = idx - ks * RPW;
      a0[slot] = *reinterpret_cast<const uint2*>(smem + abase[t] + ks * 168);
      a1[slot] = *reinterpret_cast<const uint2*>(smem + abase2[t] + ks * 168);
    };
    uint4 wlo[2], whi[2];
    auto w_issue = [&](int ks, int slot) {
      whi[slot] = b1s[((0 * 2 + ct) * KS + ks) * 64 + lane];
      wlo[slot] = b1s[((1 * 2 + ct) * KS + ks) * 64 + lane];
    };
    if (!BREG) w_issue(0, 0);
#pragma unroll
    for (int i = 0; i < D; ++i) a_issue(i, i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (!BREG && ks + 1 < KS) w_issue(ks + 1, (ks + 1) & 1);
      const bf16x8 blo = BREG ? breg[1][ks] : __builtin_bit_cast(bf16x8, wlo[ks & 1]);
      const bf16x8 bhi = BREG ? breg[0][ks] : __builtin_bit_cast(bf16x8, whi[ks & 1]);
#pragma unroll
      for (int t = 0; t < RPW; ++t) {
        const int idx = ks * RPW + t, s0 = idx % D;
        const bf16x8 x0 = __builtin_bit_cast(bf16x8, make_uint4(a0[s0].x, a0[s0].y, a1[s0].x, a1[s0].y));
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blo, x0, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bhi, x0, acc[t], 0, 0, 0);
        if (idx + D < TOT) a_issue(idx + D, s0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
  for (int t = 0; t < RPW; ++t) s += acc[t][0] + acc[t][3];
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <int RPW, int BREG, int D>
void run(int threads, const char* name) {
  float* out;
  uint4* w;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&w, 2 * 2 * KS * 64 * 16);
  (void)hipMemset(w, 0x3f, 2 * 2 * KS * 64 * 16);
  const int iters = 1000, lds = 4 * PLANE + 2 * 2 * KS * 64 * 16;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<RPW, BREG, D>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<RPW, BREG, D>), dim3(256), dim3(threads), lds, 0, out, iters, w);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<RPW, BREG, D>), dim3(256), dim3(threads), lds, 0, out, iters, w);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = 2.0 * KS * RPW * iters * (threads / 64) / 4.0;
  printf("%-40s waves/SIMD %d: %7.1f us  %.2f ns per MFMA per SIMD\n", name, threads / 256, ms * 1e3, ms * 1e6 / per_simd);
}

int main() {
  for (int threads : {256, 512}) {
    run<4, 0, 4>(threads, "4 tiles, weights from LDS, ring 4");
    run<4, 1, 4>(threads, "4 tiles, weights in registers, ring 4");
    run<7, 0, 4>(threads, "7 tiles, weights from LDS, ring 4");
    run<7, 1, 4>(threads, "7 tiles, weights in registers, ring 4");
    run<4, 0, 8>(threads, "4 tiles, weights from LDS, ring 8");
  }
  return 0;
}
