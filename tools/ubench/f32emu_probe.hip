// f32emu_probe.hip -- standalone check + timing of rela_amd/csrc/gemm_f32emu.h (no torch, no library).
//   hipcc --offload-arch=gfx950 -O3 -I rela_amd/csrc tools/ubench/f32emu_probe.hip -o tools/ubench/f32emu_probe
//   tools/ubench/f32emu_probe [N = 6554] [iters = 20]
// For conv2 / conv3 / fc of the AtariFFNet trunk at N samples: runs the 6- and 9-product kernels on random ReLU-like
// activations, compares sampled outputs with an f64 evaluation on the host and with a sequential f32 FMA chain (what
// "f32 arithmetic" means for one dot product), and times the kernels with HIP events.  One JSON object per line.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

namespace rela_amd {
void set_last_error(const char*, ...) {}
}  // namespace rela_amd
#include "gemm_f32emu.h"
#ifndef F32EMU_OCC
#define F32EMU_OCC 2
#endif

using namespace rela_amd::f32emu;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

__global__ void pack_kernel(int mode, const float* w, uint16_t* frag, int NCG, int KS) {
  pack_f32emu_at((int64_t)blockIdx.x * blockDim.x + threadIdx.x, mode, w, frag, NCG, KS);
}

struct HostProb {
  const char* name;
  int mode, K, OC, rows_per_sample, in_per_sample;
  // element offset of X(m, k) and index of W(k, n) in the state_dict layout
  int64_t (*xidx)(int m, int k);
  int64_t (*widx)(int k, int n);
};
static int64_t x2(int m, int k) {
  const int n = m / 81, pos = m % 81, oy = pos / 9, ox = pos % 9;
  const int c = k & 31, tap = k >> 5, kh = tap >> 2, kw = tap & 3;
  return (((int64_t)n * 20 + 2 * oy + kh) * 20 + 2 * ox + kw) * 32 + c;
}
static int64_t w2(int k, int n) {
  const int c = k & 31, tap = k >> 5;
  return ((n * 32 + c) * 4 + (tap >> 2)) * 4 + (tap & 3);
}
static int64_t x3(int m, int k) {
  const int n = m / 49, pos = m % 49, oy = pos / 7, ox = pos % 7;
  const int c = k & 63, tap = k >> 6, kh = tap / 3, kw = tap % 3;
  return (((int64_t)n * 9 + oy + kh) * 9 + ox + kw) * 64 + c;
}
static int64_t w3(int k, int n) {
  const int c = k & 63, tap = k >> 6;
  return ((n * 64 + c) * 3 + tap / 3) * 3 + tap % 3;
}
static int64_t xf(int m, int k) { return (int64_t)m * 3136 + k; }
static int64_t wf(int k, int n) {
  const int c = k & 63, pos = k >> 6;
  return (int64_t)n * 3136 + c * 49 + pos;
}

template <class P>
static void run(const HostProb& hp, int N, int iters) {
  const int M = N * hp.rows_per_sample;
  const size_t xe = (size_t)N * hp.in_per_sample, we = (size_t)hp.K * hp.OC, oe = (size_t)M * hp.OC;
  std::mt19937 rng(1234 + hp.mode);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> X(xe), W(we), B(hp.OC);
  for (auto& v : X) {
    const float t = nd(rng);
    v = t > 0.f ? t : 0.f;  // post-ReLU activations: half of them zero
  }
  const float ws = 1.0f / std::sqrt((float)hp.K);
  for (auto& v : W) v = nd(rng) * ws;
  for (auto& v : B) v = nd(rng) * 0.1f;
  float *dX, *dW, *dB, *dO;
  uint4* dP;
  CK(hipMalloc(&dX, xe * 4));
  CK(hipMalloc(&dW, we * 4));
  CK(hipMalloc(&dB, hp.OC * 4));
  CK(hipMalloc(&dO, oe * 4));
  CK(hipMalloc(&dP, packed_u4<P>() * 16));
  CK(hipMemcpy(dX, X.data(), xe * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, W.data(), we * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), hp.OC * 4, hipMemcpyHostToDevice));
  const int64_t pel = packed_u4<P>() * 8 / 3;  // one thread per (cg, ks, u, lane, j)
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((pel + 255) / 256)), dim3(256), 0, 0, hp.mode, dW,
                     reinterpret_cast<uint16_t*>(dP), P::NCG, P::KS);
  CK(hipDeviceSynchronize());

  // sample outputs: every row of the first and the last sample + random rows
  std::vector<int> rows;
  for (int r = 0; r < hp.rows_per_sample; ++r) rows.push_back(r), rows.push_back(M - 1 - r);
  std::uniform_int_distribution<int> ud(0, M - 1);
  for (int i = 0; i < 600; ++i) rows.push_back(ud(rng));
  std::vector<double> ref(rows.size() * hp.OC);
  std::vector<float> chain(rows.size() * hp.OC);
  double sum_abs = 0;
  for (size_t i = 0; i < rows.size(); ++i)
    for (int n = 0; n < hp.OC; ++n) {
      double a = B[n];
      float c = B[n];
      for (int k = 0; k < hp.K; ++k) {
        const float x = X[hp.xidx(rows[i], k)], w = W[hp.widx(k, n)];
        a += (double)x * (double)w;
        c = fmaf(x, w, c);
      }
      ref[i * hp.OC + n] = a > 0 ? a : 0;
      chain[i * hp.OC + n] = c > 0.f ? c : 0.f;
      sum_abs += std::fabs(a);
    }
  auto errs = [&](const std::vector<float>& got_rows, double& mx, double& mean) {
    mx = 0, mean = 0;
    for (size_t i = 0; i < ref.size(); ++i) {
      const double e = std::fabs((double)got_rows[i] - ref[i]);
      mx = std::max(mx, e), mean += e;
    }
    mean /= ref.size();
  };
  double cmx, cmean;
  errs(chain, cmx, cmean);

  std::vector<float> O(oe), got(ref.size());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int np : {6, 9}) {
    auto go = [&]() {
      if (np == 6)
        launch<P, 6, F32EMU_OCC>(dX, dP, dB, dO, M, 0);
      else
        launch<P, 9, F32EMU_OCC>(dX, dP, dB, dO, M, 0);
    };
    CK(hipMemset(dO, 0xff, oe * 4));
    go();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    CK(hipMemcpy(O.data(), dO, oe * 4, hipMemcpyDeviceToHost));
    size_t nan = 0;
    for (size_t i = 0; i < oe; ++i) nan += std::isnan(O[i]) ? 1 : 0;  // 0xff fill = NaN: an unwritten output
    for (size_t i = 0; i < rows.size(); ++i)
      for (int n = 0; n < hp.OC; ++n) got[i * hp.OC + n] = O[(size_t)rows[i] * hp.OC + n];
    double mx, mean;
    errs(got, mx, mean);
    for (int i = 0; i < 3; ++i) go();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) go();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters;
    const double flop = 2.0 * M * hp.K * hp.OC;
    printf("{\"layer\": \"%s\", \"N\": %d, \"products\": %d, \"us\": %.1f, \"f32_equiv_tflops\": %.1f, "
           "\"bf16_mfma_tflops\": %.1f, \"max_err_vs_f64\": %.3e, \"mean_err_vs_f64\": %.3e, "
           "\"f32_fma_chain_max_err\": %.3e, \"f32_fma_chain_mean_err\": %.3e, \"mean_abs_preact\": %.3e, "
           "\"unwritten\": %zu, \"checked\": %zu}\n",
           hp.name, N, np, us, flop / us * 1e-6, flop * np / us * 1e-6, mx, mean, cmx, cmean, sum_abs / ref.size(), nan,
           ref.size());
    fflush(stdout);
  }
  CK(hipFree(dX));
  CK(hipFree(dW));
  CK(hipFree(dB));
  CK(hipFree(dO));
  CK(hipFree(dP));
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 6554;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  const HostProb p2{"conv2", 1, 512, 64, 81, 12800, x2, w2};
  const HostProb p3{"conv3", 2, 576, 64, 49, 5184, x3, w3};
  const HostProb pf{"fc", 3, 3136, 512, 1, 3136, xf, wf};
  run<ProbConv2>(p2, N, iters);
  run<ProbConv3>(p3, N, iters);
  run<ProbFc>(pf, N, iters);
  return 0;
}
