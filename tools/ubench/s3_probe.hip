// s3_probe.hip -- standalone check + timing of rela_amd/csrc/gemm_s3.h (f32x3 contractions over split3 records).
//   hipcc --offload-arch=gfx950 -O3 -I rela_amd/csrc tools/ubench/s3_probe.hip -o tools/ubench/s3_probe
//   tools/ubench/s3_probe [N = 6554] [iters = 20]
// For conv2 / conv3 / fc of the AtariFFNet trunk at N samples: runs the 6- and 9-product kernels on random ReLU-like
// activations, compares sampled outputs with an f64 evaluation on the host and with a sequential f32 FMA chain (what
// "f32 arithmetic" means for one dot product), and times the kernels with HIP events.  One JSON object per line.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

namespace rela_amd {
void set_last_error(const char*, ...) {}
}  // namespace rela_amd
#include "gemm_s3.h"
#include "conv_img_s3.h"
#include "conv12_s3.h"
#ifndef F32EMU_OCC
#define F32EMU_OCC 2
#endif

using namespace rela_amd::f32emu;
namespace s3 = rela_amd::s3;

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

template <class P, class PS, int CIN>
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
  float *dX, *dW, *dB, *dO, *dU;
  uint8_t *dR, *dOR;
  uint4* dP;
  CK(hipMalloc(&dX, xe * 4));
  CK(hipMalloc(&dR, xe * 6 + 4096));
  CK(hipMalloc(&dW, we * 4));
  CK(hipMalloc(&dB, hp.OC * 4));
  CK(hipMalloc(&dO, oe * 4));
  CK(hipMalloc(&dU, oe * 4));
  CK(hipMalloc(&dOR, oe * 6));
  CK(hipMalloc(&dP, packed_u4<P>() * 16));
  CK(hipMemcpy(dX, X.data(), xe * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, W.data(), we * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), hp.OC * 4, hipMemcpyHostToDevice));
  const int64_t pel = packed_u4<P>() * 8 / 3;  // one thread per (cg, ks, u, lane, j)
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((pel + 255) / 256)), dim3(256), 0, 0, hp.mode, dW,
                     reinterpret_cast<uint16_t*>(dP), P::NCG, P::KS);
  {
    const int64_t px = (int64_t)xe / CIN, th = px * (CIN / 4);
    hipLaunchKernelGGL((s3::split_s3<CIN>), dim3((unsigned)((th + 255) / 256)), dim3(256), 0, 0, dX, dR, px);
    // round trip: records -> f32 must give the input back bit for bit
    hipLaunchKernelGGL((s3::unsplit_s3<CIN>), dim3((unsigned)((th + 255) / 256)), dim3(256), 0, 0, dR, dX, px);
    std::vector<float> X2(xe);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(X2.data(), dX, xe * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < xe; ++i) bad += memcmp(&X2[i], &X[i], 4) != 0;
    printf("{\"layer\": \"%s\", \"split_roundtrip_mismatches\": %zu, \"of\": %zu}\n", hp.name, bad, xe);
  }
  CK(hipDeviceSynchronize());

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
  for (int outs3 : {0, 1, 2}) {
    if (outs3 && hp.OC % 64) continue;
    if (outs3 == 2 && hp.mode != 2) continue;  // the image kernel: conv3 only
    if (outs3 != 2 && hp.mode == 2) continue;  // (the GEMM form owns 128 columns per block: not for conv3)
    auto go = [&]() {
      if (outs3 == 2)
        s3::launch_conv3_img(dR, dP, dB, dOR, N, 0);
      else if constexpr (PS::OC % 128 == 0) {
        if (outs3)
          s3::launch<PS, s3::kEpiReluS3>(dR, dP, dB, dOR, M, 0);
        else
          s3::launch<PS, s3::kEpiRelu>(dR, dP, dB, dO, M, 0);
      }
    };
    CK(hipMemset(dO, 0xff, oe * 4));
    CK(hipMemset(dOR, 0xff, oe * 6));
    go();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    if (outs3) {
      // records of OC channels per row -> f32 (unwritten 0xffff parts are NaNs)
      const int64_t th = (int64_t)M * (hp.OC / 4);
      if (hp.OC == 64)
        hipLaunchKernelGGL((s3::unsplit_s3<64>), dim3((unsigned)((th + 255) / 256)), dim3(256), 0, 0, dOR, dU, (int64_t)M);
      else
        hipLaunchKernelGGL((s3::unsplit_s3<512>), dim3((unsigned)((th + 255) / 256)), dim3(256), 0, 0, dOR, dU, (int64_t)M);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(O.data(), dU, oe * 4, hipMemcpyDeviceToHost));
    } else {
      CK(hipMemcpy(O.data(), dO, oe * 4, hipMemcpyDeviceToHost));
    }
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
    if constexpr (PS::OC % 128 == 0) {  // the split-K form of the same launch (raw sums, no reduce): timing only
      if (outs3 == 0) {
        const s3::Plan pl = s3::plan<PS>(M, true);
        if (pl.slices > 1) {
          float* part;
          CK(hipMalloc(&part, (size_t)pl.slices * oe * 4));
          for (int i = 0; i < 3; ++i) s3::launch<PS, s3::kEpiRaw>(dR, dP, dB, part, M, 0, pl);
          CK(hipEventRecord(e0, 0));
          for (int i = 0; i < iters; ++i) s3::launch<PS, s3::kEpiRaw>(dR, dP, dB, part, M, 0, pl);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms2 = 0;
          CK(hipEventElapsedTime(&ms2, e0, e1));
          printf("{\"layer\": \"%s\", \"N\": %d, \"splitk\": [%d, %d], \"us\": %.1f}\n", hp.name, N, pl.nrb, pl.slices, 1e3 * ms2 / iters);
          CK(hipFree(part));
        }
      }
    }
    printf("{\"layer\": \"%s\", \"N\": %d, \"out_s3\": %d, \"us\": %.1f, \"f32_equiv_tflops\": %.1f, "
           "\"bf16_mfma_tflops\": %.1f, \"max_err_vs_f64\": %.3e, \"mean_err_vs_f64\": %.3e, "
           "\"f32_fma_chain_max_err\": %.3e, \"f32_fma_chain_mean_err\": %.3e, \"mean_abs_preact\": %.3e, "
           "\"unwritten\": %zu, \"checked\": %zu}\n",
           hp.name, N, outs3, us, flop / us * 1e-6, flop * 6 / us * 1e-6, mx, mean, cmx, cmean, sum_abs / ref.size(), nan,
           ref.size());
    fflush(stdout);
  }
  CK(hipFree(dX));
  CK(hipFree(dR));
  CK(hipFree(dW));
  CK(hipFree(dB));
  CK(hipFree(dO));
  CK(hipFree(dU));
  CK(hipFree(dOR));
  CK(hipFree(dP));
}

// timing only (random digits / weights / frames: the values do not matter for the clock): conv1 -> conv2 fused
static void time_conv12(int N, int iters) {
  using F = s3::Conv12S;
  uint8_t *in, *out;
  uint4 *W1d, *B2;
  float *sc, *b1, *b2;
  const size_t in_bytes = (size_t)N * F::IN_ELEMS, out_bytes = (size_t)N * 81 * 384;
  CK(hipMalloc(&in, in_bytes));
  CK(hipMalloc(&out, out_bytes));
  CK(hipMalloc(&W1d, 3 * 2 * 4 * 64 * 16));
  CK(hipMalloc(&B2, 16 * 4 * 3 * 64 * 16));
  CK(hipMalloc(&sc, 64 * 4));
  CK(hipMalloc(&b1, 64 * 4));
  CK(hipMalloc(&b2, 64 * 4));
  std::mt19937 rng(7);
  std::vector<uint8_t> h(in_bytes);
  for (auto& v : h) v = (uint8_t)rng();
  CK(hipMemcpy(in, h.data(), in_bytes, hipMemcpyHostToDevice));
  std::vector<uint8_t> w(3 * 2 * 4 * 64 * 16);
  for (auto& v : w) v = (uint8_t)(rng() % 200);
  CK(hipMemcpy(W1d, w.data(), w.size(), hipMemcpyHostToDevice));
  std::vector<uint16_t> wb(16 * 4 * 3 * 64 * 8);
  for (auto& v : wb) v = (uint16_t)(0x3c00 + (rng() & 0xff));  // small positive bf16
  CK(hipMemcpy(B2, wb.data(), wb.size() * 2, hipMemcpyHostToDevice));
  std::vector<float> f(64, 1e-9f);
  CK(hipMemcpy(sc, f.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(b1, f.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(b2, f.data(), 256, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&s3::conv12_s3<false>), hipFuncAttributeMaxDynamicSharedMemorySize, F::LDS_TOTAL));
  auto go = [&]() {
    hipLaunchKernelGGL(s3::conv12_s3<false>, dim3(std::min(256, N)), dim3(F::kT), F::LDS_TOTAL, 0, in, W1d, sc, b1, B2, b2, out,
                       (float*)nullptr, N);
  };
  for (int i = 0; i < 3; ++i) go();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) go();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"layer\": \"conv12\", \"N\": %d, \"us\": %.1f}\n", N, 1e3 * ms / iters);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 6554;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  if (argc > 3 && atoi(argv[3]) == 12) {
    time_conv12(N, iters);
    return 0;
  }
  const HostProb p2{"conv2", 1, 512, 64, 81, 12800, x2, w2};
  const HostProb p3{"conv3", 2, 576, 64, 49, 5184, x3, w3};
  const HostProb pf{"fc", 3, 3136, 512, 1, 3136, xf, wf};
  run<ProbConv3, s3::ProbConv3, 64>(p3, N, iters);
  run<ProbFc, s3::ProbFc, 64>(pf, N, iters);
  return 0;
}
