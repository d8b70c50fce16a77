// agent_ops.hip -- the small per-env reductions around the network on the actor path.
//
//   rela_nstep_return     MultiStepTransitionBuffer::popTransition  rela/dqn_actor.h:58-106
//   rela_apex_act_from_q  ApexAgent.greedy_act + act                pyrela/apex.py:48-65
//   rela_apex_td_from_q   ApexAgent.td_err + compute_priority       pyrela/apex.py:30-45,68-78
//
// All three are tiny (K <= a few thousand rows of <= 18 floats): one workgroup each, so the
// batch-global q.min() of greedy_act (apex.py:51) needs no second launch.
// HBM-bound in principle, launch-latency-bound in practice.
#include "common.h"
#include "prof.h"

namespace rela_amd {
namespace {

constexpr int kT = 1024;

// ---- n-step --------------------------------------------------------------------------
__global__ void nstep_kernel(int n, int K, float gamma, int first, const float* __restrict__ rh,
                             const uint8_t* __restrict__ th, float* __restrict__ out_r,
                             float* __restrict__ out_b, uint8_t* __restrict__ out_t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K) return;
  float bootstrap = 1.0f;
  int next_idx = n;
  for (int step = 0; step < n; ++step) {
    if (th[((first + step) % (n + 1)) * K + i]) {  // dqn_actor.h:75-80
      bootstrap = 0.0f;
      next_idx = step;
      break;
    }
  }
  const int initial = (bootstrap != 0.0f) ? n - 1 : next_idx;  // :93
  float acc = 0.0f;
  for (int step = initial; step >= 0; --step) {
    const float prod = __fmul_rn(gamma, acc);  // un-fused, SURVEY H8
    acc = __fadd_rn(rh[((first + step) % (n + 1)) * K + i], prod);  // :96
  }
  out_r[i] = acc;
  out_b[i] = bootstrap;
  out_t[i] = th[(first % (n + 1)) * K + i];  // terminal of step 0 only :66
}

// ---- block-wide min over a [n*A] table -------------------------------------------------
__device__ float block_min(const float* __restrict__ q, int total, float* red) {
  float m = INFINITY;
  for (int i = threadIdx.x; i < total; i += blockDim.x) m = fminf(m, q[i]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int off = blockDim.x >> 1; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] = fminf(red[threadIdx.x], red[threadIdx.x + off]);
    __syncthreads();
  }
  m = red[0];
  __syncthreads();
  return m;
}

// greedy over one row: argmax_j (1 + q - qmin) * legal, first maximal index (torch argmax)
__device__ __forceinline__ int greedy_row(const float* __restrict__ q, const float* __restrict__ legal, int A,
                                          float qmin) {
  int best = 0;
  float bv = -INFINITY;
  for (int j = 0; j < A; ++j) {
    const float lq = __fmul_rn(__fsub_rn(__fadd_rn(1.0f, q[j]), qmin), legal[j]);  // apex.py:51
    if (lq > bv) {
      bv = lq;
      best = j;
    }
  }
  return best;
}

// Philox4x32-10 (Salmon et al., SC'11) -- counter-based, so a row's draw does not depend on
// how rows are batched into launches.
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                             uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c0 = n0;
  c1 = n1;
  c2 = n2;
  c3 = n3;
}

__device__ __forceinline__ void philox(uint64_t seed, uint64_t ctr, uint32_t out[4]) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// one workgroup per group of `group` rows (= one reference TorchScript call)
__global__ __launch_bounds__(kT) void act_kernel(int n, int A, int group, const float* __restrict__ q,
                                                 const float* __restrict__ legal, const float* __restrict__ eps,
                                                 uint64_t seed, uint64_t offset, int64_t* __restrict__ action) {
  __shared__ float red[kT];
  const int r0 = blockIdx.x * group;
  const int r1 = min(n, r0 + group);
  const float qmin = block_min(q + (size_t)r0 * A, (r1 - r0) * A, red);
  for (int i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
    const int g = greedy_row(q + (size_t)i * A, legal + (size_t)i * A, A, qmin);
    int a = g;
    const float e = eps ? eps[i] : 0.0f;
    if (e > 0.0f) {
      uint32_t rnd[4];
      philox(seed, offset + (uint64_t)i, rnd);
      const float u = (float)(rnd[0] >> 8) * (1.0f / 16777216.0f);  // torch.rand: 24-bit uniform [0,1)
      if (u < e) {  // rand < eps -> random legal action (multinomial over the 0/1 mask) apex.py:61-64
        int nl = 0;
        for (int j = 0; j < A; ++j) nl += legal[(size_t)i * A + j] > 0.0f;
        if (nl > 0) {
          int pick = (int)(((uint64_t)rnd[1] * (uint64_t)nl) >> 32);
          for (int j = 0; j < A; ++j) {
            if (legal[(size_t)i * A + j] > 0.0f) {
              if (pick == 0) {
                a = j;
                break;
              }
              --pick;
            }
          }
        }
      }
    }
    action[i] = a;
  }
}

__global__ __launch_bounds__(kT) void td_kernel(int n, int A, int group, const float* __restrict__ q,
                                                const float* __restrict__ qno, const float* __restrict__ qnt,
                                                const float* __restrict__ nlegal, const int64_t* __restrict__ action,
                                                const float* __restrict__ reward, const float* __restrict__ bootstrap,
                                                float gamma_n, float* __restrict__ td, float* __restrict__ prio) {
  __shared__ float red[kT];
  const int r0 = blockIdx.x * group;
  const int r1 = min(n, r0 + group);
  const float qmin = block_min(qno + (size_t)r0 * A, (r1 - r0) * A, red);  // greedy_act(next_obs) apex.py:41,51
  for (int i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
    const int na = greedy_row(qno + (size_t)i * A, nlegal + (size_t)i * A, A, qmin);
    const float qa = q[(size_t)i * A + (int)action[i]];   // :39
    const float bq = qnt[(size_t)i * A + na];              // :43
    const float g = __fmul_rn(bootstrap[i], gamma_n);      // bootstrap * (gamma ** n) * q, left to right :44
    const float tgt = __fadd_rn(reward[i], __fmul_rn(g, bq));
    const float e = __fsub_rn(tgt, qa);                    // :45
    if (td) td[i] = e;
    if (prio) prio[i] = fabsf(e);                          // :78
  }
}

}  // namespace
}  // namespace rela_amd

using namespace rela_amd;

extern "C" int rela_nstep_return(int multi_step, int K, float gamma, int first_row, const float* reward_hist_dev,
                                 const uint8_t* terminal_hist_dev, float* out_reward_dev, float* out_bootstrap_dev,
                                 uint8_t* out_terminal_dev, void* stream) {
  RELA_CHECK(multi_step >= 1 && K >= 1 && first_row >= 0 && first_row <= multi_step && reward_hist_dev && terminal_hist_dev && out_reward_dev &&
                 out_bootstrap_dev && out_terminal_dev,
             RELA_EINVAL, "rela_nstep_return: bad arguments");
  ProfScope prof("nstep_kernel", (hipStream_t)stream);
  hipLaunchKernelGGL(nstep_kernel, dim3(ceil_div(K, 256)), dim3(256), 0, (hipStream_t)stream, multi_step, K, gamma,
                     first_row, reward_hist_dev, terminal_hist_dev, out_reward_dev, out_bootstrap_dev, out_terminal_dev);
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

extern "C" int rela_apex_act_from_q(int n, int num_action, int group_rows, const float* q_dev,
                                    const float* legal_dev, const float* eps_dev, uint64_t rng_seed,
                                    uint64_t rng_offset, int64_t* action_dev, void* stream) {
  RELA_CHECK(n >= 1 && num_action >= 1 && group_rows >= 0 && q_dev && legal_dev && action_dev, RELA_EINVAL,
             "rela_apex_act_from_q: bad arguments");
  const int group = group_rows > 0 ? group_rows : n;
  const int threads = group >= 512 ? kT : 256;
  ProfScope prof("act_kernel", (hipStream_t)stream);
  hipLaunchKernelGGL(act_kernel, dim3(ceil_div(n, group)), dim3(threads), 0, (hipStream_t)stream, n, num_action, group,
                     q_dev, legal_dev, eps_dev, rng_seed, rng_offset, action_dev);
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

extern "C" int rela_apex_td_from_q(int n, int num_action, int group_rows, const float* q_dev,
                                   const float* q_next_online_dev, const float* q_next_target_dev,
                                   const float* next_legal_dev,
                                   const int64_t* action_dev, const float* reward_dev, const float* bootstrap_dev,
                                   float gamma_n, float* td_err_dev, float* priority_dev, void* stream) {
  RELA_CHECK(n >= 1 && num_action >= 1 && group_rows >= 0 && q_dev && q_next_online_dev && q_next_target_dev &&
                 next_legal_dev && action_dev && reward_dev && bootstrap_dev,
             RELA_EINVAL, "rela_apex_td_from_q: bad arguments");
  const int group = group_rows > 0 ? group_rows : n;
  const int threads = group >= 512 ? kT : 256;
  ProfScope prof("td_kernel", (hipStream_t)stream);
  hipLaunchKernelGGL(td_kernel, dim3(ceil_div(n, group)), dim3(threads), 0, (hipStream_t)stream, n, num_action, group,
                     q_dev, q_next_online_dev,
                     q_next_target_dev, next_legal_dev, action_dev, reward_dev, bootstrap_dev, gamma_n, td_err_dev,
                     priority_dev);
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Sliding frame stacks on the device (GameState::computeFeature, atari/game_state.h:53-82: a step shifts the stack by
// one 84x84 plane and appends the new frame; the first frame of an episode is repeated four times).  The env layer
// uploads only the NEWEST plane of every row into a contiguous staging array ([rows][7056]: 7,056 B instead of
// 28,224 B across PCIe, one plain 1-D copy per actor thread); this kernel writes the whole stack of the observation
// slot: plane 3 = the new plane, plane k < 3 = the previous observation's plane k + 1, or the new plane again where
// restart[row] == 1 (episode start); rows flagged 2 were uploaded whole and are left alone.  HBM bound: 28 KB read +
// 28 KB written per row.
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void slide_stacks_kernel(uint8_t* __restrict__ cur, const uint8_t* __restrict__ prev,
                                                           const uint8_t* __restrict__ fresh,
                                                           const uint8_t* __restrict__ restart, int rows) {
  constexpr int kPlane16 = 84 * 84 / 16;  // 441 sixteen-byte units per plane
  const int row = blockIdx.x, k = blockIdx.y;  // k = destination plane 0..3
  if (row >= rows || restart[row] == 2) return;
  uint4* dst = reinterpret_cast<uint4*>(cur + ((size_t)row * 4 + k) * 7056);
  const uint4* src = (k == 3 || restart[row] == 1) ? reinterpret_cast<const uint4*>(fresh + (size_t)row * 7056)
                                                   : reinterpret_cast<const uint4*>(prev + ((size_t)row * 4 + k + 1) * 7056);
  for (int i = threadIdx.x; i < kPlane16; i += 256) dst[i] = src[i];
}
}  // namespace

namespace rela_amd {
int slide_stacks(uint8_t* cur_slot, const uint8_t* prev_slot, const uint8_t* fresh_planes, const uint8_t* restart_dev, int rows,
                 hipStream_t s) {
  ProfScope prof("slide_stacks", s);
  note_launch("slide_stacks");
  hipLaunchKernelGGL(slide_stacks_kernel, dim3(rows, 4), dim3(256), 0, s, cur_slot, prev_slot, fresh_planes, restart_dev, rows);
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}
}  // namespace rela_amd
