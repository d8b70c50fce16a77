/* oracle/oracle.h -- CPU restatement of the reference hot path, in plain C.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load liboracle.so -- and only as the checker / the timed CPU
 * baseline, never as part of the product path (rela_amd/ never imports it).
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_golden.py)
 * against golden vectors produced here by the real reference compiled from
 * /root/reference (oracle/Makefile `ref`, harnesses in oracle/ref_harness/), committed
 * under tests/golden/, and against the reference's own two C++ tests
 * (rela/tests/test_concurrent_queue.cc, test_prioritized_replay.cc).
 * One documented limit: weights = pow(priority, alpha) is ATen's SLEEF powf in the
 * reference (rela/prioritized_replay.h:188,239); this file uses libm powf, which is
 * bit-identical for alpha == 1 (ATen copies) and may differ in the last bit otherwise.
 * Bookkeeping parity is therefore defined from the weights onward: every entry point
 * below also has a *_w form that takes the already-exponentiated weights.
 *
 * Every function cites the reference file:line it restates (paths under /root/reference).
 */
#ifndef RELA_ORACLE_H
#define RELA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- RNG ----------
 * libstdc++ (GCC 11) std::mt19937 + std::uniform_real_distribution<float>, the
 * generator behind rela/prioritized_replay.h:267,279,346.  Third-party algorithm
 * (not vendored in the reference): MT19937 of Matsumoto & Nishimura, seeded by
 * std::mersenne_twister_engine::seed(value) (x[i] = 1812433253*(x[i-1]^(x[i-1]>>30))+i),
 * and std::generate_canonical<float,24> = float(u32) / 2^32 clamped below 1.           */
typedef struct {
  uint32_t mt[624];
  int idx;
} oracle_mt19937;

void oracle_mt_seed(oracle_mt19937* g, uint32_t seed);
uint32_t oracle_mt_next(oracle_mt19937* g);
/* std::generate_canonical<float, 24>(mt19937) for one raw draw */
float oracle_canonical_from_u32(uint32_t u);
/* uniform_real_distribution<float>(a, b)(g) */
float oracle_uniform_float(oracle_mt19937* g, float a, float b);

/* ------------------------------------------------------- prioritized replay ----
 * rela/prioritized_replay.h:14-171 (ConcurrentQueue) and :173-348 (PrioritizedReplay).
 * Elements are opaque to the algorithm; the oracle stores one int64 tag per slot.   */
typedef struct oracle_replay oracle_replay;

oracle_replay* oracle_replay_new(int capacity, int seed, float alpha, float beta);
void oracle_replay_free(oracle_replay* r);

/* add(sample, priority) :186-200 -> blockAppend :43-78.  Returns 0, or -1 if the block
 * does not fit (the reference would block on cvSize_ :47).                           */
int oracle_replay_add(oracle_replay* r, int n, const int64_t* tags, const float* priority);
int oracle_replay_add_w(oracle_replay* r, int n, const int64_t* tags, const float* weights);

/* sample_ :258-328 with prefetch == 0.  ids = physical slots (sampledIds_), tags = the
 * sampled elements, is_w = importance weights (:320-322).  Returns 0; -2 if the
 * previous batch was not updated (:203-206); -3 if the scan ran off the end (:297-302). */
int oracle_replay_sample(oracle_replay* r, int batch, int32_t* ids, int64_t* tags, float* is_w);
/* the stratified targets of the last sample call (after the sum-0.2 clamp), for tests */
const float* oracle_replay_last_targets(const oracle_replay* r);
/* the raw (un-normalised) weights w_i picked by the last sample call (:289)          */
const float* oracle_replay_last_raw_w(const oracle_replay* r);

/* updatePriority :235-245 -> update :105-119 */
int oracle_replay_update(oracle_replay* r, int n, const float* priority);
int oracle_replay_update_w(oracle_replay* r, int n, const float* weights);

int oracle_replay_size(const oracle_replay* r);      /* safeSize_ :245-247 */
int oracle_replay_full_size(const oracle_replay* r); /* size_            */
int64_t oracle_replay_num_add(const oracle_replay* r);
int oracle_replay_head(const oracle_replay* r);
int oracle_replay_tail(const oracle_replay* r);
int oracle_replay_ring(const oracle_replay* r); /* int(1.25*capacity) :181 */
double oracle_replay_sum(const oracle_replay* r);
const float* oracle_replay_weights(const oracle_replay* r); /* ring-sized */
const uint8_t* oracle_replay_evicted(const oracle_replay* r);

/* bare ConcurrentQueue ops used by the reference's test_concurrent_queue.cc */
int oracle_cqueue_pop(oracle_replay* r, int n); /* blockPop :84-103 */

/* the sequential f64 prefix scan of :266-308 on an arbitrary float array: for each of
 * nt ascending targets, the first index k with acc > 0 && acc >= target; -1 if none.   */
void oracle_scan_search(const float* w, int n, const float* targets, int nt, int32_t* out_idx,
                        double* out_acc);

/* ------------------------------------------------------------- n-step buffer ---
 * rela/dqn_actor.h:58-106 (MultiStepTransitionBuffer::popTransition) on flat arrays:
 * rewards/terminals are [n+1][K] histories (row 0 oldest).                           */
void oracle_nstep_pop(int multi_step, int K, float gamma, const float* reward_hist,
                      const uint8_t* terminal_hist, float* out_reward, float* out_bootstrap,
                      uint8_t* out_terminal);

/* ----------------------------------------------------------------- DQN trunk ---
 * pyrela/net.py:8-55 (AtariFFNet) and pyrela/apex.py:30-78, fp32, naive loops.
 * Weight pointers follow the state_dict layout (N1 in SURVEY 8a).                  */
typedef struct {
  int num_action;
  const float *c1w, *c1b; /* (32,4,8,8),  (32) */
  const float *c2w, *c2b; /* (64,32,4,4), (64) */
  const float *c3w, *c3b; /* (64,64,3,3), (64) */
  const float *l1w, *l1b; /* (512,3136),  (512) */
  const float *vw, *vb;   /* (1,512),     (1)   */
  const float *aw, *ab;   /* (A,512),     (A)   */
} oracle_ffnet;

/* net.py:42-55: q[N,A] from s u8[N,4,84,84], legal f32[N,A] */
void oracle_ffnet_forward(const oracle_ffnet* net, int N, const uint8_t* s, const float* legal,
                          float* q);
/* apex.py:48-54: greedy action with the batch-global q.min() */
void oracle_greedy(int N, int A, const float* q, const float* legal, int64_t* action);
/* apex.py:30-45,68-78: |target - Q(s,a)| with double-DQN target */
void oracle_set_threads(int n);
void oracle_apex_priority(const oracle_ffnet* online, const oracle_ffnet* target, int N,
                          const uint8_t* s, const float* legal, const int64_t* action,
                          const float* reward, const float* bootstrap, const uint8_t* next_s,
                          const float* next_legal, float gamma_n, float* td_err_signed,
                          float* priority);

/* ------------------------------------------------------- R2D2 sequence buffer ---
 * rela/r2d2_actor.h:10-187 (R2D2TransitionBuffer) on scalar tags, and
 * R2D2Agent.aggregate_priority (pyrela/r2d2.py:103-120).                                  */
typedef struct oracle_r2d2buf oracle_r2d2buf;
oracle_r2d2buf* oracle_r2d2buf_new(int K, int multi_step, int seq_len, int burn_in);
void oracle_r2d2buf_free(oracle_r2d2buf* b);
/* push :29-87 for all K envs; returns canPop */
int oracle_r2d2buf_push(oracle_r2d2buf* b, const int64_t* tag, const int64_t* action, const float* reward,
                        const float* bootstrap, const uint8_t* terminal, const float* priority,
                        const float* hid_tag);
/* popTransition :93-170; outputs hold up to 2K sequences of T = burn+seq+n slots ([q][T]) and
 * seq_len priorities ([q][seq]); returns the number of sequences emitted                   */
int oracle_r2d2buf_pop(oracle_r2d2buf* b, float* out_len, float* out_h0, int* out_env, int64_t* tag,
                       int64_t* action, float* reward, uint8_t* terminal, float* bootstrap, float* prio);
void oracle_r2d2_aggregate(int nseq, int seq_len_const, int burn_in, float eta, const float* priority,
                           const float* seq_len, float* out);

#ifdef __cplusplus
}
#endif
#endif
