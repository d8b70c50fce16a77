/* rela_amd.h -- C ABI of the MI355X-native actor-learner hot path (librela_amd.so).
 *
 * Plain pointers, sizes and int status codes only; no torch / pybind types.  This is the
 * boundary the reference's own binding layer (rela/pybind.cc) would call into: every entry
 * point cites the reference interface it stands in for (paths under the reference repo).
 * The pybind module `rela` in rela_amd/pybind/ is a thin marshalling layer over this file
 * (see INTEGRATION.md for the binding a reference maintainer would add).
 *
 * Conventions
 *   - "dev" pointers are device (HBM) addresses on the GPU the object was created on;
 *     "host" pointers are ordinary process memory.
 *   - `stream` arguments are hipStream_t handles passed as void* (NULL = default stream).
 *     Work is stream-ordered; no entry point synchronises the device unless it says so.
 *   - Every function returns RELA_OK (0) or a negative RELA_E* code and never throws.
 *   - There is no CPU fallback: without a usable GPU the create calls return RELA_ENODEV.
 */
#ifndef RELA_AMD_H
#define RELA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RELA_OK 0
#define RELA_EINVAL (-1)     /* bad argument / shape                                          */
#define RELA_ENODEV (-2)     /* no HIP device / HIP runtime error (see rela_last_error)       */
#define RELA_ENOMEM (-3)     /* device allocation failed                                      */
#define RELA_ESTATE (-4)     /* protocol violation, e.g. sample() twice without update        */
#define RELA_ESCAN (-5)      /* a stratified target ran off the end of the ring               */
#define RELA_EWOULDBLOCK (-6) /* non-blocking add: ring is full                                */

const char* rela_last_error(void); /* thread-local message of the last failing call          */
int rela_abi_version(void);

/* A private non-blocking HIP stream for one host thread (each reference actor thread,
 * rela/context.h:39-46, gets its own so actor shards overlap on the GPU).                    */
int rela_stream_create(void** out, int device);
void rela_stream_destroy(void* stream, int device);
int rela_stream_synchronize(void* stream, int device);
/* work queued on `waiter` after this call starts only after everything queued on `signaler` so far */
int rela_stream_wait_stream(void* waiter, void* signaler, int device);
/* asynchronous host->device copy on `stream` (page-locked sources overlap with compute) */
int rela_memcpy_h2d_async(void* dst_dev, const void* src_host, int64_t bytes, void* stream, int device);

/* ===================================================================================
 * Prioritized replay  --  rela/prioritized_replay.h:173-348 (PrioritizedReplay<T>) over
 * :14-171 (ConcurrentQueue<T>), bound in rela/pybind.cc:37-59.
 *
 * Storage is device resident: one f32 weight ring of int(1.25*capacity) slots (:181),
 * an evicted-flag ring, the f64 running sum, and one row array per field of the
 * transition (structure of arrays, row r of field f at base_f + slot*row_bytes_f).
 * =================================================================================== */
typedef struct rela_replay rela_replay;

/* PrioritizedReplay(capacity, seed, alpha, beta, prefetch)  prioritized_replay.h:175-184.
 * `prefetch` is accepted for signature parity; sampling here is asynchronous device work,
 * so results always follow the prefetch == 0 ordering.                                    */
int rela_replay_create(rela_replay** out, int capacity, int seed, float alpha, float beta,
                       int prefetch, int device);
void rela_replay_destroy(rela_replay* r);

/* Declares the transition layout once, before the first add: nfields rows per slot, field
 * f being row_bytes[f] bytes (FFTransition: types.h:18-51; RNNTransition: types.h:53-73).  */
int rela_replay_set_schema(rela_replay* r, int nfields, const int64_t* row_bytes);

/* Sequence layout (RNNTransition, rela/types.h:53-73): like set_schema, but a field with
 * steps[f] = T > 1 is a sequence of T equal sub-rows and is gathered TIME-MAJOR on sample,
 * out[(t*batch + b)] = slot_b[t], i.e. RNNTransition::makeBatch's stack along dim 1
 * (rela/types.cc:140-182).  steps[f] = 1 keeps the plain [batch][row] layout.               */
int rela_replay_set_schema_seq(rela_replay* r, int nfields, const int64_t* row_bytes,
                               const int32_t* steps);

/* Frame-stack de-duplication.  The reference's transitions are VIEWS: obs of one transition and
 * next_obs of an earlier one share storage (rela/types.cc:48-67), and an Atari observation is a sliding stack of
 * four 84x84 planes of which one is new per step (atari/game_state.h:53-82).  With this schema the two stack
 * fields (field_a, field_b, each units_per_stack * unit_bytes long from the caller's point of view) are kept as
 * units_per_stack int32 references into ONE ring of units (unit = a whole stack, units_per_stack 1: 28,224 B per
 * env-step instead of 56,448; or unit = one plane, units_per_stack 4: 7,056 B); sample() rebuilds the full
 * stacks in its gather, so the learner sees exactly the batch it would get without de-duplication.  The ring
 * holds int(1.25 * capacity) + guard_units units: guard = the units a producer stores ahead of the transitions
 * that use them ((multi_step + 1 [+ 3 planes]) * rows).  Producers: reserve units (blocks like begin_add while
 * a live slot still refers to the unit that would be overwritten), write them, pass the references as the
 * rows of the two stack fields, and declare the smallest sequence number a block refers to before commit. */
int rela_replay_set_schema_dedup(rela_replay* r, int nfields, const int64_t* row_bytes, int field_a,
                                 int field_b, int64_t unit_bytes, int units_per_stack, int64_t guard_units);
/* first_seq: monotone sequence number of the first unit; first_index (may be NULL): its ring index, the
 * value a reference holds (index of unit q = (first_index + q) mod capacity, see rela_replay_dedup_info) */
int rela_replay_units_reserve(rela_replay* r, int count, int nonblocking, int64_t* first_seq, int32_t* first_index);
/* copies count units from src_dev (unit q at src_dev + q * src_pitch) into the reserved range */
int rela_replay_units_write(rela_replay* r, int64_t first_seq, int count, const void* src_dev, int64_t src_pitch,
                            void* stream);
int rela_replay_set_block_min_unit(rela_replay* r, int first_slot, int n, int64_t min_seq);
int rela_replay_dedup_info(const rela_replay* r, int* units_per_stack, int64_t* unit_bytes, int64_t* unit_capacity);

/* blockAppend in its three phases (prioritized_replay.h:43-78), for producers that assemble a
 * block piecewise (the R2D2 actor emits several sequences per pop):
 *   begin  :46-56  reserve n slots (blocks while the ring is full unless nonblocking)
 *   write  :61-66  copy rows into reserved slots [first_slot+offset, +count)
 *   commit :69-76  wait for in-order commit, then store pow(priority, alpha), add the FLOAT block
 *                  sum to sum_, and make the block visible to sample().
 * rela_replay_add is begin + write + commit.                                               */
int rela_replay_begin_add(rela_replay* r, int n, int nonblocking, int* first_slot);
int rela_replay_write_rows(rela_replay* r, int first_slot, int offset, int count,
                           const void* const* rows_dev, void* stream);
/* write, gathered: row i of the call comes from source row src_index_dev[f][i] of field f's base
 * array (row pitch = the field's row_bytes) and goes to reserved slot first_slot + dst_offset_dev[i].
 * One launch per field for any number of rows (the R2D2 actor emits all sequences of a pop so). */
int rela_replay_write_rows_gather(rela_replay* r, int first_slot, int count,
                                  const int32_t* dst_offset_dev, const void* const* bases_dev,
                                  const int32_t* const* src_index_dev, void* stream);
int rela_replay_commit_add(rela_replay* r, int first_slot, int n, const float* priority_dev,
                           void* stream);
/* The same commit for n = G*group_rows slots that stand for G consecutive reference blocks of
 * group_rows items (G actor threads batched into one launch): the float block sum and the
 * `sum_ +=` of :58-66,73 are taken per group, in order, exactly as G separate appends would.   */
int rela_replay_commit_add_grouped(rela_replay* r, int first_slot, int n, int group_rows,
                                   const float* priority_dev, void* stream);

/* Releases a reservation whose producer failed between begin and commit (no reference counterpart:
 * there an exception on an actor thread ends the process, rela/context.h:39-46).  The n slots are
 * committed in order with ZERO weight: safe_tail advances so that later blocks can commit, sum_ is
 * unchanged and a zero-weight slot can never be drawn (:286).                                  */
int rela_replay_abort_add(rela_replay* r, int first_slot, int n);

/* add(sample, priority)  prioritized_replay.h:186-200 -> blockAppend :43-78.
 * rows_dev[f] points at n consecutive rows of field f (device); priority_dev is f32[n]
 * (device).  Weights are pow(priority, alpha) (:188).  Blocks while the ring cannot take n
 * more slots (:47) unless `nonblocking`, in which case it returns RELA_EWOULDBLOCK.
 * `stream` is the producer's stream: the copy is ordered after work already queued there
 * and the producer may reuse its buffers for work queued on that stream afterwards.        */
int rela_replay_add(rela_replay* r, int n, const void* const* rows_dev, const float* priority_dev,
                    int nonblocking, void* stream);

/* sample(batchsize, device)  prioritized_replay.h:202-233 -> sample_ :258-328 + makeBatch
 * (types.cc:8-46).  Writes batch rows of every field to out_rows_dev[f] (device, batch rows
 * each) and the importance weights (:320-322) to out_weight_dev (f32[batch], device).
 * out_rows_dev may be NULL to skip the gather (benchmark.py:93-96 ignores the batch; the owner's half of the native
 * partition exchange below).  The slots such a gather-less sample evicts stay RESERVED until the update_priority that
 * ends the batch: inserts block instead of rewriting rows a remote reader may still be gathering.
 * `stream`: the consumer's stream; outputs are valid for work queued on it afterwards.
 * Exactly one batch may be outstanding (:203-206) -> RELA_ESTATE.                          */
int rela_replay_sample(rela_replay* r, int batch, void* const* out_rows_dev, float* out_weight_dev,
                       void* stream);

/* updatePriority(priority)  prioritized_replay.h:235-245 -> update :105-119.
 * priority is f32[n] for the outstanding batch; on_device selects host or device memory
 * (the reference takes a CPU tensor; a device pointer avoids the learner's per-step sync,
 * pyrela/apex.py:90).  `stream`: the stream that produced a device-side priority.          */
int rela_replay_update_priority(rela_replay* r, int n, const float* priority, int on_device,
                                void* stream);

/* Where the bulk of an insert runs (no reference counterpart: ConcurrentQueue::blockAppend copies its block with the
 * mutex released, rela/prioritized_replay.h:58-66, i.e. concurrently with a running sample_).  on = 0 (default): on the
 * replay's stream, in commit order with sample / update_priority -- right when one host thread drives actors and
 * learner and pipelines the sample path itself (bench.py).  on = 1: the row copies (56 KB per transition) and the
 * priority staging run on a second stream and only the weight / sum_ commit takes its turn -- right when an
 * independent sampler keeps the replay's stream busy (the threaded `rela` module, which turns it on).  Same results. */
int rela_replay_set_decoupled_insert(rela_replay* r, int on);
/* Learners that overlap the sample path with their own backward pass (rela_apex_learner_loss / _grad):
 * with on = 1, rela_replay_sample and a device-side rela_replay_update_priority no longer make the caller's
 * stream wait for the replay's stream; the caller inserts that wait itself with rela_replay_wait(r, stream)
 * before it reads the sampled rows (and keeps the priority buffer untouched until then).  Results do not
 * change: the replay's own work stays serialised on its stream in call order. */
int rela_replay_set_deferred_wait(rela_replay* r, int on);
int rela_replay_wait(rela_replay* r, void* stream);

/* Device pointers to what the last sample() left behind, for exchanges between replay partitions
 * (SURVEY 8e): the un-normalised weights w_i of the outstanding batch (:289) and the float sum
 * the stratified targets were drawn against (:261-262).  Valid until the next sample().       */
int rela_replay_last_sample_dev(rela_replay* r, const float** raw_w_dev, const float** sum_f_dev);
/* the size the last sample()'s IS weights used (:312,321: size_ re-read before the pop, reservations included) */
int rela_replay_last_sample_size(const rela_replay* r);

/* Teardown aid (no reference counterpart: there a producer parked on a full ring, :47, keeps its
 * Context from joining forever).  After shutdown every pending and future begin_add/add returns
 * RELA_EWOULDBLOCK immediately; sample/update keep working.                                  */
int rela_replay_shutdown(rela_replay* r);

/* capacity and ring size (int(1.25 * capacity), :181).  sample() evicts down to `capacity`, so a
 * blocking append of more than ring - capacity rows can never be satisfied: producers of large blocks
 * (the batched actor shards) insert in pieces of at most that many rows.                        */
int rela_replay_limits(const rela_replay* r, int* capacity, int* ring);
int rela_replay_size(const rela_replay* r);        /* size()    :245-247 */
int64_t rela_replay_num_add(const rela_replay* r); /* numAdd()  :251-253 */

/* Test / diagnostic taps (synchronise the replay's stream).  ids_host receives the
 * physical slots of the outstanding batch (sampledIds_), raw_w_host their un-normalised
 * weights, targets_host the stratified targets; any of them may be NULL.                    */
typedef struct {
  int32_t head, tail, size, safe_size, ring, n_sampled;
  int64_t num_add;
  double sum;       /* ConcurrentQueue::sum_ */
  int32_t dev_error; /* sticky device-side error flag (RELA_ESCAN ...) */
  int32_t pad;
} rela_replay_state;
int rela_replay_debug_state(rela_replay* r, rela_replay_state* out, int32_t* ids_host,
                            float* raw_w_host, float* targets_host);
int rela_replay_debug_weights(rela_replay* r, float* weights_host, uint8_t* evicted_host);
/* raw rows [slot, slot+count) of one field (physical slots, no wrap) -> host; synchronises */
int rela_replay_debug_read_rows(rela_replay* r, int field, int slot, int count, void* rows_host);

/* The scan primitive on its own: for nt targets (f64, host, ascending not required) over the
 * logical range [head, head+size) of a device weight ring, the first index whose
 * sequentially-rounded f64 prefix sum reaches the target (:266-308).  Synchronous; for
 * tests and for the scan roofline measurement.  out_* are host arrays of nt.               */
int rela_seqscan_search(const float* ring_dev, int64_t ring, int64_t head, int64_t size,
                        const double* targets_host, int nt, int64_t* out_index, double* out_acc,
                        float* out_w, double* out_total, void* stream);

/* Test tap: out[i] = pow(x[i], exponent) exactly as the replay evaluates torch::pow(tensor[n], exponent) of
 * rela/prioritized_replay.h:188,239,321 (SLEEF powf for the 32-wide vector part, double pow for the n % 32 tail) */
int rela_debug_pow(const float* x_dev, int n, float exponent, float* out_dev, void* stream);

/* Test hook of the scan index (csrc/seqsum.hip): 0 = normal; 1 = binade guesses perturbed, 2 = every guess
 * invalid, 3 = crossing records split one element late.  The guesses only decide how much work the exact
 * evaluation skips, so every result must be bit-identical in all modes (tests/test_replay_gpu.py).      */
int rela_seqscan_debug_perturb(int mode);

/* ===================================================================================
 * n-step return  --  MultiStepTransitionBuffer::popTransition, rela/dqn_actor.h:58-106.
 * reward_hist / terminal_hist are [multi_step+1][K] device arrays holding the deque of
 * dqn_actor.h:120-123 as a ring: step j (0 = oldest) lives in row (first_row + j) % (n+1).
 * =================================================================================== */
int rela_nstep_return(int multi_step, int K, float gamma, int first_row,
                      const float* reward_hist_dev, const uint8_t* terminal_hist_dev,
                      float* out_reward_dev, float* out_bootstrap_dev, uint8_t* out_terminal_dev,
                      void* stream);

/* ===================================================================================
 * Ape-X network and agent ops  --  pyrela/net.py:8-55 (AtariFFNet), pyrela/apex.py:30-78.
 * A net object owns one immutable, kernel-friendly copy of the parameters; ModelLocker
 * (rela/model_locker.h:11-65) swaps whole net objects.
 * =================================================================================== */
typedef struct rela_ffnet rela_ffnet;

typedef struct {
  const float *conv1_w, *conv1_b; /* net.0.weight (32,4,8,8),   net.0.bias (32)  */
  const float *conv2_w, *conv2_b; /* net.2.weight (64,32,4,4),  net.2.bias (64)  */
  const float *conv3_w, *conv3_b; /* net.4.weight (64,64,3,3),  net.4.bias (64)  */
  const float *fc_w, *fc_b;       /* linear.0.weight (512,3136), linear.0.bias   */
  const float *v_w, *v_b;         /* fc_v.weight (1,512),  fc_v.bias (1)         */
  const float *a_w, *a_b;         /* fc_a.weight (A,512),  fc_a.bias (A)         */
} rela_ffnet_params;

int rela_ffnet_create(rela_ffnet** out, int num_action, int device);
void rela_ffnet_destroy(rela_ffnet* net);
/* load_state_dict (model_locker.h:31): params are f32 in state_dict layout, host or device */
int rela_ffnet_load(rela_ffnet* net, const rela_ffnet_params* params, int params_on_device,
                    void* stream);
int rela_ffnet_num_action(const rela_ffnet* net);
/* number of completed rela_ffnet_load calls: (net, version) names one set of weights, which lets a
 * caller reuse a forward it already ran on the same input (rela_apex_actor_post_step does) */
uint64_t rela_ffnet_version(const rela_ffnet* net);
/* Arithmetic of conv2 / conv3 / fc in rela_ffnet_forward.  0 (default, the parity mode): v_mfma_f32_16x16x4_f32,
 * bit-for-bit an f32 fmaf chain.  1: split-bf16 on v_mfma_f32_16x16x32_bf16 -- every activation and weight is
 * carried as hi + lo bf16 (16 mantissa bits) and a product is a_lo*b_hi + a_hi*b_lo + a_hi*b_hi with f32
 * accumulation: ~2^-16 relative error per product instead of 2^-24, 5.3x less matrix-core time.  Stated
 * tolerance: Q-values within 1e-4 (abs + rel) of the f32 path and of the reference goldens
 * (tests/test_ffnet_gpu.py reports the greedy-action agreement).  conv1 and the heads are exact in both.
 * 2 ("f32x3"): f32 ACCURACY from the bf16 matrix cores -- conv2 / conv3 (batches of >= 512 rows) and fc (>= 4,096) take
 * every f32 operand as three exact bf16 parts (24 significand bits) and the six products with i + j <= 2, f32
 * accumulation, small terms in accumulators of their own (csrc/gemm_f32emu.h); against f64 its Q-values are as close
 * as mode 0's and as torch CPU f32's (tests/test_ffnet_gpu.py::test_ffnet_f32x3_is_f32_accurate); smaller batches run
 * mode 0's kernels, activations stay channel-last f32 between the layers. */
int rela_ffnet_set_precision(rela_ffnet* net, int mode);
int rela_ffnet_precision(const rela_ffnet* net);
/* Test tap: synchronises the device and returns the sticky give-up word of the pipelined conv1 -> conv2 kernel
 * (0 = no wave ever gave up waiting on a hand-off; anything else invalidates the forwards since the last read). */
/* Diagnostic build of the fused conv1 -> conv2 kernel: shader-clock stamps at its 12 phase boundaries for the first 8
 * frames of block 0 (waves 0 and 7) -> out_host [2][8][12] u64 (tools/conv12_phases.py). */
int rela_ffnet_debug_conv12_stamps(const rela_ffnet* net, int n, const uint8_t* s_dev, unsigned long long* out_host,
                                   void* stream);
/* Test hook: conv1 -> conv2 of the split-bf16 mode for n frames through the job form of the kernel.  a1_records
 * [n][400][hi 32 | lo 32] bf16 and a2_records [n][81][hi 64 | lo 64] bf16 are device buffers; scale_host / bias_host
 * (32 floats each, or NULL) receive conv1's packed per-channel scales and biases (conv1 runs on the int8 matrix cores,
 * csrc/ffnet.hip: conv12_i8). */
int rela_ffnet_debug_conv12_records(const rela_ffnet* net, int n, const uint8_t* s_dev, uint8_t* a1_records,
                                    uint8_t* a2_records, float* scale_host, float* bias_host, void* stream);
/* The same for conv3 of the split-bf16 mode (the first 8 groups of two frames of block 0; 5 points per group). */
int rela_ffnet_debug_conv3_stamps(const rela_ffnet* net, int n, const uint8_t* a2_records, unsigned long long* out_host,
                                  void* stream);
/* The same for fc_bf16s (positions 8..15 of block 0; 5 points per position). */
int rela_ffnet_debug_fc_stamps(const rela_ffnet* net, int n, const uint8_t* a3_records, unsigned long long* out_host,
                               void* stream);
/* bytes of scratch rela_ffnet_forward needs for a batch of n */
int64_t rela_ffnet_workspace_bytes(const rela_ffnet* net, int n);

/* AtariFFNet.forward  net.py:42-55: q[n,A] (device f32) from s u8[n,4,84,84], legal f32[n,A] */
int rela_ffnet_forward(const rela_ffnet* net, int n, const uint8_t* s_dev, const float* legal_dev,
                       float* q_dev, void* workspace_dev, int64_t workspace_bytes, void* stream);

/* ApexAgent.act  apex.py:57-65 on top of greedy_act :48-54: eps-greedy over q[n,A].
 * eps_dev f32[n]; rng_seed/rng_offset select the Philox stream for the random branch
 * (the reference uses the torch global generator: only the eps==0 branch is reproducible,
 * SURVEY H4).  action_dev is int64[n].
 * group_rows: greedy_act's q.min() (apex.py:51) spans one TorchScript call, i.e. the K rows of
 * one actor thread.  A launch that batches several actors passes K here and the minimum is
 * taken per group of K consecutive rows; 0 means the whole batch is one group.             */
int rela_apex_act_from_q(int n, int num_action, int group_rows, const float* q_dev,
                         const float* legal_dev, const float* eps_dev, uint64_t rng_seed,
                         uint64_t rng_offset, int64_t* action_dev, void* stream);

/* ApexAgent.td_err / compute_priority  apex.py:30-45,68-78 from the three Q tables:
 * q = online(s), q_next_online = online(s'), q_next_target = target(s').
 * td_err_dev (signed, may be NULL) and priority_dev (= |td_err|, may be NULL) are f32[n]. */
int rela_apex_td_from_q(int n, int num_action, int group_rows, const float* q_dev,
                        const float* q_next_online_dev,
                        const float* q_next_target_dev, const float* next_legal_dev,
                        const int64_t* action_dev, const float* reward_dev,
                        const float* bootstrap_dev, float gamma_n, float* td_err_dev,
                        float* priority_dev, void* stream);

/* ===================================================================================
 * R2D2 network  --  AtariLSTMNet, pyrela/net.py:58-163: the same conv trunk, then one LSTM layer
 * 3136 -> 512 (torch gate order i,f,g,o) and the dueling heads on the LSTM output.
 * =================================================================================== */
typedef struct rela_lstmnet rela_lstmnet;

typedef struct {
  const float *conv1_w, *conv1_b, *conv2_w, *conv2_b, *conv3_w, *conv3_b; /* net.{0,2,4}.*          */
  const float *w_ih, *w_hh;   /* lstm.weight_ih_l0 (2048,3136), lstm.weight_hh_l0 (2048,512)         */
  const float *b_ih, *b_hh;   /* lstm.bias_ih_l0 (2048), lstm.bias_hh_l0 (2048)                       */
  const float *v_w, *v_b;     /* fc_v.weight (1,512), fc_v.bias (1)                                   */
  const float *a_w, *a_b;     /* fc_a.weight (A,512), fc_a.bias (A)                                   */
} rela_lstmnet_params;

int rela_lstmnet_create(rela_lstmnet** out, int num_action, int device);
void rela_lstmnet_destroy(rela_lstmnet* net);
int rela_lstmnet_load(rela_lstmnet* net, const rela_lstmnet_params* params, int params_on_device,
                      void* stream);
/* 0 = exact f32 (default), 1 = split-bf16 MFMA conv trunk for batches of 128 rows and more (as
 * rela_ffnet_set_precision) and, from 1,024 rows up, the input side of the LSTM gate GEMM as well (h x W_hh, the
 * cell and the heads stay f32); h, c, Q within 4e-6 of mode 0.  Bumps the weight version. */
/* (mode 2, "f32x3": conv2 / conv3 of the trunk on the three-part bf16 kernels of rela_ffnet_set_precision's mode 2, from 512
 * rows; gate GEMM, cell and heads as in mode 0) */
int rela_lstmnet_set_precision(rela_lstmnet* net, int mode);
int rela_lstmnet_precision(const rela_lstmnet* net);
int rela_lstmnet_num_action(const rela_lstmnet* net);
uint64_t rela_lstmnet_version(const rela_lstmnet* net); /* as rela_ffnet_version */
int64_t rela_lstmnet_workspace_bytes(const rela_lstmnet* net, int n);

/* One time step for n rows: what AtariLSTMNet.act (net.py:110-124) and .forward with seq = 1
 * (:140-163) compute.  h_in/c_in/h_out/c_out are f32[n,512] (in and out may not alias);
 * q_dev (dueling Q, may be NULL) and adv_dev (raw fc_a output, the tensor `act` ranks, may be
 * NULL) are f32[n,A].                                                                        */
int rela_lstmnet_step(const rela_lstmnet* net, int n, const uint8_t* s_dev, const float* legal_dev,
                      const float* h_in, const float* c_in, float* h_out, float* c_out,
                      float* q_dev, float* adv_dev, void* workspace_dev, int64_t workspace_bytes,
                      void* stream);

/* ===================================================================================
 * Ape-X actor shard  --  DQNActor + MultiStepTransitionBuffer, rela/dqn_actor.h:15-211, as one
 * device-resident object for `rows` envs (rows = K for one reference actor thread, or T*K when
 * several threads are batched into one launch; group_rows = K keeps every batch-global reduction
 * of the reference at its original scope).  The observation history (multi_step+1 frame stacks
 * per env) lives in HBM: obs_t / obs_{t+n} are never re-uploaded for the priority pass and the
 * replay insert is a device-to-device row copy.  replay may be NULL (evaluation actor,
 * dqn_actor.h:141-147): then only act() is legal.
 * =================================================================================== */
typedef struct rela_apex_actor rela_apex_actor;

int rela_apex_actor_create(rela_apex_actor** out, int rows, int group_rows, int num_action,
                           int multi_step, float gamma, rela_replay* replay, uint64_t seed,
                           int device);
void rela_apex_actor_destroy(rela_apex_actor* a);

/* Device address ([rows][4][84][84] u8) the env layer may write the NEXT observation batch into
 * directly (then pass obs_host = NULL to act).                                              */
void* rela_apex_actor_obs_slot(rela_apex_actor* a);
/* Sliding frame stacks (GameState::computeFeature, atari/game_state.h:53-82: every step shifts the stack by one 84x84
 * plane and appends the new frame; the first frame of an episode is repeated four times).  The env layer uploads only
 * the NEWEST plane of every row -- 7,056 B instead of 28,224 B across PCIe -- into rela_apex_actor_plane_stage()
 * ([rows][84*84] u8 on the device, contiguous: one plain 1-D copy per actor thread), and rela_apex_actor_slide_stacks
 * writes the stacks of rela_apex_actor_obs_slot() on the device before act(): plane 3 = the new plane, plane k < 3 =
 * plane k + 1 of the previous observation slot, or the new plane again where restart_host[row] == 1 (u8[rows] on the
 * host: the row's episode just began); rows flagged 2 were uploaded whole into the slot and are left alone.
 * Stream-ordered after the uploads the caller made `stream` wait for.  The very first observation has no predecessor
 * and must be uploaded whole (RELA_ESTATE otherwise).  What it replaces: the host-side stacking the reference's env
 * does before VectorEnv::step stacks K observations (rela/env.h:63-82).                                          */
void* rela_apex_actor_plane_stage(rela_apex_actor* a);
int rela_apex_actor_slide_stacks(rela_apex_actor* a, const uint8_t* restart_host, void* stream);
/* Device addresses of the CURRENT obs["eps"] f32[rows] and obs["legal_move"] f32[rows][A]; act()
 * snapshots them into the history slot of the step, so a transition's obs side carries the values
 * of time t-n and its next_obs side those of time t (dqn_actor.h:84-90).                      */
float* rela_apex_actor_eps_dev(rela_apex_actor* a);
float* rela_apex_actor_legal_dev(rela_apex_actor* a);

/* DQNActor::act  dqn_actor.h:153-171 (TorchScript "act" = apex.py:57-65).
 * obs_host / eps_host / legal_host: host copies of obs["s"], obs["eps"], obs["legal_move"] for
 * this step, or NULL when the device copies are already current.  The chosen actions are left
 * in device memory (*action_dev_out, int64[rows], valid until the next act) and, if action_host
 * is not NULL, copied there and the stream is synchronised (the env layer needs them).        */
int rela_apex_actor_act(rela_apex_actor* a, const rela_ffnet* online, const uint8_t* obs_host,
                        const float* eps_host, const float* legal_host, int64_t* action_host,
                        const int64_t** action_dev_out, void* stream);

/* setRewardAndTerminal + postStep  dqn_actor.h:174-203: pushes (r, t) of the step just acted;
 * once multi_step+1 steps are buffered pops one n-step transition (:58-106), computes its TD
 * priority with the online/target nets (compute_priority, apex.py:68-78) and appends it to the
 * replay (:189).  *inserted (may be NULL) reports whether a block was appended.
 * With nonblocking != 0 a full ring drops the block (RELA_EWOULDBLOCK) instead of waiting.   */
int rela_apex_actor_post_step(rela_apex_actor* a, const float* reward, const uint8_t* terminal,
                              int on_device, const rela_ffnet* online, const rela_ffnet* target,
                              int nonblocking, int* inserted, void* stream);

/* Frame-stack de-duplication on the way in: the replay must have been given
 * rela_replay_set_schema_dedup with the same units_per_stack.  1: every stack enters the unit ring once (obs of
 * one transition and next_obs of another are the same stack, rela/types.cc:48-67) -- valid for any env.
 * 4: one NEW 84x84 plane per env-step; valid only for envs that stack frames as GameState::computeFeature does
 * (atari/game_state.h:53-82: slide by one plane per step, first frame of an episode repeated four times).
 * Call once, before the first act().                                                              */
int rela_apex_actor_set_dedup(rela_apex_actor* a, int units_per_stack);
int64_t rela_apex_actor_num_act(const rela_apex_actor* a); /* numAct()  dqn_actor.h:149-151 */
/* post_step evaluates online(obs) and online(next_obs) only if act() did not already evaluate that
 * observation with the same weights (n ticks ago and this tick; bit-identical, see post_step).
 * on = 1 (default): reuse both; 2: only the one of next_obs; 0: always recompute, i.e. the reference's
 * 4 forwards per step */
int rela_apex_actor_set_reuse(rela_apex_actor* a, int on);
/* diagnostic: device pointers of the last Q table of act() (it lives in the history slot that act() wrote: look the
 * pointer up after every act()) and of the last priorities */
const float* rela_apex_actor_last_q_dev(const rela_apex_actor* a);
const float* rela_apex_actor_last_priority_dev(const rela_apex_actor* a);

/* ===================================================================================
 * R2D2 actor shard  --  R2D2Actor + R2D2TransitionBuffer + MultiStepTransitionBuffer,
 * rela/r2d2_actor.h:10-353.  Per env: the n-step ring of the Ape-X shard, the recurrent state
 * with its history (historyHidden_ :345), and one window of burn_in + seq_len + multi_step slots
 * per field in HBM laid out as one replay row, so emitting a sequence is a row copy.  `eta` is
 * the max/mean mixing weight of aggregate_priority (pyrela/r2d2.py:103-120), a model constant in
 * the reference.  The replay must use the 10-field sequence schema
 *   s[T*28224] eps[T*4] legal_move[T*4A] a[T*8] reward[T*4] terminal[T] bootstrap[T*4]
 *   h0[2048] c0[2048] seq_len[4]          (RNNTransition, rela/types.h:53-73; T = window length)
 * =================================================================================== */
typedef struct rela_r2d2_actor rela_r2d2_actor;

int rela_r2d2_actor_create(rela_r2d2_actor** out, int rows, int group_rows, int num_action,
                           int multi_step, float gamma, int seq_len, int burn_in, double eta,
                           rela_replay* replay, uint64_t seed, int device);
void rela_r2d2_actor_destroy(rela_r2d2_actor* a);
void* rela_r2d2_actor_obs_slot(rela_r2d2_actor* a);
void* rela_r2d2_actor_plane_stage(rela_r2d2_actor* a);                                          /* as rela_apex_actor_plane_stage */
int rela_r2d2_actor_slide_stacks(rela_r2d2_actor* a, const uint8_t* restart_host, void* stream); /* as rela_apex_actor_slide_stacks */
/* R2D2Actor::act  r2d2_actor.h:221-249; arguments as rela_apex_actor_act */
int rela_r2d2_actor_act(rela_r2d2_actor* a, const rela_lstmnet* online, const uint8_t* obs_host,
                        const float* eps_host, const float* legal_host, int64_t* action_host,
                        const int64_t** action_dev_out, void* stream);
/* setRewardAndTerminal + postStep  r2d2_actor.h:252-302.  reward/terminal are HOST arrays (the
 * window bookkeeping branches on the terminal flags).  *n_sequences (may be NULL) = sequences
 * appended to the replay by this call.                                                      */
int rela_r2d2_actor_post_step(rela_r2d2_actor* a, const float* reward_host,
                              const uint8_t* terminal_host, const rela_lstmnet* online,
                              const rela_lstmnet* target, int nonblocking, int* n_sequences,
                              void* stream);
int64_t rela_r2d2_actor_num_act(const rela_r2d2_actor* a);
/* as rela_apex_actor_set_reuse: online_net.act(next_obs, next_hid) of compute_priority (r2d2.py:91) is the step
 * act() ran on this tick, and online_net(obs, hid) (:89) the step act() ran n ticks ago on the same frames, recurrent
 * state (historyHidden_.front()) and legal mask: with unchanged weights both are reused bit-identically.
 * on = 1 (default): both; 2: only the one of next_obs; 0: recompute */
int rela_r2d2_actor_set_reuse(rela_r2d2_actor* a, int on);
/* diagnostics: current recurrent state (which = 0: h, 1: c) f32[rows,512]; last step priorities */
const float* rela_r2d2_actor_hidden_dev(const rela_r2d2_actor* a, int which);
const float* rela_r2d2_actor_last_priority_dev(const rela_r2d2_actor* a);

/* ===================================================================================
 * Ape-X learner step  --  pyrela/main.py:206-251 with ApexAgent.loss (pyrela/apex.py:80-91):
 * td_err -> smooth_l1 * IS weight -> mean -> backward -> clip_grad_norm_ -> optimiser.
 * Replaces PyTorch autograd for AtariFFNet (net.py:8-55).  Parameters, gradients and optimiser
 * state live in one flat device buffer in rela_ffnet_params order (each tensor in state_dict
 * layout, segments padded to 4 floats).
 * =================================================================================== */
typedef struct rela_apex_learner rela_apex_learner;

/* optimizer: 0 = torch.optim.RMSprop(lr, eps) (main.py:120, alpha 0.99, no momentum),
 *            1 = torch.optim.Adam(lr, eps) (betas 0.9/0.999).  grad_clip = max_norm of
 * clip_grad_norm_ (main.py:233).  gamma ** multi_step scales the bootstrap (apex.py:44).      */
int rela_apex_learner_create(rela_apex_learner** out, int num_action, int max_batch, int multi_step,
                             float gamma, int optimizer, float lr, float eps, float grad_clip,
                             int device);
void rela_apex_learner_destroy(rela_apex_learner* l);
/* load_state_dict for online_net and target_net (target == NULL: copy of online); resets the
 * optimiser state                                                                            */
int rela_apex_learner_load(rela_apex_learner* l, const rela_ffnet_params* online,
                           const rela_ffnet_params* target, int params_on_device, void* stream);
/* ApexAgent.sync_target_with_online  apex.py:26-27 */
int rela_apex_learner_sync_target(rela_apex_learner* l, void* stream);
/* 0 = exact f32 (default).  1 = the two gradient-free forwards of td_err (online and target net on next_obs,
 * apex.py:38-42) on the split-bf16 MFMA trunk (rela_ffnet_set_precision), conv1's weight gradient and the conv2 /
 * conv3 data gradients on bf16 MFMA (hi + lo operands, f32 accumulation).  The online(obs) pass, whose activations
 * and ReLU masks the backward pass reads, always runs in f32: priorities and loss within 5e-6, every gradient within
 * 1e-4 of its largest entry of mode 0.  2 = f32x3 (rela_ffnet_set_precision): conv2 / conv3 of all three forwards (from
 * 512 rows) and the weight-gradient / fc / head GEMMs with three-part bf16 operands (csrc/gemm_bf16x3.h, PARTS = 3) --
 * f32 accuracy, mode 0's tolerances (tests/test_learner_gpu.py); conv1, the conv data gradients, loss, clip and
 * optimiser exactly as in mode 0. */
int rela_apex_learner_set_precision(rela_apex_learner* l, int mode);
/* loss + backward on one sampled batch.  rows_dev: the ten FFTransition fields in the order
 * rela_replay_sample fills them; weight_dev f32[batch] = the IS weights.  Leaves the gradient of
 * mean(smooth_l1(td_err) * weight) in the flat gradient buffer, |td_err| in priority_dev
 * (apex.py:88-90, feed it to rela_replay_update_priority) and the loss in loss_dev (may be NULL). */
int rela_apex_learner_backward(rela_apex_learner* l, int batch, const void* const* rows_dev,
                               const float* weight_dev, float* priority_dev, float* loss_dev,
                               void* stream);
/* The same step in two calls: _loss runs the three forwards of td_err, the priorities and the loss (arguments as
 * _backward); _grad the backward pass of that batch.  The backward pass does not touch the replay, so between the
 * two a caller may feed priority_dev to rela_replay_update_priority and sample the NEXT batch (into other buffers:
 * the `s` rows of this batch are read until _grad's work is done) -- the sample path then runs next to the
 * gradient kernels instead of after them.  Same results as _backward, bit for bit. */
int rela_apex_learner_loss(rela_apex_learner* l, int batch, const void* const* rows_dev,
                           const float* weight_dev, float* priority_dev, float* loss_dev, void* stream);
int rela_apex_learner_grad(rela_apex_learner* l, void* stream);
/* clip_grad_norm_ + optimiser step on the flat buffers, then re-packs the kernel-layout weights.
 * Data-parallel learners all-reduce the gradient buffer between backward and apply.          */
int rela_apex_learner_apply(rela_apex_learner* l, void* stream);
/* device views: parameters (online / target) and gradients as state_dict-layout tensors, e.g.
 * to publish new weights to the actors' nets with rela_ffnet_load(net, &p, 1, stream)        */
int rela_apex_learner_params(rela_apex_learner* l, rela_ffnet_params* online_out,
                             rela_ffnet_params* target_out);
int rela_apex_learner_grads(rela_apex_learner* l, rela_ffnet_params* grads_out);
int rela_apex_learner_flat(rela_apex_learner* l, float** params_dev, float** grads_dev, int64_t* count);
/* f32[2] on the device: total gradient norm before clipping, clip coefficient of the last apply */
const float* rela_apex_learner_stats_dev(const rela_apex_learner* l);
/* Debug / tests: the activations of online(obs) the last rela_apex_learner_loss left for the backward pass, f32,
 * channel-last: a1 [B][400][32], a2 [B][81][64], a3 [B][49][64], h [B][512] (their > 0 pattern is the ReLU mask the
 * gradients were computed with).  Valid until the next rela_apex_learner_loss.                  */
int rela_apex_learner_debug_activations(rela_apex_learner* l, float** a1, float** a2, float** a3, float** h, int* batch);

/* ===================================================================================
 * R2D2 learner step  --  pyrela/main.py:206-251 with R2D2Agent.loss (pyrela/r2d2.py:189-206):
 * td_err (:122-187: burn-in unroll without gradient, state zeroed where terminal[burn_in-1], training
 * unroll of online and target AtariLSTMNet (pyrela/net.py:127-163), per-step target with Q_target[t+n],
 * pad mask) -> smooth_l1 summed over the sequence * IS weight -> mean -> backward through the heads,
 * the LSTM (BPTT over seq_len + multi_step steps) and the conv trunk -> clip_grad_norm_ -> Adam
 * (main.py:124-126).  Replaces PyTorch autograd.  Flat buffers in rela_lstmnet_params order.
 * =================================================================================== */
typedef struct rela_r2d2_learner rela_r2d2_learner;

/* optimizer: 0 = RMSprop, 1 = Adam (the reference's choice for R2D2, main.py:124); eta = mixing weight of
 * aggregate_priority (r2d2.py:103-120); max_batch <= 1024 sequences of burn_in + seq_len + multi_step steps */
int rela_r2d2_learner_create(rela_r2d2_learner** out, int num_action, int max_batch, int multi_step,
                             float gamma, int seq_len, int burn_in, double eta, int optimizer, float lr,
                             float eps, float grad_clip, int device);
void rela_r2d2_learner_destroy(rela_r2d2_learner* l);
int rela_r2d2_learner_load(rela_r2d2_learner* l, const rela_lstmnet_params* online,
                           const rela_lstmnet_params* target, int params_on_device, void* stream);
int rela_r2d2_learner_sync_target(rela_r2d2_learner* l, void* stream); /* r2d2.py:54-55 */
/* loss + backward on one sampled batch.  rows_dev: the ten RNNTransition fields, time-major, in the order of
 * the sequence schema (s, eps, legal_move, a, reward, terminal, bootstrap, h0, c0, seq_len) as
 * rela_replay_sample fills them; weight_dev f32[batch].  Leaves the gradient of mean(loss * weight) in the flat
 * gradient buffer, the aggregated priority (r2d2.py:205) in priority_dev f32[batch], mean(loss * weight) in
 * loss_dev (may be NULL) and the per-sequence Huber sums (r2d2.py:203) in loss_seq_dev (may be NULL).       */
int rela_r2d2_learner_backward(rela_r2d2_learner* l, int batch, const void* const* rows_dev,
                               const float* weight_dev, float* priority_dev, float* loss_dev,
                               float* loss_seq_dev, void* stream);
/* The same step in two calls, as rela_apex_learner_loss / rela_apex_learner_grad: priority_dev is final after
 * _loss, so rela_replay_update_priority and the next rela_replay_sample (into other buffers) may be queued before
 * _grad; same results as _backward, bit for bit. */
int rela_r2d2_learner_loss(rela_r2d2_learner* l, int batch, const void* const* rows_dev,
                           const float* weight_dev, float* priority_dev, float* loss_dev,
                           float* loss_seq_dev, void* stream);
int rela_r2d2_learner_grad(rela_r2d2_learner* l, void* stream);
int rela_r2d2_learner_apply(rela_r2d2_learner* l, void* stream); /* clip + optimiser + repack */
int rela_r2d2_learner_params(rela_r2d2_learner* l, rela_lstmnet_params* online_out,
                             rela_lstmnet_params* target_out);
int rela_r2d2_learner_grads(rela_r2d2_learner* l, rela_lstmnet_params* grads_out);
int rela_r2d2_learner_flat(rela_r2d2_learner* l, float** params_dev, float** grads_dev, int64_t* count);
const float* rela_r2d2_learner_stats_dev(const rela_r2d2_learner* l); /* grad norm, clip coefficient */
/* The T recurrent steps run as ONE persistent launch per pass with a bounded grid barrier between steps (no reference
 * counterpart: autograd launches per step).  Synchronises `stream` and returns RELA_ESTATE if a barrier of any call
 * since the last check gave up (never observed; the results of that call are then invalid). */
int rela_r2d2_learner_check(rela_r2d2_learner* l, void* stream);
/* 1: the TARGET net's conv trunk (no gradient) runs on split-bf16 MFMA (rela_lstmnet_set_precision) and so do the GEMMs
 * of the LSTM's input side (gate GEMM of both nets, its data and weight gradients, dW_hh), the conv weight gradients
 * and the conv data gradients (hi + lo bf16 operands, f32 accumulation); the online net's conv trunk, whose
 * activations and ReLU masks the backward kernels read, stays f32.  Loss, priorities within 2e-5, gradients within
 * 2e-4 of their largest entry of mode 0.  0 (default): everything f32.  2 ("f32x3"): conv2 / conv3 of BOTH nets' trunk
 * forwards on the three-part bf16 kernels (f32 accuracy, mode 0's tolerances); everything else as in mode 0. */
int rela_r2d2_learner_set_precision(rela_r2d2_learner* l, int mode);

/* ===================================================================================
 * Live per-kernel timing (HIP events on the launch stream) for bench.py's roofline line.
 * No reference counterpart: the reference times sections with torch.cuda.synchronize()
 * (pyrela/common_utils/stopwatch.py:17-54).
 * =================================================================================== */
int rela_prof_enable(int on);
/* time only the named kernels (comma-separated; NULL or "" = all): keeps the events' own cost out
 * of a timed region that only needs the dominant kernel                                        */
int rela_prof_set_filter(const char* names);
/* synchronises the device; writes {"kernel":{"count":n,"total_ms":t},...} and clears */
int rela_prof_summary_json(char* out, int64_t cap);
/* Launch census for the parity tests: counts launches by the kernel that REALLY ran (the timing labels above are
 * per layer and shared by the f32 and the split-bf16 kernels), so a test of a fast mode can assert that the fast
 * kernels were launched and not silently replaced by the f32 ones.  rela_prof_counts_json writes
 * {"kernel": launches, ...} since the enable / the last call and clears.                         */
int rela_prof_count_enable(int on);
/* Compute units left OUT of the grids of the persistent forward kernels (one block per CU: conv1 -> conv2, conv3).  Such a
 * grid has no slack on 256 CUs: a CU held by another stream's small kernel (the replay's sample chain) when the launch
 * starts delays one block -- and the launch -- by that kernel's duration.  0 (the library default) = the whole chip, right
 * when ONE host thread pipelines actors, learner and replay (bench.py: 2.96 M env-steps/s against 2.89 M with 8); the
 * `rela` module sets 8, right next to an independent sampler thread (threaded benchmark, sliding env, sampler on: 1.96 M
 * -> 2.26 M env-steps/s).  RELA_CU_RESERVE in the environment wins over both.  The reference has no counterpart.        */
int rela_runtime_set_cu_reserve(int cus);
int rela_prof_counts_json(char* out, int64_t cap);

/* ===================================================================================
 * Native partition exchange over HIP IPC (the reference has no counterpart: its one replay lives in host
 * RAM, rela/prioritized_replay.h:156,339, and batches cross to the learner's GPU inside makeBatch, types.cc:34-43).
 *
 * With one replay partition per actor GPU and the learner in another process, the learner maps the partition -- its
 * field arrays, the ids / raw weights of its last sample and its device state -- through IPC handles and gathers the
 * B / G sampled rows ITSELF: a kernel on the learner's GPU reads the owner's HBM directly (peer reads over xGMI
 * between GPUs), writing straight into the learner's batch tensors.  No packing on the owner, no collective, no
 * staging copy.  Protocol per learner step (the ordering signals -- a few bytes -- stay with the caller, e.g.
 * torch.distributed; rela_amd/parallel.py):
 *   owner:    rela_replay_sample(part, B / G, NULL, weight_scratch, stream); synchronize; signal the learner
 *   learner:  rela_replay_remote_gather(remote, B / G, rows, raw_w, sum_f, stream); ... loss ...; synchronize; signal
 *   owner:    rela_replay_update_priority(part, ...)
 * The owner must not sample / update between its signal and the learner's.  Its actors MAY keep inserting: the library
 * holds every slot the sample evicted (a draw can land in the very range its own sample pops, and on a full ring blocked
 * producers would otherwise rewrite exactly those slots) until the owner's update_priority, i.e. until after the
 * learner's read; inserts into other slots never alias a sampled row (sampled ids are live or held).
 * =================================================================================== */
#define RELA_IPC_MAX_FIELDS 16
typedef struct rela_replay_ipc_desc { /* plain bytes: send it to the importing process by any means */
  int32_t abi, nfields, ring, device, max_batch, pad;
  int64_t row_bytes[RELA_IPC_MAX_FIELDS];
  int32_t steps[RELA_IPC_MAX_FIELDS];
  unsigned char field_handle[RELA_IPC_MAX_FIELDS][64]; /* hipIpcMemHandle_t of every field array */
  unsigned char ids_handle[64], raw_w_handle[64], state_handle[64];
} rela_replay_ipc_desc;
typedef struct rela_replay_remote rela_replay_remote;
int rela_replay_export_ipc(rela_replay* r, rela_replay_ipc_desc* out);
int rela_replay_import_ipc(rela_replay_remote** out, const rela_replay_ipc_desc* desc, int device);
void rela_replay_remote_close(rela_replay_remote* rr);
/* rows of the partition's LAST sample into out_rows_dev[f] (on this process's device, NULL = skip the field; sequence
 * fields come out time-major as from rela_replay_sample), its un-normalised weights into raw_w_out[batch] and the float
 * sum they were drawn against into sum_f_out[1] (either may be NULL).  The output tensors hold out_batch rows per (time)
 * step and this partition fills rows [out_offset, out_offset + batch): its share of a batch drawn over several
 * partitions (out_batch = 0: the outputs hold exactly this batch). */
int rela_replay_remote_gather(rela_replay_remote* rr, int batch, void* const* out_rows_dev, float* raw_w_out,
                              float* sum_f_out, int out_batch, int out_offset, void* stream);
/* Partitions of ANY size (r5).  One hipIpcMemHandle_t per field stops working at ~24 GB per field on this platform (the
 * import of a 37 GB frame-stack field does not return; rela_replay_export_ipc refuses such a partition).  A partition
 * created in CHUNKED mode allocates every field array as chunks of physical HBM of at most `bytes` behind one
 * contiguous virtual range (HIP virtual memory management; rows stay base + slot * row_bytes, every kernel of the library
 * is unchanged), and each chunk travels to the importing process as a POSIX file descriptor (a dmabuf):
 *   owner:    rela_replay_set_chunk_bytes(part, 8 << 30) BEFORE rela_replay_set_schema*   (0 = off: one hipMalloc per field;
 *             rela_runtime_set_replay_chunk_bytes sets the default of partitions created later, and RELA_REPLAY_CHUNK_GB
 *             in the environment wins over it)
 *             rela_replay_export_chunks(part, &desc, fds, RELA_IPC_MAX_FDS): desc.nfds descriptors, in field order;
 *             send desc as bytes and the descriptors with SCM_RIGHTS over a Unix socket (rela_amd/parallel.py:
 *             _FdServer), then close() them
 *   learner:  rela_replay_import_chunks(&remote, &desc, fds, desc.nfds, device); close() the descriptors;
 *             rela_replay_remote_gather / rela_replay_remote_close exactly as above
 * The small fixed arrays (ids / weights / state of the last sample) still travel as IPC handles inside desc.ipc, and so do
 * the fields of a partition that is NOT in chunked mode (field_chunks[f] = 0): rela_replay_export_chunks serves every
 * partition; rela_replay_export_ipc refuses a chunked one.  A chunked partition has no field left on an IPC handle on
 * purpose: after a 37 GB field had been mapped from its chunks, hipIpcOpenMemHandle of a 4 GB field of the same partition did
 * not return (profiles/r05_vmm_mixed_import_hang.log).  De-duplicated partitions (rela_replay_set_schema_dedup) are exported
 * with their unit ring; rela_replay_remote_gather rebuilds the frame stacks from it as rela_replay_sample does.
 * Measured on one MI355X, two processes: tests/test_ipc_gpu.py (partitions of 20 MB and 15 MB fields in 2 MB and 6 MB
 * chunks; a 2^20-row partition whose frame-stack field is 37 GB in five 8 GB chunks), profiles/r05_vmm_probe.jsonl.                                              */
#define RELA_IPC_MAX_FDS 128
typedef struct rela_replay_chunk_desc {
  rela_replay_ipc_desc ipc; /* ipc.field_handle[f] is unused where field_chunks[f] > 0 */
  int32_t abi, nfds;        /* abi = 2 */
  int32_t field_chunks[RELA_IPC_MAX_FIELDS]; /* 0: the field is ipc.field_handle[f]; n: the next n descriptors */
  int64_t chunk_bytes[RELA_IPC_MAX_FIELDS];  /* size of every chunk of the field but the last */
  int64_t mapped_bytes[RELA_IPC_MAX_FIELDS]; /* the field's virtual range (>= ring * row_bytes, whole pages) */
  /* de-duplicated partition (rela_replay_set_schema_dedup; dd_ups = 0: none): fields dd_field[0..1] hold int32 references
   * into the unit ring, which travels after the fields' descriptors (units_chunks > 0) or as units_handle */
  int32_t dd_ups, dd_field[2], units_chunks;
  int64_t dd_unit_bytes, dd_cap, units_chunk_bytes, units_mapped_bytes;
  unsigned char units_handle[64];
} rela_replay_chunk_desc;
int rela_replay_set_chunk_bytes(rela_replay* r, int64_t bytes);
int rela_runtime_set_replay_chunk_bytes(int64_t bytes);
int rela_replay_export_chunks(rela_replay* r, rela_replay_chunk_desc* out, int* fds_out, int max_fds);
int rela_replay_import_chunks(rela_replay_remote** out, const rela_replay_chunk_desc* desc, const int* fds, int nfds,
                              int device);
/* the same mapping for a device buffer this library allocated (e.g. rela_apex_learner_flat's parameter buffer): the
 * weight publish across processes -- actors load their nets straight from the learner's mapped buffer */
int rela_ipc_export_buffer(const void* dev_ptr, unsigned char handle_out[64]);
int rela_ipc_import_buffer(const unsigned char handle[64], void** dev_ptr_out, int device);
int rela_ipc_close_buffer(void* dev_ptr, int device);

/* ===================================================================================
 * Gradient all-reduce over IPC-mapped buffers (r5; SURVEY 8e.  The reference has no counterpart: one learner,
 * pyrela/main.py:206-251).  For the replicated layout -- one learner replica per GPU, one process per GPU -- the flat f32
 * gradient bucket is summed over the ranks without a collective library: every rank maps every other rank's bucket;
 * rank r sums slice r of all buckets IN RANK ORDER, ((g0 + g1) + g2) + ..., with peer reads, then copies every reduced slice
 * out of its owner's memory.  All ranks end with bit-identical sums, equal to a host-side f32 sum in rank order.
 *   every rank:  rela_ipc_allreduce_create(&ar, rank, world, bucket, count, device, device_flags, &desc)
 *                exchange the descriptors (plain bytes, any means), in rank order:  rela_ipc_allreduce_connect(ar, descs)
 *   every step:  rela_ipc_allreduce_run(ar, stream)    -- collective: every rank calls it once per step
 *                bucket := sum over ranks, in place, ordered on `stream` like a kernel
 * `bucket` must be device memory this library allocated (rela_apex_learner_flat's / rela_r2d2_learner_flat's gradients, or
 * rela_ipc_alloc_buffer).  device_flags = 1: the phases are ordered by monotonic step counters in a shared page that the
 * streams' command processors write and wait for (hipStreamWriteValue32 / hipStreamWaitValue32) -- no kernel spins, the
 * host never waits; connect() tests these operations and ALL ranks fall back to mode 0 if any rank's fail.  Mode 0: a
 * stream synchronisation and a host barrier (same page) per phase.  rela_ipc_allreduce_mode: what connect() settled on.
 * Same host only.  csrc/ipc_allreduce.hip states the hazards.  Validated with two and three processes on ONE GPU
 * (tests/test_ipc_allreduce_gpu.py); RCCL stays bench.py's default (`--allreduce ipc` selects this). */
#define RELA_IPC_ALLREDUCE_MAX_RANKS 8
typedef struct rela_ipc_allreduce rela_ipc_allreduce;
typedef struct rela_ipc_allreduce_desc { /* plain bytes */
  int32_t abi, rank, world, device, device_flags, pad;
  int64_t count, bucket_offset; /* the bucket starts bucket_offset bytes into the allocation bucket_handle names */
  unsigned char bucket_handle[64], red_handle[64];
  char shm_name[64]; /* rank 0's descriptor: the shared-memory page of the counters and the host barrier */
} rela_ipc_allreduce_desc;
int rela_ipc_allreduce_create(rela_ipc_allreduce** out, int rank, int world, float* bucket_dev, int64_t count, int device,
                              int device_flags, rela_ipc_allreduce_desc* desc_out);
int rela_ipc_allreduce_connect(rela_ipc_allreduce* ar, const rela_ipc_allreduce_desc* descs /* [world], rank order */);
int rela_ipc_allreduce_run(rela_ipc_allreduce* ar, void* stream);
int rela_ipc_allreduce_mode(const rela_ipc_allreduce* ar); /* 1 = stream value operations, 0 = host synchronisation */
void rela_ipc_allreduce_destroy(rela_ipc_allreduce* ar);
/* A page of 32-bit words shared by the processes of one host and by their GPUs' command processors (r5; csrc/ipc_page.hip):
 * the control plane of the native partition exchange.  POSIX shared memory, registered with HIP in every process, so a
 * word can be written by a stream (after everything queued before it), waited for by a stream (>=: the command
 * processor waits, no CU spins, the host goes on), read by a kernel (rela_ipc_page_dev_ptr) and read / written / waited
 * for by host threads.  Step counters only grow, so a wait binds to a value: nothing to re-arm, nothing to acknowledge.
 * rela_amd/parallel.py keeps, per partition g: sampled[g] (the owner's stream, after its gather-less sample), size[g]
 * (the partition's item count of that sample), consumed[g] (the learner's stream, after its gather and its priorities) --
 * the per-step gather of importance weights and scatter of priorities through torch.distributed are gone.       */
#define RELA_IPC_PAGE_BYTES 4096
typedef struct rela_ipc_page rela_ipc_page;
int rela_ipc_page_create(rela_ipc_page** out, char name_out[64], int device); /* zeroed; send the name to the others */
int rela_ipc_page_open(rela_ipc_page** out, const char* name, int device);
int rela_ipc_page_unlink(rela_ipc_page* p); /* the creator, once everyone has opened it */
void rela_ipc_page_close(rela_ipc_page* p);
void* rela_ipc_page_host_ptr(rela_ipc_page* p);
void* rela_ipc_page_dev_ptr(rela_ipc_page* p);
int rela_ipc_page_write32(rela_ipc_page* p, int word, uint32_t value, void* stream); /* stream operation */
int rela_ipc_page_wait32(rela_ipc_page* p, int word, uint32_t value, void* stream);  /* stream operation: word >= value */
int rela_ipc_page_host_store32(rela_ipc_page* p, int word, uint32_t value);
int rela_ipc_page_host_load32(rela_ipc_page* p, int word, uint32_t* value_out);
int rela_ipc_page_host_wait32(rela_ipc_page* p, int word, uint32_t value, double timeout_s); /* RELA_EWOULDBLOCK on timeout */
int rela_ipc_page_selftest(rela_ipc_page* p, int word, uint32_t value, int* ok_out); /* do the stream operations work here? */
/* exportable device memory for a caller that has none of the library's own (a Python learner's gradient bucket; tests):
 * a plain allocation of its own, which rela_ipc_export_buffer / rela_ipc_allreduce_create can name */
int rela_ipc_alloc_buffer(void** dev_ptr_out, int64_t bytes, int device);
int rela_ipc_free_buffer(void* dev_ptr, int device);

#ifdef __cplusplus
}
#endif
#endif /* RELA_AMD_H */
