/* oracle/replay_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Single-threaded restatement of rela/prioritized_replay.h: ConcurrentQueue (:14-171)
 * and PrioritizedReplay (:173-348, prefetch == 0).  The multi-producer protocol of
 * blockAppend (reserve under the mutex, fill unlocked, commit in slot order) collapses
 * to its sequential meaning here; what matters for parity is the arithmetic:
 *   - the per-block weight sum is accumulated in FLOAT and then added to the DOUBLE
 *     running sum_ (:58-66,73);
 *   - blockPop subtracts evicted weights into a double diff, one by one (:85-95);
 *   - update adds (float new - float old) into a double diff (:106-118);
 *   - sample_ narrows sum_ to float (:30-36,261-262), draws stratified float targets,
 *     and walks the ring from head_ with a double accumulator (:266-308).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

struct oracle_replay {
  /* PrioritizedReplay members :331-346 */
  float alpha, beta;
  int capacity_; /* logical capacity */
  int64_t num_add;
  oracle_mt19937 rng;
  int n_sampled; /* sampledIds_.size() */
  int32_t* sampled_ids;
  /* ConcurrentQueue members :156-170 */
  int ring; /* ConcurrentQueue::capacity == int(1.25*capacity) */
  int head, tail, size, safe_tail, safe_size;
  double sum;
  uint8_t* evicted;
  int64_t* elements;
  float* weights;
  /* test taps */
  float* last_targets;
  float* last_raw_w;
  int tap_cap;
};

oracle_replay* oracle_replay_new(int capacity, int seed, float alpha, float beta) {
  oracle_replay* r = (oracle_replay*)calloc(1, sizeof(*r));
  r->alpha = alpha;
  r->beta = beta;
  r->capacity_ = capacity;
  r->ring = (int)(1.25 * capacity); /* :181 */
  oracle_mt_seed(&r->rng, (uint32_t)seed); /* :183 */
  r->evicted = (uint8_t*)calloc((size_t)r->ring, 1);
  r->elements = (int64_t*)calloc((size_t)r->ring, sizeof(int64_t));
  r->weights = (float*)calloc((size_t)r->ring, sizeof(float));
  return r;
}

void oracle_replay_free(oracle_replay* r) {
  if (!r) return;
  free(r->sampled_ids);
  free(r->evicted);
  free(r->elements);
  free(r->weights);
  free(r->last_targets);
  free(r->last_raw_w);
  free(r);
}

/* blockAppend :43-78 */
int oracle_replay_add_w(oracle_replay* r, int n, const int64_t* tags, const float* w) {
  if (r->size + n > r->ring) return -1; /* would wait on cvSize_ :47 */
  int start = r->tail;
  int end = (r->tail + n) % r->ring;
  r->tail = end;
  r->size += n;
  volatile float sum = 0; /* float accumulation :58,65 */
  for (int i = 0; i < n; ++i) {
    int j = (start + i) % r->ring;
    r->elements[j] = tags ? tags[i] : 0;
    r->weights[j] = w[i];
    sum += w[i];
  }
  r->safe_tail = end;
  r->safe_size += n;
  r->sum += sum; /* double += float :73 */
  r->num_add += n; /* :190 */
  return 0;
}

/* add :186-191 */
int oracle_replay_add(oracle_replay* r, int n, const int64_t* tags, const float* priority) {
  float* w = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) w[i] = (r->alpha == 1.0f) ? priority[i] : powf(priority[i], r->alpha);
  int rc = oracle_replay_add_w(r, n, tags, w);
  free(w);
  return rc;
}

/* blockPop :84-103 */
int oracle_cqueue_pop(oracle_replay* r, int n) {
  double diff = 0;
  int head = r->head;
  for (int i = 0; i < n; ++i) {
    diff -= r->weights[head];
    r->evicted[head] = 1;
    head = (head + 1) % r->ring;
  }
  r->sum += diff;
  r->head = head;
  r->safe_size -= n;
  r->size -= n;
  return 0;
}

static void ensure_taps(oracle_replay* r, int batch) {
  if (batch <= r->tap_cap) return;
  r->last_targets = (float*)realloc(r->last_targets, sizeof(float) * (size_t)batch);
  r->last_raw_w = (float*)realloc(r->last_raw_w, sizeof(float) * (size_t)batch);
  r->sampled_ids = (int32_t*)realloc(r->sampled_ids, sizeof(int32_t) * (size_t)batch);
  r->tap_cap = batch;
}

/* sample :202-212 + sample_ :258-328 */
int oracle_replay_sample(oracle_replay* r, int batch, int32_t* ids, int64_t* tags, float* is_w) {
  if (r->n_sampled != 0) return -2; /* :203-206 */
  ensure_taps(r, batch);

  volatile float sum = (float)r->sum; /* safeSize(&sum) narrows :30-36 */
  int size = r->safe_size;
  volatile float segment = sum / (float)batch; /* :264 */

  double acc = 0;
  int next = 0;
  float w = 0;
  int id = 0;
  for (int i = 0; i < batch; ++i) {
    volatile float u = oracle_uniform_float(&r->rng, 0.0f, segment); /* :267,279 */
    volatile float off = (float)i * segment;
    volatile float rnd = u + off;
    volatile float cap = sum - 0.2f;
    if (cap < rnd) rnd = cap; /* std::min(sum - 0.2f, rand) :280 */
    r->last_targets[i] = rnd;
    int found = 0;
    while (next <= size) {
      if (acc > 0 && acc >= (double)rnd) { /* :286 */
        int phys = (r->head + (next - 1)) % r->ring;
        r->evicted[phys] = 0; /* getElementAndMark :124-128 */
        if (tags) tags[i] = r->elements[phys];
        r->last_raw_w[i] = w;
        ids[i] = id;
        found = 1;
        break;
      }
      if (next == size) return -3; /* :297-302 */
      id = (r->head + next) % r->ring; /* getWeight :130-134 */
      w = r->weights[id];
      acc += w;
      ++next;
    }
    if (!found) return -3;
  }

  size = r->size; /* storage_.size() re-read :312 */
  if (size > r->capacity_) oracle_cqueue_pop(r, size - r->capacity_); /* :313-315 */

  /* IS weights :320-322 (float tensor ops; pow is ATen/SLEEF in the reference) */
  float mx = -INFINITY;
  for (int i = 0; i < batch; ++i) {
    volatile float q = r->last_raw_w[i] / sum;
    volatile float s = (float)size * q;
    float p;
    if (r->beta == 1.0f) {
      p = 1.0f / s; /* ATen special-cases exponent -1 as reciprocal */
    } else {
      p = powf(s, -r->beta);
    }
    is_w[i] = p;
    if (p > mx) mx = p;
  }
  for (int i = 0; i < batch; ++i) is_w[i] = is_w[i] / mx;

  memcpy(r->sampled_ids, ids, sizeof(int32_t) * (size_t)batch);
  r->n_sampled = batch;
  return 0;
}

const float* oracle_replay_last_targets(const oracle_replay* r) { return r->last_targets; }
const float* oracle_replay_last_raw_w(const oracle_replay* r) { return r->last_raw_w; }

/* update :105-119 */
int oracle_replay_update_w(oracle_replay* r, int n, const float* w) {
  if (n != r->n_sampled) return -4; /* :237 */
  double diff = 0;
  for (int i = 0; i < n; ++i) {
    int id = r->sampled_ids[i];
    if (r->evicted[id]) continue;
    volatile float d = w[i] - r->weights[id]; /* float - float :113 */
    diff += d;
    r->weights[id] = w[i];
  }
  r->sum += diff;
  r->n_sampled = 0; /* sampledIds_.clear() :244 */
  return 0;
}

/* updatePriority :235-245 */
int oracle_replay_update(oracle_replay* r, int n, const float* priority) {
  float* w = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) w[i] = (r->alpha == 1.0f) ? priority[i] : powf(priority[i], r->alpha);
  int rc = oracle_replay_update_w(r, n, w);
  free(w);
  return rc;
}

int oracle_replay_size(const oracle_replay* r) { return r->safe_size; }
int oracle_replay_full_size(const oracle_replay* r) { return r->size; }
int64_t oracle_replay_num_add(const oracle_replay* r) { return r->num_add; }
int oracle_replay_head(const oracle_replay* r) { return r->head; }
int oracle_replay_tail(const oracle_replay* r) { return r->tail; }
int oracle_replay_ring(const oracle_replay* r) { return r->ring; }
double oracle_replay_sum(const oracle_replay* r) { return r->sum; }
const float* oracle_replay_weights(const oracle_replay* r) { return r->weights; }
const uint8_t* oracle_replay_evicted(const oracle_replay* r) { return r->evicted; }

/* the bare scan of :266-308 */
void oracle_scan_search(const float* w, int n, const float* targets, int nt, int32_t* out_idx,
                        double* out_acc) {
  double acc = 0;
  int next = 0;
  for (int i = 0; i < nt; ++i) {
    float rnd = targets[i];
    out_idx[i] = -1;
    if (out_acc) out_acc[i] = 0;
    while (next <= n) {
      if (acc > 0 && acc >= (double)rnd) {
        out_idx[i] = next - 1;
        if (out_acc) out_acc[i] = acc;
        break;
      }
      if (next == n) break;
      acc += w[next];
      ++next;
    }
  }
}
