/* oracle/r2d2_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Restates rela::R2D2TransitionBuffer, rela/r2d2_actor.h:10-187, on scalar tags: one window of
 * burnin + seqLen + multiStep slots per env.
 *   push  :29-87   episode start -> front-pad `burnin` slots with padLike (types.cc:69-80: zeros,
 *                  terminal = 1); remember the hidden state seen at window index seqLen (:65-68);
 *                  store the step priority at index - burnin (:71); on terminal or full window
 *                  tail-pad with padLike / priority 0 and raise canPop (:79-86).
 *   pop   :93-170  every env with a finished window emits (window, h0, len = min(L, burnin+seqLen)),
 *                  priorities[0, seqLen); if slot len-1 is terminal the env restarts (:115-118),
 *                  else slots [seqLen, seqLen+burnin+multiStep) move to the front, h0 <- the hidden
 *                  state captured at index seqLen, and a terminal inside the carried part emits a
 *                  second, short sequence at once (:119-160).
 * and R2D2Agent.aggregate_priority, pyrela/r2d2.py:103-120.
 */
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

typedef struct {
  int64_t tag, action;
  float reward, bootstrap;
  uint8_t terminal;
} slot_t;

struct oracle_r2d2buf {
  int K, n, seq, burn, T;
  int* next_idx;
  int* len;
  float *h0, *next_h0;
  slot_t* slots; /* [K][T] */
  float* prio;   /* [K][seq+n] */
  int can_pop;
};

oracle_r2d2buf* oracle_r2d2buf_new(int K, int n, int seq, int burn) {
  oracle_r2d2buf* b = (oracle_r2d2buf*)calloc(1, sizeof(*b));
  b->K = K;
  b->n = n;
  b->seq = seq;
  b->burn = burn;
  b->T = burn + seq + n;
  b->next_idx = (int*)calloc((size_t)K, sizeof(int));
  b->len = (int*)calloc((size_t)K, sizeof(int));
  b->h0 = (float*)calloc((size_t)K, sizeof(float));
  b->next_h0 = (float*)calloc((size_t)K, sizeof(float));
  b->slots = (slot_t*)calloc((size_t)K * b->T, sizeof(slot_t));
  b->prio = (float*)calloc((size_t)K * (seq + n), sizeof(float));
  return b;
}

void oracle_r2d2buf_free(oracle_r2d2buf* b) {
  if (!b) return;
  free(b->next_idx);
  free(b->len);
  free(b->h0);
  free(b->next_h0);
  free(b->slots);
  free(b->prio);
  free(b);
}

static slot_t pad_slot(void) {
  slot_t p;
  memset(&p, 0, sizeof(p));
  p.terminal = 1;
  return p;
}

int oracle_r2d2buf_push(oracle_r2d2buf* b, const int64_t* tag, const int64_t* action, const float* reward,
                        const float* bootstrap, const uint8_t* terminal, const float* priority,
                        const float* hid_tag) {
  const int T = b->T;
  for (int i = 0; i < b->K; ++i) {
    slot_t* w = b->slots + (size_t)i * T;
    float* p = b->prio + (size_t)i * (b->seq + b->n);
    if (b->next_idx[i] == 0) {
      b->h0[i] = hid_tag[i]; /* :41 (asserted all-zero upstream) */
      while (b->next_idx[i] < b->burn) w[b->next_idx[i]++] = pad_slot(); /* :46-49 */
    }
    int idx = b->next_idx[i];
    if (idx == b->seq) b->next_h0[i] = hid_tag[i]; /* :65-68 */
    slot_t t;
    t.tag = tag[i];
    t.action = action[i];
    t.reward = reward[i];
    t.bootstrap = bootstrap[i];
    t.terminal = terminal[i];
    w[idx] = t;
    p[idx - b->burn] = priority[i];
    b->next_idx[i] = ++idx;
    if (!t.terminal && idx < T) continue; /* :74-76 */
    b->len[i] = idx;                      /* :79 */
    while (b->next_idx[i] < T) {
      w[b->next_idx[i]] = pad_slot();
      p[b->next_idx[i] - b->burn] = 0.0f;
      ++b->next_idx[i];
    }
    b->can_pop = 1;
  }
  return b->can_pop;
}

static int emit(const oracle_r2d2buf* b, int i, float len, int q, float* out_len, float* out_h0, int* out_env,
                int64_t* tag, int64_t* action, float* reward, uint8_t* terminal, float* bootstrap, float* prio) {
  const int T = b->T;
  const slot_t* w = b->slots + (size_t)i * T;
  out_len[q] = len;
  out_h0[q] = b->h0[i];
  if (out_env) out_env[q] = i;
  for (int j = 0; j < T; ++j) {
    tag[(size_t)q * T + j] = w[j].tag;
    action[(size_t)q * T + j] = w[j].action;
    reward[(size_t)q * T + j] = w[j].reward;
    terminal[(size_t)q * T + j] = w[j].terminal;
    bootstrap[(size_t)q * T + j] = w[j].bootstrap;
  }
  memcpy(prio + (size_t)q * b->seq, b->prio + (size_t)i * (b->seq + b->n), sizeof(float) * (size_t)b->seq); /* :110 */
  return q + 1;
}

int oracle_r2d2buf_pop(oracle_r2d2buf* b, float* out_len, float* out_h0, int* out_env, int64_t* tag,
                       int64_t* action, float* reward, uint8_t* terminal, float* bootstrap, float* prio) {
  const int T = b->T, seq = b->seq, burn = b->burn, n = b->n;
  int q = 0;
  for (int i = 0; i < b->K; ++i) {
    if (b->len[i] == 0) continue;
    slot_t* w = b->slots + (size_t)i * T;
    float* p = b->prio + (size_t)i * (seq + n);
    const int cap = burn + seq;
    const float len = (float)(b->len[i] < cap ? b->len[i] : cap); /* :106 */
    q = emit(b, i, len, q, out_len, out_h0, out_env, tag, action, reward, terminal, bootstrap, prio);
    if (w[(int)len - 1].terminal) { /* :115-118 */
      b->next_idx[i] = 0;
    } else {
      for (int j = 0; j < burn; ++j) w[j] = w[seq + j]; /* :120-124 */
      float len2 = -1;
      for (int j = burn; j < burn + n; ++j) { /* :128-136 */
        w[j] = w[seq + j];
        /* Reference quirk: `seqPriority[j] = seqPriority[k]` (:131) indexes the (seq+n)-long priority
         * vector with SLOT indices j, k = seq + j.  For k >= seq + n that is an out-of-bounds read;
         * every entry it lands in (index >= n) is rewritten by a later push or pad before it can be
         * emitted, so any value is equivalent -- we store 0.  In-bounds copies are kept verbatim:
         * the net effect is p[j'] = p[seq + j'] for burn <= j' < n and STALE p[j'] for j' < burn. */
        p[j] = (seq + j < seq + n) ? p[seq + j] : 0.0f;
        if (w[j].terminal && len2 == -1) len2 = (float)(j + 1);
      }
      b->next_idx[i] = burn + n;
      b->h0[i] = b->next_h0[i]; /* :139 */
      if (len2 != -1) {
        const slot_t ref = w[b->next_idx[i] - 1];
        (void)ref;
        while (b->next_idx[i] < T) { /* :144-148 */
          w[b->next_idx[i]] = pad_slot();
          p[b->next_idx[i] - burn] = 0.0f;
          ++b->next_idx[i];
        }
        q = emit(b, i, len2, q, out_len, out_h0, out_env, tag, action, reward, terminal, bootstrap, prio);
        b->next_idx[i] = 0; /* :157 */
      }
    }
    b->len[i] = 0;
  }
  b->can_pop = 0;
  return q;
}

/* R2D2Agent.aggregate_priority, pyrela/r2d2.py:103-120, for nseq rows of seq_len_const entries */
void oracle_r2d2_aggregate(int nseq, int seq_len_const, int burn_in, float eta, const float* priority,
                           const float* seq_len, float* out) {
  for (int q = 0; q < nseq; ++q) {
    float sum = 0.0f, mx = -1e30f;
    for (int t = 0; t < seq_len_const; ++t) {
      const float m = ((float)t < seq_len[q]) ? 1.0f : 0.0f;
      const float v = priority[(size_t)q * seq_len_const + t] * m;
      sum += v;
      if (v > mx) mx = v;
    }
    const float mean = sum / (seq_len[q] - (float)burn_in);
    volatile float a = eta * mx;
    volatile float bterm = (1.0f - eta) * mean;
    out[q] = a + bterm;
  }
}
