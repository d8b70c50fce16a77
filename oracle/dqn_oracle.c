/* oracle/dqn_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * fp32 restatement of the Ape-X network and agent arithmetic:
 *   AtariFFNet.forward   pyrela/net.py:42-55   (s/255 -> 3x conv+ReLU -> fc512+ReLU -> v, a)
 *   AtariFFNet.duel      pyrela/net.py:33-39   (q = v + a*legal - mean_A(a*legal))
 *   ApexAgent.greedy_act pyrela/apex.py:48-54  (batch-global q.min(), first-index argmax)
 *   ApexAgent.td_err     pyrela/apex.py:30-45  (double-DQN n-step target)
 *   compute_priority     pyrela/apex.py:68-78  (|td_err|)
 * Convolutions are direct loops, k-sequential accumulation per output element, weights
 * re-laid [k][oc] so the inner loop is contiguous.  The library kernels the reference
 * dispatches to (MKL-DNN / MIOpen) sum in a different order, so agreement with the
 * reference is to fp32 tolerance (1e-4 abs/rel, tests/test_oracle_golden.py), not bitwise.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define FS 4
#define IH 84
#define IW 84
#define C1 32
#define O1 20
#define C2 64
#define O2 9
#define C3 64
#define O3 7
#define FLAT 3136
#define HID 512

typedef struct {
  float* w1t; /* [4*8*8][32]   */
  float* w2t; /* [32*4*4][64]  */
  float* w3t; /* [64*3*3][64]  */
  float* l1t; /* [3136][512]   */
} packed_w;

static void transpose_kxo(const float* w, int oc, int k, float* out) {
  for (int o = 0; o < oc; ++o)
    for (int i = 0; i < k; ++i) out[(size_t)i * oc + o] = w[(size_t)o * k + i];
}

static packed_w* pack(const oracle_ffnet* net) {
  packed_w* p = (packed_w*)malloc(sizeof(*p));
  p->w1t = (float*)malloc(sizeof(float) * 256 * C1);
  p->w2t = (float*)malloc(sizeof(float) * 512 * C2);
  p->w3t = (float*)malloc(sizeof(float) * 576 * C3);
  p->l1t = (float*)malloc(sizeof(float) * FLAT * HID);
  transpose_kxo(net->c1w, C1, 256, p->w1t);
  transpose_kxo(net->c2w, C2, 512, p->w2t);
  transpose_kxo(net->c3w, C3, 576, p->w3t);
  transpose_kxo(net->l1w, HID, FLAT, p->l1t);
  return p;
}

static void unpack(packed_w* p) {
  free(p->w1t);
  free(p->w2t);
  free(p->w3t);
  free(p->l1t);
  free(p);
}

/* generic direct conv, NCHW in / NCHW out, bias + ReLU */
static void conv_relu(const float* in, int cin, int ih, int iw, const float* wt, const float* bias,
                      int cout, int kh, int kw, int stride, int oh, int ow, float* out) {
  float acc[64];
  for (int oy = 0; oy < oh; ++oy) {
    for (int ox = 0; ox < ow; ++ox) {
      for (int o = 0; o < cout; ++o) acc[o] = 0.0f;
      int k = 0;
      for (int c = 0; c < cin; ++c) {
        for (int y = 0; y < kh; ++y) {
          const float* row = in + ((size_t)c * ih + (size_t)(oy * stride + y)) * iw + ox * stride;
          for (int x = 0; x < kw; ++x, ++k) {
            float v = row[x];
            const float* wr = wt + (size_t)k * cout;
            for (int o = 0; o < cout; ++o) acc[o] += v * wr[o];
          }
        }
      }
      for (int o = 0; o < cout; ++o) {
        float r = acc[o] + bias[o];
        out[((size_t)o * oh + oy) * ow + ox] = r > 0.0f ? r : 0.0f;
      }
    }
  }
}

static void forward_one(const oracle_ffnet* net, const packed_w* p, const uint8_t* s,
                        const float* legal, float* q) {
  const int A = net->num_action;
  float* x0 = (float*)malloc(sizeof(float) * FS * IH * IW);
  float* x1 = (float*)malloc(sizeof(float) * C1 * O1 * O1);
  float* x2 = (float*)malloc(sizeof(float) * C2 * O2 * O2);
  float* x3 = (float*)malloc(sizeof(float) * FLAT);
  float h[HID];
  for (int i = 0; i < FS * IH * IW; ++i) x0[i] = (float)s[i] / 255.0f; /* net.py:46 */
  conv_relu(x0, FS, IH, IW, p->w1t, net->c1b, C1, 8, 8, 4, O1, O1, x1);
  conv_relu(x1, C1, O1, O1, p->w2t, net->c2b, C2, 4, 4, 2, O2, O2, x2);
  conv_relu(x2, C2, O2, O2, p->w3t, net->c3b, C3, 3, 3, 1, O3, O3, x3);
  /* linear 3136 -> 512 + ReLU (net.py:27-29,50) */
  for (int o = 0; o < HID; ++o) h[o] = 0.0f;
  for (int k = 0; k < FLAT; ++k) {
    float v = x3[k];
    const float* wr = p->l1t + (size_t)k * HID;
    for (int o = 0; o < HID; ++o) h[o] += v * wr[o];
  }
  for (int o = 0; o < HID; ++o) {
    float r = h[o] + net->l1b[o];
    h[o] = r > 0.0f ? r : 0.0f;
  }
  /* heads (net.py:30-31,51-52) */
  float v = 0.0f;
  for (int k = 0; k < HID; ++k) v += h[k] * net->vw[k];
  v += net->vb[0];
  float a[64];
  float mean = 0.0f;
  for (int j = 0; j < A; ++j) {
    float t = 0.0f;
    for (int k = 0; k < HID; ++k) t += h[k] * net->aw[(size_t)j * HID + k];
    t += net->ab[j];
    a[j] = t * legal[j]; /* legal_a :37 */
    mean += a[j];
  }
  mean /= (float)A; /* mean over A, not over #legal :38 */
  for (int j = 0; j < A; ++j) q[j] = v + a[j] - mean;
  free(x0);
  free(x1);
  free(x2);
  free(x3);
}

typedef struct {
  const oracle_ffnet* net;
  const packed_w* p;
  const uint8_t* s;
  const float* legal;
  float* q;
  int lo, hi;
} job_t;

static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  const int A = j->net->num_action;
  for (int n = j->lo; n < j->hi; ++n)
    forward_one(j->net, j->p, j->s + (size_t)n * FS * IH * IW, j->legal + (size_t)n * A,
                j->q + (size_t)n * A);
  return NULL;
}

static int g_threads = 1;
void oracle_set_threads(int n) { g_threads = n < 1 ? 1 : (n > 256 ? 256 : n); }

void oracle_ffnet_forward(const oracle_ffnet* net, int N, const uint8_t* s, const float* legal,
                          float* q) {
  packed_w* p = pack(net);
  int T = g_threads < N ? g_threads : (N > 0 ? N : 1);
  pthread_t th[256];
  job_t jobs[256];
  for (int t = 0; t < T; ++t) {
    jobs[t] = (job_t){net, p, s, legal, q, (int)((long)N * t / T), (int)((long)N * (t + 1) / T)};
    if (T > 1) pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  if (T > 1) {
    for (int t = 0; t < T; ++t) pthread_join(th[t], NULL);
  } else {
    worker(&jobs[0]);
  }
  unpack(p);
}

void oracle_greedy(int N, int A, const float* q, const float* legal, int64_t* action) {
  float mn = INFINITY;
  for (int i = 0; i < N * A; ++i)
    if (q[i] < mn) mn = q[i]; /* q.min() over the whole batch, apex.py:51 */
  for (int n = 0; n < N; ++n) {
    int best = 0;
    float bv = -INFINITY;
    for (int j = 0; j < A; ++j) {
      volatile float t = 1.0f + q[n * A + j];
      volatile float u = t - mn;
      float lq = u * legal[n * A + j]; /* (1 + q - q.min()) * legal_move */
      if (lq > bv) {                   /* first maximal index, as torch argmax */
        bv = lq;
        best = j;
      }
    }
    action[n] = best;
  }
}

void oracle_apex_priority(const oracle_ffnet* online, const oracle_ffnet* target, int N,
                          const uint8_t* s, const float* legal, const int64_t* action,
                          const float* reward, const float* bootstrap, const uint8_t* next_s,
                          const float* next_legal, float gamma_n, float* td_err_signed,
                          float* priority) {
  const int A = online->num_action;
  float* q = (float*)malloc(sizeof(float) * (size_t)N * A);
  float* qn = (float*)malloc(sizeof(float) * (size_t)N * A);
  float* qt = (float*)malloc(sizeof(float) * (size_t)N * A);
  int64_t* na = (int64_t*)malloc(sizeof(int64_t) * (size_t)N);
  oracle_ffnet_forward(online, N, s, legal, q);             /* apex.py:38 */
  oracle_ffnet_forward(online, N, next_s, next_legal, qn);  /* :41 via greedy_act */
  oracle_greedy(N, A, qn, next_legal, na);
  oracle_ffnet_forward(target, N, next_s, next_legal, qt);  /* :42 */
  for (int n = 0; n < N; ++n) {
    float qa = q[n * A + action[n]];
    float bq = qt[n * A + na[n]];
    volatile float g = bootstrap[n] * gamma_n; /* bootstrap * (gamma**n) * q, left to right :44 */
    volatile float gb = g * bq;
    volatile float tgt = reward[n] + gb;
    float e = tgt - qa;
    if (td_err_signed) td_err_signed[n] = e;
    if (priority) priority[n] = fabsf(e);
  }
  free(q);
  free(qn);
  free(qt);
  free(na);
}
