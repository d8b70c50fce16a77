// CPU shim: executes the SeqPlan of rela_amd/csrc/r2d2_seq_core.h on plain host arrays with the
// same row-move semantics the device kernels of actor_r2d2.hip implement, so the host logic can
// be compared with the oracle / reference traces without a GPU.  TEST INFRASTRUCTURE.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../rela_amd/csrc/r2d2_seq_core.h"

using namespace rela_amd;

namespace {
struct Slot {
  int64_t tag, action;
  float reward, bootstrap;
  uint8_t terminal;
};
struct Shim {
  int K, n, seq, burn, T;
  SeqBook book;
  std::vector<Slot> win;    // [K][T]
  std::vector<float> prio;  // [K][seq+n]
  std::vector<float> h0, nh0;
  Shim(int K, int n, int seq, int burn)
      : K(K), n(n), seq(seq), burn(burn), T(burn + seq + n), book(K, n, seq, burn), win((size_t)K * T),
        prio((size_t)K * (seq + n), 0.f), h0(K, 0.f), nh0(K, 0.f) {}
};
void pad(Shim* s, const SeqRange& r) {
  for (int j = r.begin; j < r.end; ++j) {
    Slot& w = s->win[(size_t)r.env * s->T + j];
    std::memset(&w, 0, sizeof(w));
    w.terminal = 1;
    if (r.zero_prio) s->prio[(size_t)r.env * (s->seq + s->n) + (j - s->burn)] = 0.f;
  }
}
}  // namespace

extern "C" {
void* shim_r2d2_new(int K, int n, int seq, int burn) { return new Shim(K, n, seq, burn); }
void shim_r2d2_free(void* p) { delete (Shim*)p; }

// returns the number of emitted sequences; outputs as in oracle_r2d2buf_pop
int shim_r2d2_step(void* p, const int64_t* tag, const int64_t* action, const float* reward, const float* bootstrap,
                   const uint8_t* terminal, const float* priority, const float* hid, float* out_len, float* out_h0,
                   int64_t* otag, int64_t* oact, float* orew, uint8_t* oterm, float* oboot, float* oprio) {
  Shim* s = (Shim*)p;
  SeqPlan plan;
  s->book.step(terminal, &plan);
  const int T = s->T, P = s->seq + s->n;
  for (const auto& r : plan.front_pad) pad(s, r);
  for (int i = 0; i < s->K; ++i) {  // "r2d2_write_step"
    if (plan.flags[i] & 1) s->h0[i] = hid[i];
    if (plan.flags[i] & 2) s->nh0[i] = hid[i];
    const int j = plan.write_slot[i];
    s->win[(size_t)i * T + j] = Slot{tag[i], action[i], reward[i], bootstrap[i], terminal[i]};
    s->prio[(size_t)i * P + (j - s->burn)] = priority[i];
  }
  for (const auto& r : plan.tail_pad) pad(s, r);
  if (!plan.can_pop) return 0;
  // "r2d2_collect_prio": priority rows of every emit from the PRE-carry state
  int q = 0;
  for (const auto& e : plan.emits) {
    const float* po = &s->prio[(size_t)e.env * P];
    for (int j = 0; j < s->seq; ++j) {
      float v;
      if (!e.second) v = po[j];
      else if (j < s->n) v = (j >= s->burn) ? po[s->seq + j] : po[j];
      else v = 0.f;
      oprio[(size_t)q * s->seq + j] = v;
    }
    out_len[q] = (float)e.len;
    ++q;
  }
  // rows, in stream order: first emits -> carry -> carry pads -> second emits
  auto copy_rows = [&](int qi, int env) {
    for (int j = 0; j < T; ++j) {
      const Slot& w = s->win[(size_t)env * T + j];
      otag[(size_t)qi * T + j] = w.tag;
      oact[(size_t)qi * T + j] = w.action;
      orew[(size_t)qi * T + j] = w.reward;
      oterm[(size_t)qi * T + j] = w.terminal;
      oboot[(size_t)qi * T + j] = w.bootstrap;
    }
  };
  q = 0;
  for (const auto& e : plan.emits) {
    if (!e.second) {
      copy_rows(q, e.env);
      out_h0[q] = s->h0[e.env];
    }
    ++q;
  }
  for (int env : plan.carry_env) {  // "r2d2_carry"
    for (int j = 0; j < s->burn + s->n; ++j) s->win[(size_t)env * T + j] = s->win[(size_t)env * T + s->seq + j];
    float* po = &s->prio[(size_t)env * P];
    for (int j = s->burn; j < s->n; ++j) po[j] = po[s->seq + j];
    s->h0[env] = s->nh0[env];
  }
  for (const auto& r : plan.carry_pad) pad(s, r);
  q = 0;
  for (const auto& e : plan.emits) {
    if (e.second) {
      copy_rows(q, e.env);
      out_h0[q] = s->h0[e.env];
    }
    ++q;
  }
  return q;
}
}
