// r2d2_seq_core.h -- host bookkeeping of the R2D2 sequence windows (no HIP in this header).
//
// Restates the INDEX logic of rela::R2D2TransitionBuffer (rela/r2d2_actor.h:10-187) and turns
// it into a plan of device row moves.  The payload (frames, actions, rewards, per-step
// priorities, hidden states) never visits the host: each env owns one window of
// T = burn_in + seq_len + multi_step slots per field in HBM, and the plan says which slot this
// step's transition goes to, which slot ranges become padding (padLike, types.cc:69-80), which
// envs emit a sequence, and which windows carry their tail to the front.
//
//   push  :29-87    pop  :93-170
// One reference quirk is kept on purpose (see oracle/r2d2_oracle.c): after a carry-over the first
// `multi_step` priority entries of the window are p[j] = p_old[seq_len + j] for j >= burn_in and
// the STALE p_old[j] for j < burn_in (:128-131 index the priority vector with slot indices).
#pragma once
#include <cstdint>
#include <vector>

namespace rela_amd {

struct SeqRange {
  int env, begin, end;  // slots [begin, end)
  int zero_prio;        // also zero the per-step priorities of these slots
};

struct SeqEmit {
  int env;
  int len;     // seqLen of the emitted RNNTransition (:106,133)
  int second;  // 1: the short sequence emitted after a carry-over (:141-157)
};

struct SeqPlan {
  // ---- push ----
  std::vector<int32_t> write_slot;  // [K] window slot for this step's transition
  std::vector<uint8_t> flags;       // [K] bit0: h0 <- hid (:41), bit1: next_h0 <- hid (:65-68)
  std::vector<SeqRange> front_pad;  // before the write (:46-49)
  std::vector<SeqRange> tail_pad;   // after the write (:80-84)
  bool can_pop = false;
  // ---- pop (only when can_pop), in reference order ----
  std::vector<SeqEmit> emits;      // all sequences, emission order = replay slot order
  std::vector<int32_t> carry_env;  // envs whose window tail moves to the front (:119-139)
  std::vector<SeqRange> carry_pad; // padding after the carry for envs that emit a second sequence
};

class SeqBook {
 public:
  SeqBook(int K, int multi_step, int seq_len, int burn_in)
      : K_(K), n_(multi_step), seq_(seq_len), burn_(burn_in), T_(burn_in + seq_len + multi_step),
        next_(K, 0), len_(K, 0), term_((size_t)K * (burn_in + seq_len + multi_step), 0) {}

  int window() const { return T_; }

  // One env-step for all K envs: `terminal[i]` is the terminal flag of the n-step transition
  // being pushed (= the env's terminal at time t).  Fills `plan` (push part and, if a window
  // finished, the pop part).
  void step(const uint8_t* terminal, SeqPlan* plan) {
    plan->write_slot.assign(K_, 0);
    plan->flags.assign(K_, 0);
    plan->front_pad.clear();
    plan->tail_pad.clear();
    plan->emits.clear();
    plan->carry_env.clear();
    plan->carry_pad.clear();
    bool can_pop = false;
    for (int i = 0; i < K_; ++i) {
      uint8_t* tf = &term_[(size_t)i * T_];
      if (next_[i] == 0) {
        plan->flags[i] |= 1;
        if (burn_ > 0) {
          plan->front_pad.push_back({i, 0, burn_, 0});
          for (int j = 0; j < burn_; ++j) tf[j] = 1;
        }
        next_[i] = burn_;
      }
      const int idx = next_[i];
      if (idx == seq_) plan->flags[i] |= 2;
      plan->write_slot[i] = idx;
      tf[idx] = terminal[i];
      next_[i] = idx + 1;
      if (!terminal[i] && next_[i] < T_) continue;
      len_[i] = next_[i];
      if (next_[i] < T_) {
        plan->tail_pad.push_back({i, next_[i], T_, 1});
        for (int j = next_[i]; j < T_; ++j) tf[j] = 1;
        next_[i] = T_;
      }
      can_pop = true;
    }
    plan->can_pop = can_pop;
    if (!can_pop) return;
    for (int i = 0; i < K_; ++i) {
      if (len_[i] == 0) continue;
      uint8_t* tf = &term_[(size_t)i * T_];
      const int L = len_[i] < burn_ + seq_ ? len_[i] : burn_ + seq_;
      plan->emits.push_back({i, L, 0});
      if (tf[L - 1]) {
        next_[i] = 0;
      } else {
        plan->carry_env.push_back(i);
        int len2 = -1;
        for (int j = 0; j < burn_ + n_; ++j) {
          tf[j] = tf[seq_ + j];
          if (j >= burn_ && tf[j] && len2 == -1) len2 = j + 1;
        }
        next_[i] = burn_ + n_;
        if (len2 != -1) {
          if (next_[i] < T_) {
            plan->carry_pad.push_back({i, next_[i], T_, 1});
            for (int j = next_[i]; j < T_; ++j) tf[j] = 1;
          }
          plan->emits.push_back({i, len2, 1});
          next_[i] = 0;
        }
      }
      len_[i] = 0;
    }
  }

 private:
  const int K_, n_, seq_, burn_, T_;
  std::vector<int> next_, len_;
  std::vector<uint8_t> term_;  // terminal flag of every window slot (padding = 1)
};

}  // namespace rela_amd
