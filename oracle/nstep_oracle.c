/* oracle/nstep_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Restates MultiStepTransitionBuffer::popTransition, rela/dqn_actor.h:58-106, on flat
 * [n+1][K] histories (row 0 = oldest step t, row n = step t+n):
 *   bootstrap_i = 0 and next index = first step in [0,n) whose terminal flag is set (:73-87),
 *   reward_i    = Horner  acc = r_step + gamma*acc  from that step (or n-1) down to 0 (:90-98),
 *   terminal_i  = terminal flag of step 0 only (:66).
 * gamma*acc + r is evaluated un-fused (the canonical form chosen in SURVEY H8).
 */
#include "oracle.h"

void oracle_nstep_pop(int multi_step, int K, float gamma, const float* reward_hist,
                      const uint8_t* terminal_hist, float* out_reward, float* out_bootstrap,
                      uint8_t* out_terminal) {
  for (int i = 0; i < K; ++i) {
    float bootstrap = 1.0f;
    int next_idx = multi_step;
    for (int step = 0; step < multi_step; ++step) {
      if (terminal_hist[step * K + i]) {
        bootstrap = 0.0f;
        next_idx = step;
        break;
      }
    }
    int initial = (bootstrap != 0.0f) ? multi_step - 1 : next_idx;
    volatile float acc = 0.0f;
    for (int step = initial; step >= 0; --step) {
      volatile float prod = gamma * acc;
      acc = reward_hist[step * K + i] + prod;
    }
    out_reward[i] = acc;
    out_bootstrap[i] = bootstrap;
    out_terminal[i] = terminal_hist[i];
  }
}
