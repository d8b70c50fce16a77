// rela/frame_row_env.h -- OPTIONAL extension of the env plug-in boundary (not in the reference's rela/env.h).
//
// An Env may additionally derive from rela::FrameRowEnv.  VectorEnv::append finds the interface with dynamic_cast
// and, once its page-locked batch tensor for "s" exists, hands every such env the address of ITS row: the env then
// renders its observation straight into the buffer the actor DMAs from (no per-step row copy), and may declare that
// its frame stack SLIDES -- every step() shifts the stack by one 84x84 plane and writes one new plane, reset() fills
// all four planes with the first frame of the episode, as GameState::computeFeature does (atari/game_state.h:53-82).
// For a VectorEnv whose envs all slide, only the newest plane of each row crosses PCIe (7,056 B instead of 28,224 B
// per env-step) and the actor shard completes the stacks on the device.
//
// Envs that do not implement it (anything compiled against the reference's three-virtual rela::Env) keep working
// through the copying path; the three pure virtuals of rela::Env and their order -- the ABI -- are untouched.
#pragma once
#include <cstdint>

namespace rela {

class FrameRowEnv {
 public:
  virtual ~FrameRowEnv() = default;
  // `row` = 4*84*84 page-locked bytes that stay valid for the VectorEnv's lifetime and already hold the env's current
  // observation; from now on the env's obs["s"] must alias them.
  virtual void bindFrameRow(uint8_t* row) = 0;
  // true: the stack slides by exactly one plane per step() and reset() repeats the first plane four times
  virtual bool slidingStack() const = 0;
};

}  // namespace rela
