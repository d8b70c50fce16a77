// rela/env.h -- the environment plug-in boundary (drop-in for the reference's rela/env.h:13-102).
//
// A native env module (the reference's `atari`, our `synth_atari`) subclasses rela::Env and
// registers it against rela.Env with py::class_<MyEnv, rela::Env, std::shared_ptr<MyEnv>>.
// The three pure virtuals and their order are the ABI.
//
// VectorEnv differs from the reference internally: instead of torch::stack-ing K fresh tensors
// per step it keeps ONE persistent page-locked batch tensor per observation key and copies each
// env's row into it, so the actor can DMA the whole [K,4,84,84] block to its HBM history slot
// with a single asynchronous copy.  The TensorDicts it returns alias those buffers and are
// valid until the next reset()/step() call.  r4: the per-env path issues no tensor operation (row
// addresses and per-env action views are cached: a torch op costs 1-3 us through the dispatcher and
// there were six per env-step), envs implementing rela::FrameRowEnv render into their row directly,
// and when ALL envs declare a sliding stack the batch carries one extra key, "__stack_restart"
// (u8[K]: 1 = the row's stack was just restarted by reset()), which tells this module's actors
// that only plane 3 of each row is new.
#pragma once
#include <memory>
#include <tuple>
#include <vector>

#include "rela/frame_row_env.h"
#include "rela/types.h"

namespace rela {

class Env {
 public:
  Env() = default;
  virtual ~Env() = default;

  // first observation of a new episode
  virtual TensorDict reset() = 0;
  // (observation, reward, terminal) after applying `action` ({"a": 0-dim or [1] int64})
  virtual std::tuple<TensorDict, float, bool> step(const TensorDict& action) = 0;
  virtual bool terminated() const = 0;
};

class VectorEnv {
 public:
  VectorEnv() = default;
  virtual ~VectorEnv() = default;

  void append(std::shared_ptr<Env> env);
  int size() const { return (int)envs_.size(); }

  // Resets only the envs whose episode ended (all of them on the first call) and returns the
  // batched observation; rows of running envs keep the observation of their last step.
  virtual TensorDict reset(const TensorDict& previous);

  // Steps every env with its row of `action`; returns (obs batch, reward f32[K], terminal bool[K]).
  virtual std::tuple<TensorDict, torch::Tensor, torch::Tensor> step(const TensorDict& action);

  virtual bool anyTerminated() const;
  virtual bool allTerminated() const;

 private:
  struct KeyRows {
    std::string key;
    uint8_t* base = nullptr;
    int64_t rowBytes = 0;
    torch::ScalarType dtype = torch::kFloat32;
  };
  void storeRow(const TensorDict& obs, int row);
  void createBatch(const TensorDict& firstObs);

  std::vector<std::shared_ptr<Env>> envs_;
  std::vector<FrameRowEnv*> frameEnvs_;  // envs_[i]'s optional extension (nullptr: copying path)
  TensorDict batch_;                     // persistent, page-locked when a GPU is present
  std::vector<KeyRows> rows_;            // raw row addresses of batch_'s tensors
  bool sliding_ = false;                 // every env declared a sliding stack
  uint8_t* restart_ = nullptr;           // "__stack_restart" flags (aliases batch_'s tensor)
  torch::Tensor reward_, terminal_, actionBuf_;
  std::vector<TensorDict> envAction_;    // per env: {"a": 0-dim view of actionBuf_[i]}
};

}  // namespace rela
