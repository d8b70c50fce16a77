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
// valid until the next reset()/step() call.
#pragma once
#include <memory>
#include <tuple>
#include <vector>

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

  void append(std::shared_ptr<Env> env) { envs_.push_back(std::move(env)); }
  int size() const { return (int)envs_.size(); }

  // Resets only the envs whose episode ended (all of them on the first call) and returns the
  // batched observation; rows of running envs keep the observation of their last step.
  virtual TensorDict reset(const TensorDict& previous);

  // Steps every env with its row of `action`; returns (obs batch, reward f32[K], terminal bool[K]).
  virtual std::tuple<TensorDict, torch::Tensor, torch::Tensor> step(const TensorDict& action);

  virtual bool anyTerminated() const;
  virtual bool allTerminated() const;

 private:
  void storeRow(const TensorDict& obs, int row);

  std::vector<std::shared_ptr<Env>> envs_;
  TensorDict batch_;  // persistent, page-locked when a GPU is present
  torch::Tensor reward_, terminal_;
};

}  // namespace rela
