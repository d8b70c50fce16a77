// rela/types.h -- public value types of the `rela` module (drop-in for the reference's
// rela/types.h:10-73): what Python sees as rela.FFTransition / rela.RNNTransition and what a
// native env module exchanges with the runtime (TensorDict).
//
// In this engine transitions are NOT the unit of storage (the replay keeps structure-of-arrays
// rows in HBM, include/rela_amd.h); these structs only carry a sampled batch back to Python.
#pragma once
#include <torch/extension.h>

#include <string>
#include <unordered_map>

namespace rela {

using TensorDict = std::unordered_map<std::string, torch::Tensor>;

// Batch of feed-forward transitions, fields as bound in rela/pybind.cc:20-26.
struct FFTransition {
  TensorDict obs;          // {"s" u8[B,4,84,84], "eps" f32[B,1], "legal_move" f32[B,A]}
  TensorDict action;       // {"a" i64[B]}
  torch::Tensor reward;    // f32[B]  n-step return
  torch::Tensor terminal;  // bool[B]
  torch::Tensor bootstrap; // f32[B]
  TensorDict nextObs;      // obs n steps later
};

// Batch of sequences, fields as bound in rela/pybind.cc:28-35 (R2D2 path).
struct RNNTransition {
  TensorDict obs;  // [T,B,...]
  TensorDict h0;   // {"h0","c0"} f32[1,B,512]
  TensorDict action;
  torch::Tensor reward, terminal, bootstrap;  // [T,B]
  torch::Tensor seqLen;                       // f32[B]
};

}  // namespace rela
