// Golden-vector generator for the prioritized replay path.
//
// TEST INFRASTRUCTURE, build-time-here only: this harness is compiled against the
// real reference headers with -I/root/reference (see oracle/Makefile, target
// _ref/replay_kat).  Neither the reference sources nor this binary travel to the
// GPU box; only the JSON vectors it prints are committed (tests/golden/).
//
// It drives rela::PrioritizedReplay<FFTransition> (rela/prioritized_replay.h:173-348)
// through its public C++ API (add :186-200, sample :202-233, updatePriority :235-245)
// with a script read from stdin, and prints one JSON object per op.
//
// script grammar (one op per line, floats as 8-hex-digit IEEE-754 bit patterns):
//   new <capacity> <seed> <alpha> <beta> <prefetch>
//   add <n> <tag0> <p_0> ... <p_{n-1}>       (transition i carries reward == tag0+i)
//   sample <batch>
//   update <n> <p_0> ... <p_{n-1}>
#include <cstdint>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include <torch/extension.h>
#include <torch/torch.h>

// expose the bookkeeping (head/tail/sum/ids) for deep pinning of the restatement
#define private public
#include "rela/prioritized_replay.h"
#undef private

using namespace rela;

static float hex2f(const std::string& s) {
  uint32_t u = (uint32_t)std::stoul(s, nullptr, 16);
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

static std::string f2hex(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  std::ostringstream os;
  os << std::hex << std::setw(8) << std::setfill('0') << u;
  return os.str();
}

static std::string d2hex(double d) {
  uint64_t u;
  std::memcpy(&u, &d, 8);
  std::ostringstream os;
  os << std::hex << std::setw(16) << std::setfill('0') << u;
  return os.str();
}

static void dumpState(FFPrioritizedReplay& r) {
  auto& q = r.storage_;
  std::cout << "\"head\":" << q.head_ << ",\"tail\":" << q.tail_ << ",\"size\":" << q.size_
            << ",\"safe_size\":" << q.safeSize_ << ",\"sum\":\"" << d2hex(q.sum_)
            << "\",\"num_add\":" << r.numAdd();
}

int main() {
  std::unique_ptr<FFPrioritizedReplay> replay;
  std::string line;
  while (std::getline(std::cin, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream is(line);
    std::string op;
    is >> op;
    if (op == "new") {
      int cap, seed, prefetch;
      std::string a, b;
      is >> cap >> seed >> a >> b >> prefetch;
      replay = std::make_unique<FFPrioritizedReplay>(cap, seed, hex2f(a), hex2f(b), prefetch);
      std::cout << "{\"op\":\"new\",\"ring\":" << replay->storage_.capacity << "}" << std::endl;
    } else if (op == "add") {
      int n;
      long tag0;
      is >> n >> tag0;
      auto prio = torch::zeros({n}, torch::kFloat32);
      auto acc = prio.accessor<float, 1>();
      for (int i = 0; i < n; ++i) {
        std::string p;
        is >> p;
        acc[i] = hex2f(p);
      }
      auto tag = torch::arange(tag0, tag0 + n).to(torch::kFloat32);
      TensorDict empty = {};
      FFTransition t(empty, empty, tag, tag, tag, empty);
      replay->add(t, prio);
      // the exponentiated weights exactly as add() computes them (ATen pow, :188)
      auto pw = torch::pow(prio, replay->alpha_);
      std::cout << "{\"op\":\"add\",\"stored_w\":[";
      for (int i = 0; i < n; ++i) std::cout << (i ? "," : "") << "\"" << f2hex(pw[i].item<float>()) << "\"";
      std::cout << "],";
      dumpState(*replay);
      std::cout << "}" << std::endl;
    } else if (op == "sample") {
      int bs;
      is >> bs;
      FFTransition batch;
      torch::Tensor w;
      std::tie(batch, w) = replay->sample(bs, "cpu");
      std::cout << "{\"op\":\"sample\",\"ids\":[";
      for (int i = 0; i < bs; ++i) std::cout << (i ? "," : "") << replay->sampledIds_[i];
      std::cout << "],\"tags\":[";
      for (int i = 0; i < bs; ++i) std::cout << (i ? "," : "") << (long)batch.reward[i].item<float>();
      std::cout << "],\"w\":[";
      for (int i = 0; i < bs; ++i) std::cout << (i ? "," : "") << "\"" << f2hex(w[i].item<float>()) << "\"";
      std::cout << "],";
      dumpState(*replay);
      std::cout << "}" << std::endl;
    } else if (op == "update") {
      int n;
      is >> n;
      auto prio = torch::zeros({n}, torch::kFloat32);
      auto acc = prio.accessor<float, 1>();
      for (int i = 0; i < n; ++i) {
        std::string p;
        is >> p;
        acc[i] = hex2f(p);
      }
      replay->updatePriority(prio);
      // per-occurrence weights exactly as updatePriority() computes them (ATen pow, :239)
      auto pw = torch::pow(prio, replay->alpha_);
      std::cout << "{\"op\":\"update\",\"stored_w\":[";
      for (int i = 0; i < n; ++i) std::cout << (i ? "," : "") << "\"" << f2hex(pw[i].item<float>()) << "\"";
      std::cout << "],";
      dumpState(*replay);
      std::cout << "}" << std::endl;
    } else {
      std::cerr << "bad op: " << op << std::endl;
      return 2;
    }
  }
  return 0;
}
