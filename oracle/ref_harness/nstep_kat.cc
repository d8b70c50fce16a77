// Golden-vector generator for the n-step return buffer.
//
// TEST INFRASTRUCTURE, build-time-here only (see oracle/Makefile, target _ref/nstep_kat).
// Drives rela::MultiStepTransitionBuffer (rela/dqn_actor.h:15-124) exactly as
// DQNActor does (act :153-171 -> setRewardAndTerminal :174-177 -> postStep :181-190)
// and prints what popTransition (:58-106) returns.
//
// script grammar (floats as 8-hex-digit bit patterns):
//   new <multi_step> <batch> <gamma>
//   step <r_0> .. <r_{K-1}> <t_0> .. <t_{K-1}>       (t_i in {0,1})
#include <cstdint>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>

#include <torch/extension.h>
#include <torch/torch.h>

#include "rela/dqn_actor.h"

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

int main() {
  std::unique_ptr<MultiStepTransitionBuffer> buf;
  int K = 0;
  long stepIdx = 0;
  std::string line;
  while (std::getline(std::cin, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream is(line);
    std::string op;
    is >> op;
    if (op == "new") {
      int n;
      std::string g;
      is >> n >> K >> g;
      buf = std::make_unique<MultiStepTransitionBuffer>(n, K, hex2f(g));
      stepIdx = 0;
      std::cout << "{\"op\":\"new\"}" << std::endl;
    } else if (op == "step") {
      auto r = torch::zeros({K}, torch::kFloat32);
      auto t = torch::zeros({K}, torch::kBool);
      for (int i = 0; i < K; ++i) {
        std::string s;
        is >> s;
        r[i] = hex2f(s);
      }
      for (int i = 0; i < K; ++i) {
        int b;
        is >> b;
        t[i] = (bool)b;
      }
      TensorDict obs = {{"tag", torch::full({K}, (float)stepIdx)}};
      TensorDict act = {{"a", torch::full({K}, (int64_t)stepIdx, torch::kInt64)}};
      buf->pushObsAndAction(obs, act);
      buf->pushRewardAndTerminal(r, t);
      ++stepIdx;
      if (!buf->canPop()) {
        std::cout << "{\"op\":\"step\",\"pop\":false}" << std::endl;
        continue;
      }
      auto tr = buf->popTransition();
      std::cout << "{\"op\":\"step\",\"pop\":true,\"obs_step\":" << (long)tr.obs.at("tag")[0].item<float>()
                << ",\"next_obs_step\":" << (long)tr.nextObs.at("tag")[0].item<float>()
                << ",\"action_step\":" << tr.action.at("a")[0].item<int64_t>() << ",\"reward\":[";
      for (int i = 0; i < K; ++i) std::cout << (i ? "," : "") << "\"" << f2hex(tr.reward[i].item<float>()) << "\"";
      std::cout << "],\"bootstrap\":[";
      for (int i = 0; i < K; ++i) std::cout << (i ? "," : "") << tr.bootstrap[i].item<float>();
      std::cout << "],\"terminal\":[";
      for (int i = 0; i < K; ++i) std::cout << (i ? "," : "") << (int)tr.terminal[i].item<bool>();
      std::cout << "]}" << std::endl;
    } else {
      std::cerr << "bad op: " << op << std::endl;
      return 2;
    }
  }
  return 0;
}
