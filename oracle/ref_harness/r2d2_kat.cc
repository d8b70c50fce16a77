// Golden-vector generator for the R2D2 sequence buffer.
//
// TEST INFRASTRUCTURE, build-time-here only (oracle/Makefile, target _ref/r2d2_kat).
// Drives rela::R2D2TransitionBuffer (rela/r2d2_actor.h:10-187) the way R2D2Actor::postStep
// (:272-302) does -- push (:29-87) every step, popTransition (:93-170) whenever canPop -- with
// tagged transitions, and prints every emitted sequence.
//
// script grammar:
//   new <batch> <multi_step> <seq_len> <burn_in>
//   push <t_0..t_{K-1}> <p_0..p_{K-1}>       terminal flags (0/1) and per-step priorities (hex f32)
// The transition of env i at push number s carries obs tag s*100+i, action s, reward s+0.5,
// bootstrap (s%2); the hidden state passed along is s+1 (0 for an env at episode start, which
// the buffer asserts, :42-44).
#include <cstdint>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>

#include <torch/extension.h>
#include <torch/torch.h>

#include "rela/r2d2_actor.h"

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
  std::unique_ptr<R2D2TransitionBuffer> buf;
  int K = 0, n = 0, seqLen = 0, burnin = 0;
  long step = 0;
  std::vector<bool> fresh;
  std::string line;
  while (std::getline(std::cin, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream is(line);
    std::string op;
    is >> op;
    if (op == "new") {
      is >> K >> n >> seqLen >> burnin;
      buf = std::make_unique<R2D2TransitionBuffer>(K, n, seqLen, burnin);
      fresh.assign(K, true);
      step = 0;
      std::cout << "{\"op\":\"new\"}" << std::endl;
    } else if (op == "push") {
      auto term = torch::zeros({K}, torch::kBool);
      auto prio = torch::zeros({K}, torch::kFloat32);
      for (int i = 0; i < K; ++i) {
        int b;
        is >> b;
        term[i] = (bool)b;
      }
      for (int i = 0; i < K; ++i) {
        std::string p;
        is >> p;
        prio[i] = hex2f(p);
      }
      auto tag = torch::arange(K).to(torch::kFloat32) + (float)(step * 100);
      TensorDict obs = {{"tag", tag}};
      TensorDict act = {{"a", torch::full({K}, (int64_t)step, torch::kInt64)}};
      auto reward = torch::full({K}, (float)step + 0.5f);
      auto boot = torch::full({K}, (float)(step % 2));
      FFTransition t(obs, act, reward, term, boot, obs);
      auto h = torch::full({1, K, 2}, (float)(step + 1));
      for (int i = 0; i < K; ++i)
        if (fresh[i]) h.narrow(1, i, 1).zero_();
      TensorDict hid = {{"h0", h}};
      buf->push(t, prio, hid);
      for (int i = 0; i < K; ++i) fresh[i] = term[i].item<bool>();
      ++step;
      if (!buf->canPop()) {
        std::cout << "{\"op\":\"push\",\"pop\":false}" << std::endl;
        continue;
      }
      std::vector<RNNTransition> seqs;
      torch::Tensor sp, lens;
      std::tie(seqs, sp, lens) = buf->popTransition();
      std::cout << "{\"op\":\"push\",\"pop\":true,\"seqs\":[";
      for (size_t q = 0; q < seqs.size(); ++q) {
        const auto& s = seqs[q];
        const int T = (int)s.reward.size(0);
        std::cout << (q ? "," : "") << "{\"len\":" << lens[q].item<float>() << ",\"h0\":" << s.h0.at("h0")[0][0].item<float>()
                  << ",\"tag\":[";
        for (int j = 0; j < T; ++j) std::cout << (j ? "," : "") << (long)s.obs.at("tag")[j].item<float>();
        std::cout << "],\"a\":[";
        for (int j = 0; j < T; ++j) std::cout << (j ? "," : "") << s.action.at("a")[j].item<int64_t>();
        std::cout << "],\"reward\":[";
        for (int j = 0; j < T; ++j) std::cout << (j ? "," : "") << s.reward[j].item<float>();
        std::cout << "],\"terminal\":[";
        for (int j = 0; j < T; ++j) std::cout << (j ? "," : "") << (int)s.terminal[j].item<bool>();
        std::cout << "],\"bootstrap\":[";
        for (int j = 0; j < T; ++j) std::cout << (j ? "," : "") << s.bootstrap[j].item<float>();
        std::cout << "],\"prio\":[";
        for (int j = 0; j < seqLen; ++j) std::cout << (j ? "," : "") << "\"" << f2hex(sp[q][j].item<float>()) << "\"";
        std::cout << "]}";
      }
      std::cout << "]}" << std::endl;
    } else {
      std::cerr << "bad op: " << op << std::endl;
      return 2;
    }
  }
  return 0;
}
