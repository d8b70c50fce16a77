// synth_atari.cc -- second native module (`synth_atari`), the counterpart of the reference's
// atari/pybind.cc:12-30: a C++ subclass of rela::Env registered against rela.Env.
//
// ALE and ROMs are not available in this pipeline, so the env is synthetic but
// keeps the observation contract of atari/atari_env.h:83-155: {"s": u8[4,84,84], "eps": f32[1],
// "legal_move": f32[A]}, clipped rewards in {-1,0,1}, fixed episode length.  Frames come from the
// 32-bit LCG of SURVEY 8d (x <- 1664525 x + 1013904223, top byte), so a run is reproducible from
// (seed, actions) alone.
#include <pybind11/pybind11.h>

#include <cstring>
#include <torch/extension.h>

#include "rela/env.h"

namespace py = pybind11;

namespace {

class SyntheticAtariEnv : public rela::Env {
 public:
  // slidingStack: the observation is a stack of four planes of which ONE is new per step and the first plane of
  // an episode is repeated four times -- GameState::computeFeature's stacking (atari/game_state.h:53-82).  The
  // default (false) draws four fresh planes per step, the LCG frames of SURVEY 8d the goldens were recorded with.
  SyntheticAtariEnv(int seed, float eps, int numAction, int episodeLen, bool slidingStack = false)
      : state_((uint32_t)seed), numAction_(numAction), episodeLen_(episodeLen), sliding_(slidingStack), steps_(0),
        terminal_(true), episodeReward_(0.f) {
    eps_ = torch::full({1}, eps, torch::kFloat32);  // shape [1]: SURVEY H6
    legal_ = torch::ones({numAction}, torch::kFloat32);
    frame_ = torch::zeros({4, 84, 84}, torch::kUInt8);
  }

  int numAction() const { return numAction_; }
  float getEpisodeReward() const { return episodeReward_; }

  rela::TensorDict reset() final {
    steps_ = 0;
    terminal_ = false;
    episodeReward_ = 0.f;
    fillFrame(true);
    return observation();
  }

  std::tuple<rela::TensorDict, float, bool> step(const rela::TensorDict& action) final {
    const int64_t a = action.at("a").item<int64_t>();
    if (a < 0 || a >= numAction_) throw std::out_of_range("SyntheticAtariEnv: action out of range");
    fillFrame(false);
    const uint32_t x = next();
    float reward = 0.f;
    if ((a & 1) == 0) reward = (float)((int)((x >> 24) % 3) - 1);
    episodeReward_ += reward;
    ++steps_;
    if (steps_ >= episodeLen_) terminal_ = true;
    return std::make_tuple(observation(), reward, terminal_);
  }

  bool terminated() const final { return terminal_; }

 private:
  uint32_t next() {
    state_ = state_ * 1664525u + 1013904223u;
    return state_;
  }
  void fillFrame(bool episodeStart) {
    uint8_t* p = frame_.data_ptr<uint8_t>();
    constexpr int kPlane = 84 * 84;
    if (!sliding_) {
      for (int i = 0; i < 4 * kPlane; ++i) p[i] = (uint8_t)(next() >> 24);
      return;
    }
    if (!episodeStart) std::memmove(p, p + kPlane, 3 * kPlane);
    for (int i = 0; i < kPlane; ++i) p[3 * kPlane + i] = (uint8_t)(next() >> 24);
    if (episodeStart)
      for (int k = 0; k < 3; ++k) std::memcpy(p + k * kPlane, p + 3 * kPlane, kPlane);
  }
  rela::TensorDict observation() const { return {{"s", frame_}, {"eps", eps_}, {"legal_move", legal_}}; }

  uint32_t state_;
  const int numAction_, episodeLen_;
  const bool sliding_;
  int steps_;
  bool terminal_;
  float episodeReward_;
  torch::Tensor eps_, legal_, frame_;
};

}  // namespace

PYBIND11_MODULE(synth_atari, m) {
  py::module_::import("rela");  // registers the rela.Env base class
  py::class_<SyntheticAtariEnv, rela::Env, std::shared_ptr<SyntheticAtariEnv>>(m, "SyntheticAtariEnv")
      .def(py::init<int, float, int, int, bool>(), py::arg("seed"), py::arg("eps"), py::arg("num_action"),
           py::arg("episode_len"), py::arg("sliding_stack") = false)
      .def("num_action", &SyntheticAtariEnv::numAction)
      .def("reset", &SyntheticAtariEnv::reset)
      .def("step", &SyntheticAtariEnv::step)
      .def("terminated", &SyntheticAtariEnv::terminated)
      .def("get_episode_reward", &SyntheticAtariEnv::getEpisodeReward);
}
