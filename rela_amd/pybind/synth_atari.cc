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
// this repo's optional in-place rendering extension; absent when this file is compiled against the REFERENCE's
// rela/env.h for the oracle (oracle/Makefile: _ref/synth_atari), where the env is a plain three-virtual rela::Env
#if __has_include("rela/frame_row_env.h")
#include "rela/frame_row_env.h"
#define RELA_HAS_FRAME_ROW 1
#define RELA_FRAME_ROW_BASE , public rela::FrameRowEnv
#else
#define RELA_HAS_FRAME_ROW 0
#define RELA_FRAME_ROW_BASE
#endif

namespace py = pybind11;

namespace {

// n bytes of the LCG stream (x <- 1664525 x + 1013904223, top byte of every state) starting AFTER `state`; returns the
// last state.  The recurrence is a serial dependency chain (~4 cycles per byte: 28,224 bytes = 37-40 us per frame
// stack, which bounded the threaded benchmark at ~430 k env-steps/s on 16 cores), so 32 consecutive states are carried
// in four 8-lane vectors and advanced by the 32-step jump x_{k+32} = A32 x_k + C32: the same bytes in the same order,
// bit for bit (6 us per frame stack).  Vector k, lane j holds x_{i+4j+k+1}: the top bytes of lane j of the four
// vectors are four CONSECUTIVE output bytes, so the packed word is stored as it is.
typedef uint32_t v8u __attribute__((vector_size(32)));
__attribute__((target_clones("avx2", "default"))) uint32_t lcgFill(uint32_t state, uint8_t* out, int n) {
  constexpr uint32_t a = 1664525u, c = 1013904223u;
  v8u l[4];
  uint32_t A32 = 1, C32 = 0;
  for (int p = 0; p < 32; ++p) {
    state = state * a + c;
    l[p & 3][p >> 2] = state;
    C32 = C32 * a + c;
    A32 *= a;
  }
  int i = 0;
  for (; i + 32 <= n; i += 32) {
    const v8u w = (l[0] >> 24) | ((l[1] >> 24) << 8) | ((l[2] >> 24) << 16) | (l[3] & 0xFF000000u);
    std::memcpy(out + i, &w, 32);
    for (int k = 0; k < 4; ++k) l[k] = l[k] * A32 + C32;
  }
  // l[0][0] is x_{i+1}: the serial tail (n not a multiple of 32) and the returned state continue from
  // x_i = (x_{i+1} - c) * a^-1 (mod 2^32)
  constexpr uint32_t aInv = 4276115653u;  // a * aInv == 1 (mod 2^32)
  uint32_t last = (l[0][0] - c) * aInv;
  for (; i < n; ++i) {
    last = last * a + c;
    out[i] = (uint8_t)(last >> 24);
  }
  return last;
}

class SyntheticAtariEnv : public rela::Env RELA_FRAME_ROW_BASE {
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
    const auto& at = action.at("a");
    const int64_t a = (at.device().is_cpu() && at.scalar_type() == torch::kInt64 && at.numel() == 1)
                          ? *at.data_ptr<int64_t>()  // (item() costs a dispatcher round trip per env-step)
                          : at.item<int64_t>();
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

#if RELA_HAS_FRAME_ROW
  // render into the VectorEnv's page-locked row from now on (it already holds the current observation)
  void bindFrameRow(uint8_t* row) final { frame_ = torch::from_blob(row, {4, 84, 84}, torch::kUInt8); }
  bool slidingStack() const final { return sliding_; }
#endif

 private:
  uint32_t next() {
    state_ = state_ * 1664525u + 1013904223u;
    return state_;
  }
  void fillFrame(bool episodeStart) {
    uint8_t* p = frame_.data_ptr<uint8_t>();
    constexpr int kPlane = 84 * 84;
    if (!sliding_) {
      state_ = lcgFill(state_, p, 4 * kPlane);
      return;
    }
    if (!episodeStart) std::memmove(p, p + kPlane, 3 * kPlane);
    state_ = lcgFill(state_, p + 3 * kPlane, kPlane);
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

// Zero-cost env for measuring the ENGINE's own ceiling through rela.Context / BasicThreadLoop / DQNActor (the
// observation is a constant frame stack, the reward 0, episodes end after episode_len steps): whatever rate the
// threaded benchmark reaches with it is what the runtime -- thread loop, VectorEnv, upload, cohort barrier, device
// tick -- can do when the env costs nothing.  The reward is a cheap pseudo-random FLOAT in [-0.5, 0.5): with constant
// frames the Q-values are constant, and constant TD priorities make every float block sum of the replay round the
// same way, so sum_ drifts systematically above the stored weights until a scan runs off the ring -- where the
// reference aborts (prioritized_replay.h:297-302) and this engine raises.
class NullAtariEnv : public rela::Env RELA_FRAME_ROW_BASE {
 public:
  NullAtariEnv(float eps, int numAction, int episodeLen, int seed = 1)
      : numAction_(numAction), episodeLen_(episodeLen), state_((uint32_t)seed * 2654435761u + 12345u) {
    eps_ = torch::full({1}, eps, torch::kFloat32);
    legal_ = torch::ones({numAction}, torch::kFloat32);
    frame_ = torch::full({4, 84, 84}, 17, torch::kUInt8);
  }
  int numAction() const { return numAction_; }
  rela::TensorDict reset() final {
    steps_ = 0;
    terminal_ = false;
    return {{"s", frame_}, {"eps", eps_}, {"legal_move", legal_}};
  }
  std::tuple<rela::TensorDict, float, bool> step(const rela::TensorDict& action) final {
    (void)action;
    if (++steps_ >= episodeLen_) terminal_ = true;
    state_ = state_ * 1664525u + 1013904223u;
    const float reward = (float)(state_ >> 8) * (1.0f / 16777216.0f) - 0.5f;
    return std::make_tuple(rela::TensorDict{{"s", frame_}, {"eps", eps_}, {"legal_move", legal_}}, reward, terminal_);
  }
  bool terminated() const final { return terminal_; }
#if RELA_HAS_FRAME_ROW
  void bindFrameRow(uint8_t* row) final { frame_ = torch::from_blob(row, {4, 84, 84}, torch::kUInt8); }
  bool slidingStack() const final { return true; }  // a constant frame is a stack that slides onto itself
#endif

 private:
  const int numAction_, episodeLen_;
  uint32_t state_;
  int steps_ = 0;
  bool terminal_ = true;
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
  py::class_<NullAtariEnv, rela::Env, std::shared_ptr<NullAtariEnv>>(m, "NullAtariEnv")
      .def(py::init<float, int, int, int>(), py::arg("eps"), py::arg("num_action"), py::arg("episode_len"), py::arg("seed") = 1)
      .def("num_action", &NullAtariEnv::numAction)
      .def("reset", &NullAtariEnv::reset)
      .def("step", &NullAtariEnv::step)
      .def("terminated", &NullAtariEnv::terminated);
}
