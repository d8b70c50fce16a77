// rela_module.cc -- the pybind11 module `rela`: drop-in for the reference's rela/pybind.cc:19-108.
//
// Same 13 Python names, constructor signatures and method names; underneath, every hot-path
// object is a handle on the C ABI of include/rela_amd.h (librela_amd.so, HIP/gfx950):
//
//   FFPrioritizedReplay  -> rela_replay_*        (device-resident ring, exact scan)
//   DQNActor             -> rela_apex_actor_*    (device-resident history, MFMA forward)
//   ModelLocker          -> rela_ffnet_*         (N versioned device weight sets + in-flight counts)
//   Context / BasicThreadLoop / VectorEnv / Env  -> host C++ threads, as in the reference
//
// Threading and error behaviour follow SURVEY 8b: bound methods run with the GIL held, actor
// threads never take it; protocol violations raise on the Python thread and terminate on actor
// threads.  There is no CPU execution path: actors need a "cuda:N" ModelLocker.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <torch/extension.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>

#include "rela/env.h"
#include "rela/types.h"
#include "rela_amd.h"

namespace py = pybind11;
using namespace rela;

namespace {

[[noreturn]] void fail(const std::string& where, int code) {
  throw std::runtime_error(where + " failed (" + std::to_string(code) + "): " + rela_last_error());
}
inline void check(int code, const char* where) {
  if (code != RELA_OK) fail(where, code);
}

int parseDevice(const std::string& device) {
  if (device == "cpu") return -1;
  if (device == "cuda") return 0;
  if (device.rfind("cuda:", 0) == 0) return std::stoi(device.substr(5));
  throw std::invalid_argument("unsupported device string: " + device);
}

// hipStream_t of torch's current stream on `device`, fetched through Python (GIL held).
void* torchCurrentStream(int device) {
  py::object s = py::module_::import("torch").attr("cuda").attr("current_stream")(device);
  return reinterpret_cast<void*>(s.attr("cuda_stream").cast<uintptr_t>());
}

constexpr int64_t kObsBytes = 4 * 84 * 84;

// RELA_THREADED_STATS=1: where the actor threads' wall time goes (summed over threads, printed when the Context dies):
// the drop-in's throughput is bound by host work per env-step, which no GPU profiler sees.
struct ThreadedStats {
  std::atomic<int64_t> envStep{0}, envReset{0}, actPrep{0}, actWait{0}, actLead{0}, postWait{0}, postLead{0}, setRT{0};
  std::atomic<int64_t> envSteps{0}, ticks{0};
  const bool on = [] {
    const char* e = std::getenv("RELA_THREADED_STATS");
    return e && e[0] == '1';
  }();
  static int64_t now() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  void print() const {
    if (!on || envSteps.load() == 0) return;
    const double n = (double)envSteps.load();
    auto us = [n](const std::atomic<int64_t>& v) { return v.load() * 1e-3 / n; };
    std::fprintf(stderr,
                 "[rela threaded stats] env-steps %lld, cohort ticks %lld; thread-us per env-step: env.step %.2f, env.reset %.2f, "
                 "act: prep+upload %.2f, barrier wait %.2f, leader launch+sync %.2f; setRewardAndTerminal %.2f; "
                 "postStep: barrier wait %.2f, leader launch %.2f\n",
                 (long long)envSteps.load(), (long long)ticks.load(), us(envStep), us(envReset), us(actPrep), us(actWait),
                 us(actLead), us(setRT), us(postWait), us(postLead));
  }
};
ThreadedStats gStats;
struct StatTimer {
  std::atomic<int64_t>& acc;
  const int64_t t0;
  explicit StatTimer(std::atomic<int64_t>& a) : acc(a), t0(gStats.on ? ThreadedStats::now() : 0) {}
  ~StatTimer() {
    if (gStats.on) acc += ThreadedStats::now() - t0;
  }
};

}  // namespace

// =====================================================================================
// VectorEnv (rela/env.h:29-102) -- persistent batch buffers
// =====================================================================================
namespace rela {

void VectorEnv::storeRow(const TensorDict& obs, int row) {
  const int K = (int)envs_.size();
  for (const auto& kv : obs) {
    auto it = batch_.find(kv.first);
    if (it == batch_.end()) {
      std::vector<int64_t> shape{K};
      for (auto d : kv.second.sizes()) shape.push_back(d);
      auto t = torch::zeros(shape, kv.second.options().device(torch::kCPU));
      if (torch::cuda::is_available()) t = t.pin_memory();
      it = batch_.emplace(kv.first, t).first;
    }
    auto dst = it->second[row];
    const auto& src = kv.second;
    if (src.is_contiguous() && src.device().is_cpu() && src.dtype() == dst.dtype() && src.numel() == dst.numel()) {
      std::memcpy(dst.data_ptr(), src.data_ptr(), (size_t)src.nbytes());
    } else {
      dst.copy_(src);
    }
  }
}

TensorDict VectorEnv::reset(const TensorDict& previous) {
  const bool first = previous.empty();
  for (size_t i = 0; i < envs_.size(); ++i) {
    if (first || envs_[i]->terminated()) storeRow(envs_[i]->reset(), (int)i);
  }
  return batch_;
}

std::tuple<TensorDict, torch::Tensor, torch::Tensor> VectorEnv::step(const TensorDict& action) {
  const int K = (int)envs_.size();
  if (!reward_.defined()) {
    reward_ = torch::zeros({K}, torch::kFloat32);
    terminal_ = torch::zeros({K}, torch::kBool);
  }
  float* r = reward_.data_ptr<float>();
  bool* t = terminal_.data_ptr<bool>();
  for (int i = 0; i < K; ++i) {
    TensorDict a;
    for (const auto& kv : action) a.emplace(kv.first, kv.second[i]);
    TensorDict obs;
    float reward;
    bool terminal;
    std::tie(obs, reward, terminal) = envs_[i]->step(a);
    storeRow(obs, i);
    r[i] = reward;
    t[i] = terminal;
  }
  return std::make_tuple(batch_, reward_, terminal_);
}

bool VectorEnv::anyTerminated() const {
  for (const auto& e : envs_)
    if (e->terminated()) return true;
  return false;
}

bool VectorEnv::allTerminated() const {
  for (const auto& e : envs_)
    if (!e->terminated()) return false;
  return true;
}

// =====================================================================================
// Actor interface (rela/actor.h:9-21)
// =====================================================================================
class Actor {
 public:
  virtual ~Actor() = default;
  virtual TensorDict act(TensorDict& obs) = 0;
  virtual void setRewardAndTerminal(torch::Tensor& r, torch::Tensor& t) = 0;
  virtual void postStep() = 0;
  // runtime hook, not part of the reference interface: the thread that drove this actor has left
  // its main loop (lets batched cohorts release the remaining members)
  virtual void onLoopExit() {}
  // runtime hook: the owning Context is being destroyed -- unpark anything that could block a join
  virtual void onShutdown() {}
};

// =====================================================================================
// ModelLocker (rela/model_locker.h:11-65): N weight versions per act device, each an
// (online, target) pair of rela_ffnet objects, with in-flight counts.  update_model waits until
// the next slot is idle, refills it from pyModel.state_dict() and publishes it.
// =====================================================================================
class ModelLocker {
 public:
  enum Kind { kFF = 0, kLSTM = 1 };
  struct Lease {
    int id;
    Kind kind;
    const void* online;  // rela_ffnet* or rela_lstmnet*
    const void* target;
  };

  ModelLocker(std::vector<py::object> pyModels, const std::string& device)
      : device(device), deviceIndex(parseDevice(device)), pyModels_(std::move(pyModels)) {
    if (pyModels_.empty()) throw std::invalid_argument("ModelLocker needs at least one model");
    const size_t n = pyModels_.size();
    online_.assign(n, nullptr);
    target_.assign(n, nullptr);
    inFlight_.assign(n, 0);
    if (py::hasattr(pyModels_[0], "eta")) eta_ = pyModels_[0].attr("eta").cast<double>();
    if (deviceIndex >= 0) loadSlot(0, pyModels_[0]);
  }

  ~ModelLocker() {
    for (auto* p : online_) destroyNet(p);
    for (auto* p : target_) destroyNet(p);
  }

  void updateModel(py::object pyModel) {
    int id;
    {
      std::unique_lock<std::mutex> lk(m_);
      id = (latest_ + 1) % (int)inFlight_.size();
      cv_.wait(lk, [&] { return inFlight_[id] == 0; });  // model_locker.h:27-28
    }
    if (deviceIndex >= 0) {
      loadSlot(id, pyModel);
    } else {
      pyModels_[id].attr("load_state_dict")(pyModel.attr("state_dict")());  // cpu locker: bookkeeping only
    }
    std::lock_guard<std::mutex> lk(m_);
    latest_ = id;
  }

  Lease getModel() {
    if (deviceIndex < 0)
      throw std::runtime_error("ModelLocker('cpu'): this engine has no CPU actor path; use a cuda device");
    std::lock_guard<std::mutex> lk(m_);
    ++inFlight_[latest_];
    return Lease{latest_, kind_, online_[latest_], target_[latest_]};
  }

  void releaseModel(int id) {
    std::lock_guard<std::mutex> lk(m_);
    if (--inFlight_[id] == 0) cv_.notify_all();
  }

  int numAction() const { return numAction_; }
  Kind kind() const { return kind_; }
  double eta() const { return eta_; }

  const std::string device;
  const int deviceIndex;

 private:
  void destroyNet(void* p) {
    if (!p) return;
    if (kind_ == kFF) rela_ffnet_destroy(static_cast<rela_ffnet*>(p));
    else rela_lstmnet_destroy(static_cast<rela_lstmnet*>(p));
  }

  std::vector<torch::Tensor> fetch(py::dict& sd, const std::string& prefix, const std::vector<const char*>& keys) {
    std::vector<torch::Tensor> out;
    for (const char* k : keys) {
      const std::string key = prefix + k;
      if (!sd.contains(py::str(key))) throw std::runtime_error("ModelLocker: state_dict has no '" + key + "'");
      auto t = sd[py::str(key)].cast<torch::Tensor>().detach();
      out.push_back(t.to(torch::Device(torch::kCUDA, (c10::DeviceIndex)deviceIndex), torch::kFloat32).contiguous());
    }
    return out;
  }

  // RELA_PRECISION=bf16x2: the actors' conv trunks on split-bf16 MFMA (Q within 2e-6 of the default exact f32 mode,
  // DESIGN 4.3b); anything else keeps the parity mode
  static bool fastPrecision() {
    const char* e = std::getenv("RELA_PRECISION");
    return e && std::string(e) == "bf16x2";
  }

  void loadNet(void*& net, py::dict& sd, const std::string& prefix) {
    void* stream = torchCurrentStream(deviceIndex);
    std::vector<torch::Tensor> t;
    if (kind_ == kFF) {
      t = fetch(sd, prefix, {"net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias",
                             "linear.0.weight", "linear.0.bias", "fc_v.weight", "fc_v.bias", "fc_a.weight", "fc_a.bias"});
      const int A = (int)t[10].size(0);
      checkActions(A);
      auto* n = static_cast<rela_ffnet*>(net);
      if (!n) {
        check(rela_ffnet_create(&n, A, deviceIndex), "rela_ffnet_create");
        check(rela_ffnet_set_precision(n, fastPrecision() ? 1 : 0), "rela_ffnet_set_precision");
      }
      net = n;
      auto f = [&](int i) { return t[i].data_ptr<float>(); };
      rela_ffnet_params p{f(0), f(1), f(2), f(3), f(4), f(5), f(6), f(7), f(8), f(9), f(10), f(11)};
      check(rela_ffnet_load(n, &p, 1, stream), "rela_ffnet_load");
    } else {
      t = fetch(sd, prefix, {"net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias",
                             "lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0",
                             "fc_v.weight", "fc_v.bias", "fc_a.weight", "fc_a.bias"});
      const int A = (int)t[12].size(0);
      checkActions(A);
      auto* n = static_cast<rela_lstmnet*>(net);
      if (!n) {
        check(rela_lstmnet_create(&n, A, deviceIndex), "rela_lstmnet_create");
        check(rela_lstmnet_set_precision(n, fastPrecision() ? 1 : 0), "rela_lstmnet_set_precision");
      }
      net = n;
      auto f = [&](int i) { return t[i].data_ptr<float>(); };
      rela_lstmnet_params p{f(0), f(1), f(2), f(3), f(4), f(5), f(6), f(7), f(8), f(9), f(10), f(11), f(12), f(13)};
      check(rela_lstmnet_load(n, &p, 1, stream), "rela_lstmnet_load");
    }
    // the packing kernels read `t` on torch's stream: finish them before the tensors die
    py::module_::import("torch").attr("cuda").attr("current_stream")(deviceIndex).attr("synchronize")();
  }

  void checkActions(int A) {
    if (numAction_ == 0) numAction_ = A;
    if (A != numAction_) throw std::runtime_error("ModelLocker: num_action changed between updates");
  }

  void loadSlot(int id, py::object& pyModel) {
    // Leases are released when an actor has QUEUED its kernels; drain the device so nothing that
    // still reads this slot's old weights is in flight (the reference's model call is synchronous).
    py::module_::import("torch").attr("cuda").attr("synchronize")(deviceIndex);
    py::dict sd = pyModel.attr("state_dict")();
    if (!kindKnown_) {
      kind_ = sd.contains(py::str("online_net.lstm.weight_ih_l0")) ? kLSTM : kFF;
      kindKnown_ = true;
    }
    loadNet(online_[id], sd, "online_net.");
    loadNet(target_[id], sd, "target_net.");
  }

  std::vector<py::object> pyModels_;
  std::vector<void*> online_, target_;
  std::vector<int> inFlight_;
  int latest_ = 0;
  int numAction_ = 0;
  Kind kind_ = kFF;
  bool kindKnown_ = false;
  double eta_ = 0.9;
  std::mutex m_;
  std::condition_variable cv_;
};

// Not in the reference's surface: what a replay PARTITION contributes to the importance weights of a batch
// drawn over several partitions -- the un-normalised weights w_i of the last sample, the float
// sum they were drawn against and the size its weights used (prioritized_replay.h:289,261,312).
static std::tuple<torch::Tensor, torch::Tensor, int> lastSampleRaw(rela_replay* h, int device, int n) {
  if (!h || n <= 0) throw std::runtime_error("last_sample_raw: nothing was sampled");
  const float* raw = nullptr;
  const float* sum = nullptr;
  check(rela_replay_last_sample_dev(h, &raw, &sum), "rela_replay_last_sample_dev");
  auto opt = torch::TensorOptions().dtype(torch::kFloat32).device(torch::Device(torch::kCUDA, (c10::DeviceIndex)device));
  // the replay's stream wrote them; sample() already made torch's current stream wait for that work
  auto w = torch::from_blob(const_cast<float*>(raw), {n}, opt).clone();
  auto s = torch::from_blob(const_cast<float*>(sum), {1}, opt).clone();
  return std::make_tuple(w, s, rela_replay_last_sample_size(h));
}

// =====================================================================================
// FFPrioritizedReplay (rela/prioritized_replay.h:173-348 as bound in pybind.cc:37-47)
// =====================================================================================
class FFPrioritizedReplay {
 public:
  FFPrioritizedReplay(int capacity, int seed, float alpha, float beta, int prefetch)
      : capacity_(capacity), seed_(seed), alpha_(alpha), beta_(beta), prefetch_(prefetch) {}

  ~FFPrioritizedReplay() { rela_replay_destroy(h_); }

  // created lazily by the first actor that knows the device and the action count
  rela_replay* handle(int device, int numAction) {
    std::lock_guard<std::mutex> lk(m_);
    if (!h_) {
      check(rela_replay_create(&h_, capacity_, seed_, alpha_, beta_, prefetch_, device), "rela_replay_create");
      const int64_t A = numAction;
      const int64_t rb[10] = {kObsBytes, kObsBytes, 4, 4, 4 * A, 4 * A, 8, 4, 1, 4};
      // RELA_REPLAY_DEDUP=stack|plane: frame-stack de-duplication (include/rela_amd.h).  "stack" is
      // valid for any env (what the reference's shared tensor views do); "plane" needs an env that slides its
      // stack by one frame per step as atari/game_state.h:53-82 does.
      const char* dd = std::getenv("RELA_REPLAY_DEDUP");
      const int ups = dd && std::string(dd) == "stack" ? 1 : (dd && std::string(dd) == "plane" ? 4 : 0);
      if (ups > 0) {
        const char* gd = std::getenv("RELA_REPLAY_DEDUP_GUARD");
        const int64_t guard = gd ? std::atoll(gd) : 131072;  // units stored ahead of their transitions
        check(rela_replay_set_schema_dedup(h_, 10, rb, 0, 1, kObsBytes / ups, ups, guard), "rela_replay_set_schema_dedup");
      } else {
        check(rela_replay_set_schema(h_, 10, rb), "rela_replay_set_schema");
      }
      device_ = device;
      numAction_ = numAction;
    } else if (device != device_ || numAction != numAction_) {
      throw std::runtime_error(
          "FFPrioritizedReplay: one replay partition lives on one GPU; actors on another device need their own "
          "partition (SURVEY 8e)");
    }
    return h_;
  }

  int size() const { return h_ ? rela_replay_size(h_) : 0; }
  int numAdd() const { return h_ ? (int)rela_replay_num_add(h_) : 0; }
  void shutdown() {
    std::lock_guard<std::mutex> lk(m_);
    if (h_) rela_replay_shutdown(h_);
  }

  // prefetch > 0 (rela/prioritized_replay.h:223-230: sampler futures that run next to the learner): here a DEVICE-side
  // prefetch -- update_priority k queues sample k + 1 right behind itself on the replay's stream, so its latency-bound
  // kernels run while the Python learner loop gets from update_priority back to sample; sample() then hands that
  // batch over.  One outstanding batch at a time as in the library's protocol, and the same sequence of library calls
  // as without prefetch: ids, weights and rows are bit-identical to prefetch = 0 (tests/test_e2e_gpu.py).
  std::tuple<FFTransition, torch::Tensor> sample(int batchsize, const std::string& device) {
    if (prefetched_ && prefetchedBatch_ == batchsize && prefetchedDevice_ == device) {
      auto r = std::move(*prefetched_);
      prefetched_.reset();
      return r;
    }
    if (prefetched_) throw std::runtime_error("FFPrioritizedReplay.sample: batch size / device changed under prefetch");
    return sampleNow(batchsize, device);
  }

  std::tuple<FFTransition, torch::Tensor> sampleNow(int batchsize, const std::string& device) {
    if (!h_) throw std::runtime_error("FFPrioritizedReplay.sample: the replay is empty");
    const auto dev = torch::Device(torch::kCUDA, (c10::DeviceIndex)device_);
    auto u8 = torch::TensorOptions().dtype(torch::kUInt8).device(dev);
    auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
    const int64_t B = batchsize, A = numAction_;
    FFTransition b;
    b.obs["s"] = torch::empty({B, 4, 84, 84}, u8);
    b.nextObs["s"] = torch::empty({B, 4, 84, 84}, u8);
    b.obs["eps"] = torch::empty({B, 1}, f32);
    b.nextObs["eps"] = torch::empty({B, 1}, f32);
    b.obs["legal_move"] = torch::empty({B, A}, f32);
    b.nextObs["legal_move"] = torch::empty({B, A}, f32);
    b.action["a"] = torch::empty({B}, torch::TensorOptions().dtype(torch::kInt64).device(dev));
    b.reward = torch::empty({B}, f32);
    b.terminal = torch::empty({B}, torch::TensorOptions().dtype(torch::kBool).device(dev));
    b.bootstrap = torch::empty({B}, f32);
    auto weight = torch::empty({B}, f32);
    void* rows[10] = {b.obs["s"].data_ptr(),          b.nextObs["s"].data_ptr(),          b.obs["eps"].data_ptr(),
                      b.nextObs["eps"].data_ptr(),    b.obs["legal_move"].data_ptr(),     b.nextObs["legal_move"].data_ptr(),
                      b.action["a"].data_ptr(),       b.reward.data_ptr(),                b.terminal.data_ptr(),
                      b.bootstrap.data_ptr()};
    check(rela_replay_sample(h_, batchsize, rows, weight.data_ptr<float>(), torchCurrentStream(device_)),
          "FFPrioritizedReplay.sample");
    lastBatch_ = batchsize;
    lastDevice_ = device;
    const int want = parseDevice(device);
    if (want != device_) {  // learner on another GPU (or the cpu): move the batch, types.cc:34-43
      const auto target = want < 0 ? torch::Device(torch::kCPU) : torch::Device(torch::kCUDA, (c10::DeviceIndex)want);
      auto mv = [&](torch::Tensor& t) { t = t.to(target); };
      for (auto* d : {&b.obs, &b.action, &b.nextObs})
        for (auto& kv : *d) mv(kv.second);
      mv(b.reward);
      mv(b.terminal);
      mv(b.bootstrap);
      mv(weight);
    }
    return std::make_tuple(std::move(b), weight);
  }

  std::tuple<torch::Tensor, torch::Tensor, int> lastSampleRaw_() { return lastSampleRaw(h_, device_, lastBatch_); }

  void updatePriority(const torch::Tensor& priority) {
    if (!h_) throw std::runtime_error("FFPrioritizedReplay.update_priority: nothing was sampled");
    if (priority.dim() != 1) throw std::invalid_argument("update_priority expects a 1-D tensor");  // :236
    auto p = priority.detach().to(torch::kFloat32).contiguous();
    if (p.is_cuda()) {
      if (p.device().index() != device_) p = p.to(torch::Device(torch::kCUDA, (c10::DeviceIndex)device_));
      check(rela_replay_update_priority(h_, (int)p.numel(), p.data_ptr<float>(), 1, torchCurrentStream(device_)),
            "FFPrioritizedReplay.update_priority");
      keep_ = p;  // consumed asynchronously on the replay's stream
    } else {
      check(rela_replay_update_priority(h_, (int)p.numel(), p.data_ptr<float>(), 0, nullptr),
            "FFPrioritizedReplay.update_priority");
    }
    if (prefetch_ > 0 && lastBatch_ > 0 && rela_replay_size(h_) >= lastBatch_) {
      prefetched_.emplace(sampleNow(lastBatch_, lastDevice_));
      prefetchedBatch_ = lastBatch_;
      prefetchedDevice_ = lastDevice_;
    }
  }

 private:
  const int capacity_, seed_;
  const float alpha_, beta_;
  const int prefetch_;
  std::mutex m_;
  rela_replay* h_ = nullptr;
  int device_ = -1, numAction_ = 0;
  int lastBatch_ = 0;
  std::string lastDevice_;
  torch::Tensor keep_;
  std::optional<std::tuple<FFTransition, torch::Tensor>> prefetched_;
  int prefetchedBatch_ = 0;
  std::string prefetchedDevice_;
};

// =====================================================================================
// RNNPrioritizedReplay (PrioritizedReplay<RNNTransition>, pybind.cc:49-59): one slot = one sequence
// of T = burn_in + seq_len + multi_step steps; batches come back time-major (types.cc:140-182).
// =====================================================================================
class RNNPrioritizedReplay {
 public:
  RNNPrioritizedReplay(int capacity, int seed, float alpha, float beta, int prefetch)
      : capacity_(capacity), seed_(seed), alpha_(alpha), beta_(beta), prefetch_(prefetch) {}
  ~RNNPrioritizedReplay() { rela_replay_destroy(h_); }

  rela_replay* handle(int device, int numAction, int T) {
    std::lock_guard<std::mutex> lk(m_);
    if (!h_) {
      check(rela_replay_create(&h_, capacity_, seed_, alpha_, beta_, prefetch_, device), "rela_replay_create");
      const int64_t A = numAction, t = T;
      const int64_t rb[10] = {t * kObsBytes, t * 4, t * 4 * A, t * 8, t * 4, t, t * 4, 2048, 2048, 4};
      const int32_t st[10] = {T, T, T, T, T, T, T, 1, 1, 1};
      check(rela_replay_set_schema_seq(h_, 10, rb, st), "rela_replay_set_schema_seq");
      device_ = device;
      numAction_ = numAction;
      T_ = T;
    } else if (device != device_ || numAction != numAction_ || T != T_) {
      throw std::runtime_error("RNNPrioritizedReplay: actors disagree on device / action count / window length");
    }
    return h_;
  }

  int size() const { return h_ ? rela_replay_size(h_) : 0; }
  int numAdd() const { return h_ ? (int)rela_replay_num_add(h_) : 0; }
  void shutdown() {
    std::lock_guard<std::mutex> lk(m_);
    if (h_) rela_replay_shutdown(h_);
  }

  // (prefetch: as FFPrioritizedReplay::sample)
  std::tuple<RNNTransition, torch::Tensor> sample(int batchsize, const std::string& device) {
    if (prefetched_ && prefetchedBatch_ == batchsize && prefetchedDevice_ == device) {
      auto r = std::move(*prefetched_);
      prefetched_.reset();
      return r;
    }
    if (prefetched_) throw std::runtime_error("RNNPrioritizedReplay.sample: batch size / device changed under prefetch");
    return sampleNow(batchsize, device);
  }

  std::tuple<RNNTransition, torch::Tensor> sampleNow(int batchsize, const std::string& device) {
    if (!h_) throw std::runtime_error("RNNPrioritizedReplay.sample: the replay is empty");
    const auto dev = torch::Device(torch::kCUDA, (c10::DeviceIndex)device_);
    auto opt = [&](torch::ScalarType t) { return torch::TensorOptions().dtype(t).device(dev); };
    const int64_t B = batchsize, A = numAction_, T = T_;
    RNNTransition b;
    b.obs["s"] = torch::empty({T, B, 4, 84, 84}, opt(torch::kUInt8));
    b.obs["eps"] = torch::empty({T, B, 1}, opt(torch::kFloat32));
    b.obs["legal_move"] = torch::empty({T, B, A}, opt(torch::kFloat32));
    b.action["a"] = torch::empty({T, B}, opt(torch::kInt64));
    b.reward = torch::empty({T, B}, opt(torch::kFloat32));
    b.terminal = torch::empty({T, B}, opt(torch::kBool));
    b.bootstrap = torch::empty({T, B}, opt(torch::kFloat32));
    b.h0["h0"] = torch::empty({1, B, 512}, opt(torch::kFloat32));
    b.h0["c0"] = torch::empty({1, B, 512}, opt(torch::kFloat32));
    b.seqLen = torch::empty({B}, opt(torch::kFloat32));
    auto weight = torch::empty({B}, opt(torch::kFloat32));
    void* rows[10] = {b.obs["s"].data_ptr(), b.obs["eps"].data_ptr(), b.obs["legal_move"].data_ptr(),
                      b.action["a"].data_ptr(), b.reward.data_ptr(), b.terminal.data_ptr(), b.bootstrap.data_ptr(),
                      b.h0["h0"].data_ptr(), b.h0["c0"].data_ptr(), b.seqLen.data_ptr()};
    check(rela_replay_sample(h_, batchsize, rows, weight.data_ptr<float>(), torchCurrentStream(device_)),
          "RNNPrioritizedReplay.sample");
    lastBatch_ = batchsize;
    lastDevice_ = device;
    const int want = parseDevice(device);
    if (want != device_) {
      const auto target = want < 0 ? torch::Device(torch::kCPU) : torch::Device(torch::kCUDA, (c10::DeviceIndex)want);
      auto mv = [&](torch::Tensor& t) { t = t.to(target); };
      for (auto* d : {&b.obs, &b.action, &b.h0})
        for (auto& kv : *d) mv(kv.second);
      mv(b.reward);
      mv(b.terminal);
      mv(b.bootstrap);
      mv(b.seqLen);
      mv(weight);
    }
    return std::make_tuple(std::move(b), weight);
  }

  void updatePriority(const torch::Tensor& priority) {
    if (!h_) throw std::runtime_error("RNNPrioritizedReplay.update_priority: nothing was sampled");
    if (priority.dim() != 1) throw std::invalid_argument("update_priority expects a 1-D tensor");
    auto p = priority.detach().to(torch::kFloat32).contiguous();
    if (p.is_cuda()) {
      if (p.device().index() != device_) p = p.to(torch::Device(torch::kCUDA, (c10::DeviceIndex)device_));
      check(rela_replay_update_priority(h_, (int)p.numel(), p.data_ptr<float>(), 1, torchCurrentStream(device_)),
            "RNNPrioritizedReplay.update_priority");
      keep_ = p;
    } else {
      check(rela_replay_update_priority(h_, (int)p.numel(), p.data_ptr<float>(), 0, nullptr),
            "RNNPrioritizedReplay.update_priority");
    }
    if (prefetch_ > 0 && lastBatch_ > 0 && rela_replay_size(h_) >= lastBatch_) {
      prefetched_.emplace(sampleNow(lastBatch_, lastDevice_));
      prefetchedBatch_ = lastBatch_;
      prefetchedDevice_ = lastDevice_;
    }
  }

 private:
  const int capacity_, seed_;
  const float alpha_, beta_;
  const int prefetch_;
  std::mutex m_;
  rela_replay* h_ = nullptr;
  int device_ = -1, numAction_ = 0, T_ = 0;
  int lastBatch_ = 0;
  std::string lastDevice_;
  torch::Tensor keep_;
  std::optional<std::tuple<RNNTransition, torch::Tensor>> prefetched_;
  int prefetchedBatch_ = 0;
  std::string prefetchedDevice_;

 public:
  std::tuple<torch::Tensor, torch::Tensor, int> lastSampleRaw_() { return lastSampleRaw(h_, device_, lastBatch_); }
};

// the actor side of a de-duplicating replay (RELA_REPLAY_DEDUP, see FFPrioritizedReplay::handle)
static void enableDedup(rela_apex_actor* a, rela_replay* rep) {
  if (!rep) return;
  int ups = 0;
  check(rela_replay_dedup_info(rep, &ups, nullptr, nullptr), "rela_replay_dedup_info");
  if (ups > 0) check(rela_apex_actor_set_dedup(a, ups), "rela_apex_actor_set_dedup");
}

// =====================================================================================
// ActorCohort -- cross-thread inference batching ("inference batches cross-thread per GPU").
//
// The T training DQNActors of one Context that share (ModelLocker, replay, K, n, gamma) are backed
// by ONE device shard of T*K rows (rela_apex_actor with group_rows = K).  Every thread still runs
// the reference loop (thread_loop.h:74-105) on its own K envs; act() / postStep() rendezvous at a
// barrier and the last arriver launches the batched work for all rows.  Scope-sensitive arithmetic
// is unchanged: q.min() is taken per group of K rows (one reference TorchScript call) and every
// member's transitions are committed as their own K-slot block (in member order), so the replay
// sees exactly what T independent actors would have appended -- only in a fixed order.
// =====================================================================================
class ActorCohort {
 public:
  ActorCohort(std::shared_ptr<ModelLocker> locker, std::shared_ptr<FFPrioritizedReplay> replay, int multiStep, int K,
              float gamma, int members)
      : locker_(std::move(locker)), replay_(std::move(replay)), n_(multiStep), K_(K), gamma_(gamma), T_(members),
        numAct_(members) {
    for (auto& c : numAct_) c.store(0);
  }

  // R2D2 flavour: the members are R2D2Actors, the shard is a rela_r2d2_actor (r2d2_actor.h:189-353)
  ActorCohort(std::shared_ptr<ModelLocker> locker, std::shared_ptr<RNNPrioritizedReplay> replay, int multiStep, int K,
              float gamma, int seqLen, int burnin, int members)
      : locker_(std::move(locker)), rnnReplay_(std::move(replay)), lstm_(true), seqLen_(seqLen), burnin_(burnin),
        n_(multiStep), K_(K), gamma_(gamma), T_(members), numAct_(members) {
    for (auto& c : numAct_) c.store(0);
  }

  ~ActorCohort() {
    rela_apex_actor_destroy(h_);
    rela_r2d2_actor_destroy(hr_);
    const int dev = locker_->deviceIndex;
    if (compute_) rela_stream_destroy(compute_, dev);
    if (upload_) rela_stream_destroy(upload_, dev);
  }

  int numAct(int member) const { return (int)numAct_[member].load(); }

  TensorDict act(int member, TensorDict& obs) {
    const auto& s = obs.at("s");
    const auto& legal = obs.at("legal_move");
    const auto& eps = obs.at("eps");
    if (s.size(0) != K_ || s.numel() != (int64_t)K_ * kObsBytes || s.dtype() != torch::kUInt8)
      throw std::runtime_error("DQNActor.act: obs['s'] must be uint8 [batchsize,4,84,84]");
    const int A = (int)legal.size(1);
    const int64_t tPrep = gStats.on ? ThreadedStats::now() : 0;
    std::unique_lock<std::mutex> lk(m_);
    if (draining_) return drained();
    if (!created_) create(A);
    // this member's rows: frames go straight to the HBM history slot on the upload stream,
    // the per-env constants to the host staging (uploaded by the leader when they changed)
    auto sc = s.contiguous();
    void* slot = lstm_ ? rela_r2d2_actor_obs_slot(hr_) : rela_apex_actor_obs_slot(h_);
    check(rela_memcpy_h2d_async(static_cast<uint8_t*>(slot) + (int64_t)member * K_ * kObsBytes,
                                sc.data_ptr(), (int64_t)K_ * kObsBytes, upload_, locker_->deviceIndex),
          "rela_memcpy_h2d_async");
    keepObs_[member] = sc;
    auto e = eps.reshape({K_}).to(torch::kFloat32).contiguous();
    auto l = legal.to(torch::kFloat32).contiguous();
    float* ed = epsAll_.data_ptr<float>() + (int64_t)member * K_;
    float* ld = legalAll_.data_ptr<float>() + (int64_t)member * K_ * A;
    if (!constsValid_ || std::memcmp(ed, e.data_ptr(), e.nbytes()) != 0 || std::memcmp(ld, l.data_ptr(), l.nbytes()) != 0) {
      std::memcpy(ed, e.data_ptr(), e.nbytes());
      std::memcpy(ld, l.data_ptr(), l.nbytes());
      constsDirty_ = true;
    }
    if (gStats.on) gStats.actPrep += ThreadedStats::now() - tPrep;
    rendezvous(lk, [&] {
      check(rela_stream_wait_stream(compute_, upload_, locker_->deviceIndex), "rela_stream_wait_stream");
      auto lease = locker_->getModel();
      const float* e = constsDirty_ ? epsAll_.data_ptr<float>() : nullptr;
      const float* l = constsDirty_ ? legalAll_.data_ptr<float>() : nullptr;
      const int rc = lstm_ ? rela_r2d2_actor_act(hr_, static_cast<const rela_lstmnet*>(lease.online), nullptr, e, l,
                                                 actionAll_.data_ptr<int64_t>(), nullptr, compute_)
                           : rela_apex_actor_act(h_, static_cast<const rela_ffnet*>(lease.online), nullptr, e, l,
                                                 actionAll_.data_ptr<int64_t>(), nullptr, compute_);
      locker_->releaseModel(lease.id);
      check(rc, lstm_ ? "R2D2Actor.act (batched)" : "DQNActor.act (batched)");
      constsDirty_ = false;
      constsValid_ = true;
      gStats.ticks += 1;
    }, gStats.actWait, gStats.actLead);
    if (draining_) return drained();
    numAct_[member] += K_;
    return TensorDict{{"a", actionAll_.narrow(0, (int64_t)member * K_, K_)}};
  }

  void setRewardAndTerminal(int member, torch::Tensor& r, torch::Tensor& t) {
    auto rf = r.to(torch::kFloat32).contiguous();
    auto tb = t.to(torch::kBool).contiguous();
    std::lock_guard<std::mutex> lk(m_);
    if (!created_ || draining_) return;
    std::memcpy(rewardAll_.data_ptr<float>() + (int64_t)member * K_, rf.data_ptr(), (size_t)K_ * sizeof(float));
    std::memcpy(terminalAll_.data_ptr<bool>() + (int64_t)member * K_, tb.data_ptr(), (size_t)K_);
  }

  void postStep(int member) {
    (void)member;
    std::unique_lock<std::mutex> lk(m_);
    if (!created_ || draining_) return;
    rendezvous(lk, [&] {
      auto lease = locker_->getModel();
      const uint8_t* term = reinterpret_cast<const uint8_t*>(terminalAll_.data_ptr<bool>());
      const int rc =
          lstm_ ? rela_r2d2_actor_post_step(hr_, rewardAll_.data_ptr<float>(), term,
                                            static_cast<const rela_lstmnet*>(lease.online),
                                            static_cast<const rela_lstmnet*>(lease.target), 0, nullptr, compute_)
                : rela_apex_actor_post_step(h_, rewardAll_.data_ptr<float>(), term, 0,
                                            static_cast<const rela_ffnet*>(lease.online),
                                            static_cast<const rela_ffnet*>(lease.target), 0, nullptr, compute_);
      locker_->releaseModel(lease.id);
      if (rc != RELA_EWOULDBLOCK) check(rc, "postStep (batched)");  // dropped block after shutdown
      // The next round's frames land in the history slot this tick just read (the ring reuses
      // slot `head`): uploads must start after the tick's queued kernels and row copies.
      check(rela_stream_wait_stream(upload_, compute_, locker_->deviceIndex), "rela_stream_wait_stream");
    }, gStats.postWait, gStats.postLead);
  }

  void shutdown() {
    if (lstm_)
      rnnReplay_->shutdown();
    else
      replay_->shutdown();
    std::lock_guard<std::mutex> lk(m_);
    draining_ = true;
    cv_.notify_all();
  }

  void leave(int member) {
    (void)member;
    std::lock_guard<std::mutex> lk(m_);
    draining_ = true;  // one member gone: the cohort cannot complete another round
    cv_.notify_all();
  }

 private:
  template <class F>
  void rendezvous(std::unique_lock<std::mutex>& lk, F&& leaderWork, std::atomic<int64_t>& waitAcc,
                  std::atomic<int64_t>& leadAcc) {
    const uint64_t gen = generation_;
    if (++arrived_ == T_) {
      StatTimer st(leadAcc);
      try {
        leaderWork();
      } catch (...) {
        draining_ = true;
        arrived_ = 0;
        ++generation_;
        cv_.notify_all();
        throw;
      }
      arrived_ = 0;
      ++generation_;
      cv_.notify_all();
    } else {
      StatTimer st(waitAcc);
      cv_.wait(lk, [&] { return generation_ != gen || draining_; });
    }
  }

  TensorDict drained() { return TensorDict{{"a", torch::zeros({K_}, torch::kInt64)}}; }

  void create(int A) {
    if (locker_->kind() != (lstm_ ? ModelLocker::kLSTM : ModelLocker::kFF))
      throw std::runtime_error(lstm_ ? "R2D2Actor needs an AtariLSTMNet-shaped agent in its ModelLocker"
                                     : "DQNActor needs an AtariFFNet-shaped agent in its ModelLocker");
    const int dev = locker_->deviceIndex;
    static std::atomic<uint64_t> counter{0};
    check(rela_stream_create(&compute_, dev), "rela_stream_create");
    check(rela_stream_create(&upload_, dev), "rela_stream_create");
    if (lstm_) {
      // one pop of the shard commits the sequences of all members as ONE block, in row (= member)
      // order; the reference would issue one block per thread (only the float block-sum grouping of
      // sum_ differs, far below the fp tolerance of the priorities themselves)
      rela_replay* rep = rnnReplay_->handle(dev, A, burnin_ + seqLen_ + n_);
      check(rela_r2d2_actor_create(&hr_, T_ * K_, K_, A, n_, gamma_, seqLen_, burnin_, locker_->eta(), rep,
                                   0xC2B2AE3D27D4EB4Full * (++counter), dev),
            "rela_r2d2_actor_create");
    } else {
      rela_replay* rep = replay_->handle(dev, A);
      check(rela_apex_actor_create(&h_, T_ * K_, K_, A, n_, gamma_, rep, 0xA24BAED4963EE407ull * (++counter), dev),
            "rela_apex_actor_create");
      enableDedup(h_, rep);
    }
    created_ = true;
    auto pin = [](torch::Tensor t) { return torch::cuda::is_available() ? t.pin_memory() : t; };
    const int64_t R = (int64_t)T_ * K_;
    actionAll_ = pin(torch::zeros({R}, torch::kInt64));
    epsAll_ = pin(torch::zeros({R}, torch::kFloat32));
    legalAll_ = pin(torch::zeros({R, A}, torch::kFloat32));
    rewardAll_ = pin(torch::zeros({R}, torch::kFloat32));
    terminalAll_ = pin(torch::zeros({R}, torch::kBool));
    keepObs_.resize(T_);
  }

  std::shared_ptr<ModelLocker> locker_;
  std::shared_ptr<FFPrioritizedReplay> replay_;
  std::shared_ptr<RNNPrioritizedReplay> rnnReplay_;
  const bool lstm_ = false;
  const int seqLen_ = 0, burnin_ = 0;
  const int n_, K_;
  const float gamma_;
  const int T_;
  rela_apex_actor* h_ = nullptr;
  rela_r2d2_actor* hr_ = nullptr;
  bool created_ = false;
  void *compute_ = nullptr, *upload_ = nullptr;
  torch::Tensor actionAll_, epsAll_, legalAll_, rewardAll_, terminalAll_;
  std::vector<torch::Tensor> keepObs_;
  std::vector<std::atomic<int64_t>> numAct_;
  bool constsValid_ = false, constsDirty_ = false, draining_ = false;
  std::mutex m_;
  std::condition_variable cv_;
  int arrived_ = 0;
  uint64_t generation_ = 0;
};

// =====================================================================================
// DQNActor (rela/dqn_actor.h:126-211)
// =====================================================================================
class DQNActor : public Actor {
 public:
  DQNActor(std::shared_ptr<ModelLocker> locker, int multiStep, int batchsize, float gamma,
           std::shared_ptr<FFPrioritizedReplay> replay)
      : batchsize_(batchsize), multiStep_(multiStep), gamma_(gamma), locker_(std::move(locker)),
        replay_(std::move(replay)) {}

  // evaluation mode: one env, no replay (dqn_actor.h:141-147)
  explicit DQNActor(std::shared_ptr<ModelLocker> locker)
      : batchsize_(1), multiStep_(1), gamma_(1.f), locker_(std::move(locker)), replay_(nullptr) {}

  ~DQNActor() override {
    rela_apex_actor_destroy(h_);
    if (stream_) rela_stream_destroy(stream_, locker_->deviceIndex);
  }

  int numAct() const {
    if (cohort_) return cohort_->numAct(member_);
    return h_ ? (int)rela_apex_actor_num_act(h_) : 0;
  }

  // batching key: actors that agree on all of these may share one device shard
  bool trainable() const { return replay_ != nullptr; }
  const void* lockerKey() const { return locker_.get(); }
  const void* replayKey() const { return replay_.get(); }
  int batchsize() const { return batchsize_; }
  int multiStep() const { return multiStep_; }
  float gamma() const { return gamma_; }
  std::shared_ptr<ModelLocker> locker() const { return locker_; }
  std::shared_ptr<FFPrioritizedReplay> replay() const { return replay_; }
  void joinCohort(std::shared_ptr<ActorCohort> c, int member) {
    cohort_ = std::move(c);
    member_ = member;
  }
  void onLoopExit() override {
    if (cohort_) cohort_->leave(member_);
  }
  void onShutdown() override {
    if (cohort_) cohort_->shutdown();
    if (replay_) replay_->shutdown();
  }

  TensorDict act(TensorDict& obs) override {
    if (cohort_) return cohort_->act(member_, obs);
    const auto& s = obs.at("s");
    const auto& legal = obs.at("legal_move");
    const auto& eps = obs.at("eps");
    if (s.size(0) != batchsize_ || s.numel() != (int64_t)batchsize_ * kObsBytes || s.dtype() != torch::kUInt8)
      throw std::runtime_error("DQNActor.act: obs['s'] must be uint8 [batchsize,4,84,84]");
    const int A = (int)legal.size(1);
    if (!h_) {
      static std::atomic<uint64_t> counter{0};
      rela_replay* rep = replay_ ? replay_->handle(locker_->deviceIndex, A) : nullptr;
      check(rela_stream_create(&stream_, locker_->deviceIndex), "rela_stream_create");
      check(rela_apex_actor_create(&h_, batchsize_, batchsize_, A, multiStep_, gamma_, rep,
                                   0x9E3779B97F4A7C15ull * (++counter), locker_->deviceIndex),
            "rela_apex_actor_create");
      enableDedup(h_, rep);
      action_ = torch::zeros({batchsize_}, torch::kInt64);
      if (torch::cuda::is_available()) action_ = action_.pin_memory();
      epsHost_ = torch::zeros({batchsize_}, torch::kFloat32);
      legalHost_ = torch::zeros({batchsize_, A}, torch::kFloat32);
    }
    // per-env constants: upload only when they changed
    const float* epsPtr = nullptr;
    const float* legalPtr = nullptr;
    auto e = eps.reshape({batchsize_}).to(torch::kFloat32).contiguous();
    if (!constsValid_ || std::memcmp(e.data_ptr(), epsHost_.data_ptr(), e.nbytes()) != 0) {
      epsHost_.copy_(e);
      epsPtr = epsHost_.data_ptr<float>();
    }
    auto l = legal.to(torch::kFloat32).contiguous();
    if (!constsValid_ || std::memcmp(l.data_ptr(), legalHost_.data_ptr(), l.nbytes()) != 0) {
      legalHost_.copy_(l);
      legalPtr = legalHost_.data_ptr<float>();
    }
    constsValid_ = true;
    auto sc = s.contiguous();
    auto lease = locker_->getModel();
    if (lease.kind != ModelLocker::kFF) {
      locker_->releaseModel(lease.id);
      throw std::runtime_error("DQNActor needs an AtariFFNet-shaped agent in its ModelLocker");
    }
    const int rc = rela_apex_actor_act(h_, static_cast<const rela_ffnet*>(lease.online), sc.data_ptr<uint8_t>(), epsPtr, legalPtr,
                                       action_.data_ptr<int64_t>(), nullptr, stream_);
    locker_->releaseModel(lease.id);
    check(rc, "DQNActor.act");
    return TensorDict{{"a", action_}};
  }

  void setRewardAndTerminal(torch::Tensor& r, torch::Tensor& t) override {
    if (!replay_) throw std::runtime_error("DQNActor: evaluation actor has no replay");  // :175
    if (cohort_) return cohort_->setRewardAndTerminal(member_, r, t);
    reward_ = r.to(torch::kFloat32).contiguous();
    terminal_ = t.to(torch::kBool).contiguous();
  }

  void postStep() override {
    if (!replay_) throw std::runtime_error("DQNActor: evaluation actor has no replay");  // :182
    if (cohort_) return cohort_->postStep(member_);
    auto lease = locker_->getModel();
    const int rc = rela_apex_actor_post_step(h_, reward_.data_ptr<float>(),
                                             reinterpret_cast<const uint8_t*>(terminal_.data_ptr<bool>()), 0,
                                             static_cast<const rela_ffnet*>(lease.online),
                                             static_cast<const rela_ffnet*>(lease.target), 0, nullptr, stream_);
    locker_->releaseModel(lease.id);
    if (rc != RELA_EWOULDBLOCK) check(rc, "DQNActor.postStep");  // dropped block after replay shutdown
  }

 private:
  const int batchsize_, multiStep_;
  const float gamma_;
  std::shared_ptr<ModelLocker> locker_;
  std::shared_ptr<FFPrioritizedReplay> replay_;
  rela_apex_actor* h_ = nullptr;
  void* stream_ = nullptr;  // this actor thread's private HIP stream
  torch::Tensor action_, epsHost_, legalHost_, reward_, terminal_;
  bool constsValid_ = false;
  std::shared_ptr<ActorCohort> cohort_;  // set when this actor is batched with its siblings
  int member_ = -1;
};

// =====================================================================================
// R2D2Actor (rela/r2d2_actor.h:189-353)
// =====================================================================================
class R2D2Actor : public Actor {
 public:
  R2D2Actor(std::shared_ptr<ModelLocker> locker, int multiStep, int batchsize, float gamma, int seqLen, int burnin,
            std::shared_ptr<RNNPrioritizedReplay> replay)
      : batchsize_(batchsize), multiStep_(multiStep), gamma_(gamma), seqLen_(seqLen), burnin_(burnin),
        locker_(std::move(locker)), replay_(std::move(replay)) {
    if (burnin_ > seqLen_ || multiStep_ > seqLen_)  // r2d2_actor.h:25-26
      throw std::invalid_argument("R2D2Actor needs burn_in <= seq_len and multi_step <= seq_len");
  }

  // evaluation mode (r2d2_actor.h:208-215)
  explicit R2D2Actor(std::shared_ptr<ModelLocker> locker)
      : batchsize_(1), multiStep_(1), gamma_(1.f), seqLen_(1), burnin_(0), locker_(std::move(locker)),
        replay_(nullptr) {}

  ~R2D2Actor() override {
    rela_r2d2_actor_destroy(h_);
    if (stream_) rela_stream_destroy(stream_, locker_->deviceIndex);
  }

  int numAct() const {
    if (cohort_) return cohort_->numAct(member_);
    return h_ ? (int)rela_r2d2_actor_num_act(h_) : 0;
  }
  // batching key: actors that agree on all of these may share one device shard
  bool trainable() const { return replay_ != nullptr; }
  const void* lockerKey() const { return locker_.get(); }
  const void* replayKey() const { return replay_.get(); }
  int batchsize() const { return batchsize_; }
  int multiStep() const { return multiStep_; }
  float gamma() const { return gamma_; }
  int seqLen() const { return seqLen_; }
  int burnin() const { return burnin_; }
  std::shared_ptr<ModelLocker> locker() const { return locker_; }
  std::shared_ptr<RNNPrioritizedReplay> replay() const { return replay_; }
  void joinCohort(std::shared_ptr<ActorCohort> c, int member) {
    cohort_ = std::move(c);
    member_ = member;
  }
  void onLoopExit() override {
    if (cohort_) cohort_->leave(member_);
  }
  void onShutdown() override {
    if (cohort_) cohort_->shutdown();
    if (replay_) replay_->shutdown();
  }

  TensorDict act(TensorDict& obs) override {
    if (cohort_) return cohort_->act(member_, obs);
    const auto& s = obs.at("s");
    const auto& legal = obs.at("legal_move");
    const auto& eps = obs.at("eps");
    if (s.size(0) != batchsize_ || s.numel() != (int64_t)batchsize_ * kObsBytes || s.dtype() != torch::kUInt8)
      throw std::runtime_error("R2D2Actor.act: obs['s'] must be uint8 [batchsize,4,84,84]");
    const int A = (int)legal.size(1);
    if (!h_) {
      static std::atomic<uint64_t> counter{0};
      const int T = burnin_ + seqLen_ + multiStep_;
      rela_replay* rep = replay_ ? replay_->handle(locker_->deviceIndex, A, T) : nullptr;
      check(rela_stream_create(&stream_, locker_->deviceIndex), "rela_stream_create");
      check(rela_r2d2_actor_create(&h_, batchsize_, batchsize_, A, multiStep_, gamma_, seqLen_, burnin_,
                                   locker_->eta(), rep, 0xD1B54A32D192ED03ull * (++counter), locker_->deviceIndex),
            "rela_r2d2_actor_create");
      action_ = torch::zeros({batchsize_}, torch::kInt64);
      if (torch::cuda::is_available()) action_ = action_.pin_memory();
      epsHost_ = torch::zeros({batchsize_}, torch::kFloat32);
      legalHost_ = torch::zeros({batchsize_, A}, torch::kFloat32);
    }
    const float* epsPtr = nullptr;
    const float* legalPtr = nullptr;
    auto e = eps.reshape({batchsize_}).to(torch::kFloat32).contiguous();
    if (!constsValid_ || std::memcmp(e.data_ptr(), epsHost_.data_ptr(), e.nbytes()) != 0) {
      epsHost_.copy_(e);
      epsPtr = epsHost_.data_ptr<float>();
    }
    auto l = legal.to(torch::kFloat32).contiguous();
    if (!constsValid_ || std::memcmp(l.data_ptr(), legalHost_.data_ptr(), l.nbytes()) != 0) {
      legalHost_.copy_(l);
      legalPtr = legalHost_.data_ptr<float>();
    }
    constsValid_ = true;
    auto sc = s.contiguous();
    auto lease = locker_->getModel();
    if (lease.kind != ModelLocker::kLSTM) {
      locker_->releaseModel(lease.id);
      throw std::runtime_error("R2D2Actor needs an AtariLSTMNet-shaped agent in its ModelLocker");
    }
    const int rc = rela_r2d2_actor_act(h_, static_cast<const rela_lstmnet*>(lease.online), sc.data_ptr<uint8_t>(),
                                       epsPtr, legalPtr, action_.data_ptr<int64_t>(), nullptr, stream_);
    locker_->releaseModel(lease.id);
    check(rc, "R2D2Actor.act");
    return TensorDict{{"a", action_}};
  }

  void setRewardAndTerminal(torch::Tensor& r, torch::Tensor& t) override {
    if (!replay_) throw std::runtime_error("R2D2Actor: evaluation actor has no replay");
    if (cohort_) return cohort_->setRewardAndTerminal(member_, r, t);
    reward_ = r.to(torch::kFloat32).contiguous();
    terminal_ = t.to(torch::kBool).contiguous();
  }

  void postStep() override {
    if (!replay_) throw std::runtime_error("R2D2Actor: evaluation actor has no replay");
    if (cohort_) return cohort_->postStep(member_);
    auto lease = locker_->getModel();
    const int rc = rela_r2d2_actor_post_step(h_, reward_.data_ptr<float>(),
                                             reinterpret_cast<const uint8_t*>(terminal_.data_ptr<bool>()),
                                             static_cast<const rela_lstmnet*>(lease.online),
                                             static_cast<const rela_lstmnet*>(lease.target), 0, nullptr, stream_);
    locker_->releaseModel(lease.id);
    if (rc != RELA_EWOULDBLOCK) check(rc, "R2D2Actor.postStep");  // dropped block after replay shutdown
  }

 private:
  const int batchsize_, multiStep_;
  const float gamma_;
  const int seqLen_, burnin_;
  std::shared_ptr<ModelLocker> locker_;
  std::shared_ptr<RNNPrioritizedReplay> replay_;
  rela_r2d2_actor* h_ = nullptr;
  void* stream_ = nullptr;
  torch::Tensor action_, epsHost_, legalHost_, reward_, terminal_;
  bool constsValid_ = false;
  std::shared_ptr<ActorCohort> cohort_;  // set when this actor is batched with its siblings
  int member_ = -1;
};

// =====================================================================================
// ThreadLoop / BasicThreadLoop / Context (rela/thread_loop.h:12-111, rela/context.h:14-76)
// =====================================================================================
class ThreadLoop {
 public:
  ThreadLoop() = default;
  ThreadLoop(const ThreadLoop&) = delete;
  ThreadLoop& operator=(const ThreadLoop&) = delete;
  virtual ~ThreadLoop() = default;

  virtual void terminate() { stop_.store(true); }
  virtual void pause() {
    std::lock_guard<std::mutex> lk(m_);
    paused_ = true;
  }
  virtual void resume() {
    {
      std::lock_guard<std::mutex> lk(m_);
      paused_ = false;
    }
    cv_.notify_all();
  }
  virtual bool terminated() { return stop_.load(); }
  virtual void mainLoop() = 0;
  virtual std::shared_ptr<Actor> actor() const { return nullptr; }

 protected:
  // blocks while paused; like the reference, terminate() alone does not wake a paused loop
  void pauseGate() {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [this] { return !paused_; });
  }

 private:
  std::atomic<bool> stop_{false};
  std::mutex m_;
  std::condition_variable cv_;
  bool paused_ = false;
};

class BasicThreadLoop : public ThreadLoop {
 public:
  BasicThreadLoop(std::shared_ptr<Actor> actor, std::shared_ptr<VectorEnv> env, bool eval)
      : actor_(std::move(actor)), env_(std::move(env)), eval_(eval) {
    if (eval_ && env_->size() != 1) throw std::invalid_argument("eval thread loops drive exactly one env");
  }

  void mainLoop() final {
    TensorDict obs;
    torch::Tensor r, t;
    while (!terminated()) {
      {
        StatTimer st(gStats.envReset);
        obs = env_->reset(obs);
      }
      while (!env_->anyTerminated() && !terminated()) {
        pauseGate();
        TensorDict action = actor_->act(obs);
        {
          StatTimer st(gStats.envStep);
          std::tie(obs, r, t) = env_->step(action);
        }
        gStats.envSteps += env_->size();
        if (eval_) continue;
        {
          StatTimer st(gStats.setRT);
          actor_->setRewardAndTerminal(r, t);
        }
        actor_->postStep();
      }
      if (eval_) break;  // one episode
    }
    actor_->onLoopExit();
  }

  std::shared_ptr<Actor> actor() const override { return actor_; }

 private:
  std::shared_ptr<Actor> actor_;
  std::shared_ptr<VectorEnv> env_;
  const bool eval_;
};

class Context {
 public:
  Context() = default;
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;

  ~Context() {
    for (auto& l : loops_) l->terminate();
    for (auto& l : loops_) l->resume();  // unlike the reference, never leave a paused thread unjoinable
    for (auto& l : loops_)
      if (auto a = l->actor()) a->onShutdown();  // ... nor one parked on a full replay ring
    for (auto& th : threads_)
      if (th.joinable()) th.join();
    gStats.print();
  }

  int pushThreadLoop(std::shared_ptr<ThreadLoop> loop) {
    if (started_) throw std::runtime_error("Context: push_env_thread after start");
    loops_.push_back(std::move(loop));
    return (int)loops_.size();
  }

  void start() {
    started_ = true;
    formCohorts();
    for (size_t i = 0; i < loops_.size(); ++i) {
      threads_.emplace_back([this, i] {
        // The reference lets an exception on an actor thread reach std::terminate (rela/context.h:39-46).
        // Here the first one is kept and re-raised on the Python thread by terminated(); the loop counts
        // as done, its cohort is told (the siblings drain instead of waiting at the barrier forever) and
        // every other loop is asked to stop, so the learner fails with a traceback instead of an abort.
        try {
          loops_[i]->mainLoop();
        } catch (...) {
          {
            std::lock_guard<std::mutex> lk(errM_);
            if (!error_) error_ = std::current_exception();
          }
          try {
            if (auto a = loops_[i]->actor()) a->onLoopExit();
          } catch (...) {
          }
          for (auto& l : loops_) l->terminate();
        }
        ++done_;
      });
    }
  }
  void pause() {
    for (auto& l : loops_) l->pause();
  }
  void resume() {
    for (auto& l : loops_) l->resume();
  }
  void terminate() {
    for (auto& l : loops_) l->terminate();
  }
  bool terminated() {
    {
      std::lock_guard<std::mutex> lk(errM_);
      if (error_) {
        std::exception_ptr e = error_;
        error_ = nullptr;  // raised once; afterwards terminated() reports the joined state
        std::rethrow_exception(e);
      }
    }
    return done_.load() == (int)loops_.size();
  }

 private:
  // Training DQNActors (R2D2Actors) of this context that share (locker, replay, K, n, gamma[, seq_len,
  // burn_in]) are batched into one device shard (ActorCohort); a lone actor keeps its private shard.
  // RELA_NO_COHORT=1 opts out.
  template <class ActorT, class Same, class Make>
  void formCohortsOf(Same same, Make make) {
    std::vector<std::vector<std::shared_ptr<ActorT>>> buckets;
    for (auto& l : loops_) {
      auto a = std::dynamic_pointer_cast<ActorT>(l->actor());
      if (!a || !a->trainable()) continue;
      bool placed = false;
      for (auto& b : buckets) {
        if (same(*b.front(), *a)) {
          b.push_back(a);
          placed = true;
          break;
        }
      }
      if (!placed) buckets.push_back({a});
    }
    for (auto& b : buckets) {
      if (b.size() < 2) continue;
      auto cohort = make(*b.front(), (int)b.size());
      for (size_t i = 0; i < b.size(); ++i) b[i]->joinCohort(cohort, (int)i);
    }
  }

  void formCohorts() {
    if (const char* off = std::getenv("RELA_NO_COHORT"))
      if (off[0] == '1') return;
    formCohortsOf<DQNActor>(
        [](const DQNActor& f, const DQNActor& a) {
          return f.lockerKey() == a.lockerKey() && f.replayKey() == a.replayKey() && f.batchsize() == a.batchsize() &&
                 f.multiStep() == a.multiStep() && f.gamma() == a.gamma();
        },
        [](const DQNActor& f, int members) {
          return std::make_shared<ActorCohort>(f.locker(), f.replay(), f.multiStep(), f.batchsize(), f.gamma(), members);
        });
    formCohortsOf<R2D2Actor>(
        [](const R2D2Actor& f, const R2D2Actor& a) {
          return f.lockerKey() == a.lockerKey() && f.replayKey() == a.replayKey() && f.batchsize() == a.batchsize() &&
                 f.multiStep() == a.multiStep() && f.gamma() == a.gamma() && f.seqLen() == a.seqLen() &&
                 f.burnin() == a.burnin();
        },
        [](const R2D2Actor& f, int members) {
          return std::make_shared<ActorCohort>(f.locker(), f.replay(), f.multiStep(), f.batchsize(), f.gamma(),
                                               f.seqLen(), f.burnin(), members);
        });
  }

  bool started_ = false;
  std::mutex errM_;
  std::exception_ptr error_;  // first exception thrown on an actor thread
  std::atomic<int> done_{0};
  std::vector<std::shared_ptr<ThreadLoop>> loops_;
  std::vector<std::thread> threads_;
};

}  // namespace rela

PYBIND11_MODULE(rela, m) {
  m.doc() = "MI355X-native drop-in for facebookresearch/rela's `rela` module (C ABI: include/rela_amd.h)";

  py::class_<FFTransition, std::shared_ptr<FFTransition>>(m, "FFTransition")
      .def_readwrite("obs", &FFTransition::obs)
      .def_readwrite("action", &FFTransition::action)
      .def_readwrite("reward", &FFTransition::reward)
      .def_readwrite("terminal", &FFTransition::terminal)
      .def_readwrite("bootstrap", &FFTransition::bootstrap)
      .def_readwrite("next_obs", &FFTransition::nextObs);

  py::class_<RNNTransition, std::shared_ptr<RNNTransition>>(m, "RNNTransition")
      .def_readwrite("obs", &RNNTransition::obs)
      .def_readwrite("h0", &RNNTransition::h0)
      .def_readwrite("action", &RNNTransition::action)
      .def_readwrite("reward", &RNNTransition::reward)
      .def_readwrite("terminal", &RNNTransition::terminal)
      .def_readwrite("bootstrap", &RNNTransition::bootstrap)
      .def_readwrite("seq_len", &RNNTransition::seqLen);

  py::class_<FFPrioritizedReplay, std::shared_ptr<FFPrioritizedReplay>>(m, "FFPrioritizedReplay")
      .def(py::init<int, int, float, float, int>())  // capacity, seed, alpha, beta, prefetch
      .def("size", &FFPrioritizedReplay::size)
      .def("num_add", &FFPrioritizedReplay::numAdd)
      .def("sample", &FFPrioritizedReplay::sample)
      .def("update_priority", &FFPrioritizedReplay::updatePriority)
      .def("last_sample_raw", &FFPrioritizedReplay::lastSampleRaw_);  // partition exchange only

  py::class_<RNNPrioritizedReplay, std::shared_ptr<RNNPrioritizedReplay>>(m, "RNNPrioritizedReplay")
      .def(py::init<int, int, float, float, int>())
      .def("size", &RNNPrioritizedReplay::size)
      .def("num_add", &RNNPrioritizedReplay::numAdd)
      .def("sample", &RNNPrioritizedReplay::sample)
      .def("update_priority", &RNNPrioritizedReplay::updatePriority)
      .def("last_sample_raw", &RNNPrioritizedReplay::lastSampleRaw_);  // partition exchange only

  py::class_<Env, std::shared_ptr<Env>>(m, "Env");

  py::class_<VectorEnv, std::shared_ptr<VectorEnv>>(m, "VectorEnv")
      .def(py::init<>())
      .def("append", &VectorEnv::append, py::keep_alive<1, 2>());

  py::class_<ThreadLoop, std::shared_ptr<ThreadLoop>>(m, "ThreadLoop");

  py::class_<BasicThreadLoop, ThreadLoop, std::shared_ptr<BasicThreadLoop>>(m, "BasicThreadLoop")
      .def(py::init<std::shared_ptr<Actor>, std::shared_ptr<VectorEnv>, bool>());

  py::class_<Context>(m, "Context")
      .def(py::init<>())
      .def("push_env_thread", &Context::pushThreadLoop, py::keep_alive<1, 2>())
      .def("start", &Context::start)
      .def("pause", &Context::pause)
      .def("resume", &Context::resume)
      .def("terminate", &Context::terminate)
      .def("terminated", &Context::terminated);

  py::class_<ModelLocker, std::shared_ptr<ModelLocker>>(m, "ModelLocker")
      .def(py::init<std::vector<py::object>, const std::string&>())
      .def("update_model", &ModelLocker::updateModel);

  py::class_<Actor, std::shared_ptr<Actor>>(m, "Actor");

  py::class_<DQNActor, Actor, std::shared_ptr<DQNActor>>(m, "DQNActor")
      .def(py::init<std::shared_ptr<ModelLocker>, int, int, float, std::shared_ptr<FFPrioritizedReplay>>())
      .def(py::init<std::shared_ptr<ModelLocker>>())
      .def("num_act", &DQNActor::numAct);

  py::class_<R2D2Actor, Actor, std::shared_ptr<R2D2Actor>>(m, "R2D2Actor")
      .def(py::init<std::shared_ptr<ModelLocker>, int, int, float, int, int, std::shared_ptr<RNNPrioritizedReplay>>())
      .def(py::init<std::shared_ptr<ModelLocker>>())
      .def("num_act", &R2D2Actor::numAct);
}
