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

#include <algorithm>
#include <atomic>
#include <climits>
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
  // {"env_steps": n, "ticks": n, "<phase>_us": thread-microseconds per env-step since the last reset}; reset = true
  // zeroes the counters (diagnostics for rela_amd/pyrela/benchmark.py; not part of the reference's surface)
  py::dict snapshot(bool reset) {
    py::dict d;
    const double n = std::max<double>(1.0, (double)envSteps.load());
    d["env_steps"] = envSteps.load();
    d["ticks"] = ticks.load();
    auto put = [&](const char* k, std::atomic<int64_t>& v) {
      d[k] = v.load() * 1e-3 / n;
      if (reset) v.store(0);
    };
    put("env_step_us", envStep);
    put("env_reset_us", envReset);
    put("act_prep_upload_us", actPrep);
    put("act_barrier_wait_us", actWait);
    put("act_leader_us", actLead);
    put("set_reward_terminal_us", setRT);
    put("post_barrier_wait_us", postWait);
    put("post_leader_us", postLead);
    if (reset) {
      envSteps.store(0);
      ticks.store(0);
    }
    return d;
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

void VectorEnv::append(std::shared_ptr<Env> env) {
  if (!batch_.empty()) throw std::runtime_error("VectorEnv.append after the first reset()");
  frameEnvs_.push_back(dynamic_cast<FrameRowEnv*>(env.get()));
  envs_.push_back(std::move(env));
}

// Allocates the persistent batch (one page-locked tensor per observation key, shaped [K, ...] like the first
// observation) and caches the raw row addresses; binds the rows of envs that render in place.
void VectorEnv::createBatch(const TensorDict& firstObs) {
  const int K = (int)envs_.size();
  auto pin = [](torch::Tensor t) { return torch::cuda::is_available() ? t.pin_memory() : t; };
  for (const auto& kv : firstObs) {
    std::vector<int64_t> shape{K};
    for (auto d : kv.second.sizes()) shape.push_back(d);
    auto t = pin(torch::zeros(shape, kv.second.options().device(torch::kCPU)));
    batch_.emplace(kv.first, t);
    KeyRows kr;
    kr.key = kv.first;
    kr.base = static_cast<uint8_t*>(t.data_ptr());
    kr.rowBytes = K > 0 ? (int64_t)t.nbytes() / K : 0;
    kr.dtype = t.scalar_type();
    rows_.push_back(kr);
  }
  sliding_ = K > 0;
  for (auto* f : frameEnvs_) sliding_ = sliding_ && f && f->slidingStack();
  if (sliding_ && batch_.count("s")) {
    auto flags = pin(torch::ones({K}, torch::kUInt8));
    restart_ = flags.data_ptr<uint8_t>();
    batch_.emplace("__stack_restart", flags);
  } else {
    sliding_ = false;
  }
}

void VectorEnv::storeRow(const TensorDict& obs, int row) {
  const bool first = batch_.empty();
  if (first) createBatch(obs);
  for (const auto& kv : obs) {
    const KeyRows* kr = nullptr;
    for (const auto& r : rows_)
      if (r.key == kv.first) {
        kr = &r;
        break;
      }
    if (!kr) throw std::runtime_error("VectorEnv: observation key '" + kv.first + "' was not in the first observation");
    const auto& src = kv.second;
    uint8_t* dst = kr->base + (int64_t)row * kr->rowBytes;
    if (src.data_ptr() == dst) continue;  // rendered in place (FrameRowEnv)
    if (src.is_contiguous() && src.device().is_cpu() && src.scalar_type() == kr->dtype && (int64_t)src.nbytes() == kr->rowBytes) {
      std::memcpy(dst, src.data_ptr(), (size_t)kr->rowBytes);
    } else {
      batch_.at(kv.first)[row].copy_(src);
    }
  }
}

TensorDict VectorEnv::reset(const TensorDict& previous) {
  const bool first = previous.empty();
  for (size_t i = 0; i < envs_.size(); ++i) {
    if (first || envs_[i]->terminated()) {
      storeRow(envs_[i]->reset(), (int)i);
      if (restart_) restart_[i] = 1;
    }
  }
  if (first) {  // the rows hold every env's first observation: envs that can, render into them from now on
    for (const auto& r : rows_)
      if (r.key == "s" && r.rowBytes == kObsBytes)
        for (size_t i = 0; i < envs_.size(); ++i)
          if (frameEnvs_[i]) frameEnvs_[i]->bindFrameRow(r.base + (int64_t)i * r.rowBytes);
  }
  return batch_;
}

std::tuple<TensorDict, torch::Tensor, torch::Tensor> VectorEnv::step(const TensorDict& action) {
  const int K = (int)envs_.size();
  if (!reward_.defined()) {
    reward_ = torch::zeros({K}, torch::kFloat32);
    terminal_ = torch::zeros({K}, torch::kBool);
    // per-env action dicts over ONE persistent buffer: {"a": 0-dim int64 view of actionBuf_[i]}
    actionBuf_ = torch::zeros({K}, torch::kInt64);
    envAction_.resize(K);
    for (int i = 0; i < K; ++i)
      envAction_[i].emplace("a", torch::from_blob(actionBuf_.data_ptr<int64_t>() + i, {}, torch::kInt64));
  }
  float* r = reward_.data_ptr<float>();
  bool* t = terminal_.data_ptr<bool>();
  // the common case -- one key "a", int64[K] on the host -- needs no tensor operation per env
  const torch::Tensor* a = nullptr;
  if (action.size() == 1) {
    auto it = action.find("a");
    if (it != action.end() && it->second.scalar_type() == torch::kInt64 && it->second.device().is_cpu() &&
        it->second.is_contiguous() && it->second.numel() == K)
      a = &it->second;
  }
  if (a) std::memcpy(actionBuf_.data_ptr<int64_t>(), a->data_ptr<int64_t>(), (size_t)K * sizeof(int64_t));
  for (int i = 0; i < K; ++i) {
    TensorDict sliced;
    if (!a)
      for (const auto& kv : action) sliced.emplace(kv.first, kv.second[i]);
    TensorDict obs;
    float reward;
    bool terminal;
    std::tie(obs, reward, terminal) = envs_[i]->step(a ? envAction_[i] : sliced);
    storeRow(obs, i);
    if (restart_) restart_[i] = 0;
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

  // A "cpu" locker (the eval locker of pyrela/main.py:116, BASELINE C1's CPU-actor plumbing, pyrela/eval.py:9-36) keeps
  // its Python replicas on the host exactly as the reference does, and its ACTORS run on the GPU `execDevice`
  // (RELA_CPU_LOCKER_DEVICE, default 0) in the exact f32 parity mode whatever RELA_PRECISION says: there is no CPU
  // execution path in this engine, and the f32 mode is the one pinned to the reference's CPU results.
  ModelLocker(std::vector<py::object> pyModels, const std::string& device)
      : device(device), deviceIndex(parseDevice(device)), execDevice(deviceIndex >= 0 ? deviceIndex : cpuLockerDevice()),
        pyModels_(std::move(pyModels)) {
    if (pyModels_.empty()) throw std::invalid_argument("ModelLocker needs at least one model");
    const size_t n = pyModels_.size();
    online_.assign(n, nullptr);
    target_.assign(n, nullptr);
    inFlight_.assign(n, 0);
    if (py::hasattr(pyModels_[0], "eta")) eta_ = pyModels_[0].attr("eta").cast<double>();
    if (deviceIndex >= 0 || torch::cuda::is_available()) loadSlot(0, pyModels_[0]);
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
    if (deviceIndex < 0) pyModels_[id].attr("load_state_dict")(pyModel.attr("state_dict")());  // model_locker.h:31
    if (deviceIndex >= 0 || torch::cuda::is_available()) loadSlot(id, pyModel);
    std::lock_guard<std::mutex> lk(m_);
    latest_ = id;
  }

  Lease getModel() {
    std::lock_guard<std::mutex> lk(m_);
    if (!online_[latest_])
      throw std::runtime_error("ModelLocker('cpu'): its actors run on a GPU in f32 mode and no HIP device is visible; "
                               "this engine has no CPU actor path");
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

  const std::string device;  // what the caller asked for (public member of the reference class, model_locker.h:54)
  const int deviceIndex;     // -1 for "cpu"
  const int execDevice;      // the GPU this locker's nets live on and its actors run on

 private:
  static int cpuLockerDevice() {
    const char* e = std::getenv("RELA_CPU_LOCKER_DEVICE");
    return e ? std::atoi(e) : 0;
  }
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
      out.push_back(t.to(torch::Device(torch::kCUDA, (c10::DeviceIndex)execDevice), torch::kFloat32).contiguous());
    }
    return out;
  }

  // RELA_PRECISION: "f32x3" = conv2 / conv3 / fc of the AtariFFNet actors with f32 operands as three exact bf16 parts on the
  // bf16 matrix cores (f32 accuracy, csrc/gemm_f32emu.h; conv2 / conv3 from 512 rows, fc from 4,096; the recurrent net:
  // conv2 / conv3 of its trunk); "bf16x2" = the fast mode (two bf16 parts: 16-bit significands, |dQ| < 2e-5 max|Q|);
  // anything else = exact f32 MFMA.  "cpu" lockers: always exact f32.
  int precisionMode(bool /*recurrent*/) const {
    const char* e = std::getenv("RELA_PRECISION");
    if (deviceIndex < 0 || !e) return 0;
    const std::string m(e);
    if (m == "bf16x2") return 1;
    if (m == "f32x3") return 2;
    return 0;
  }

  void loadNet(void*& net, py::dict& sd, const std::string& prefix) {
    void* stream = torchCurrentStream(execDevice);
    std::vector<torch::Tensor> t;
    if (kind_ == kFF) {
      t = fetch(sd, prefix, {"net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias",
                             "linear.0.weight", "linear.0.bias", "fc_v.weight", "fc_v.bias", "fc_a.weight", "fc_a.bias"});
      const int A = (int)t[10].size(0);
      checkActions(A);
      auto* n = static_cast<rela_ffnet*>(net);
      if (!n) {
        check(rela_ffnet_create(&n, A, execDevice), "rela_ffnet_create");
        check(rela_ffnet_set_precision(n, precisionMode(false)), "rela_ffnet_set_precision");
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
        check(rela_lstmnet_create(&n, A, execDevice), "rela_lstmnet_create");
        check(rela_lstmnet_set_precision(n, precisionMode(true)), "rela_lstmnet_set_precision");
      }
      net = n;
      auto f = [&](int i) { return t[i].data_ptr<float>(); };
      rela_lstmnet_params p{f(0), f(1), f(2), f(3), f(4), f(5), f(6), f(7), f(8), f(9), f(10), f(11), f(12), f(13)};
      check(rela_lstmnet_load(n, &p, 1, stream), "rela_lstmnet_load");
    }
    // the packing kernels read `t` on torch's stream: finish them before the tensors die
    py::module_::import("torch").attr("cuda").attr("current_stream")(execDevice).attr("synchronize")();
  }

  void checkActions(int A) {
    if (numAction_ == 0) numAction_ = A;
    if (A != numAction_) throw std::runtime_error("ModelLocker: num_action changed between updates");
  }

  void loadSlot(int id, py::object& pyModel) {
    // Leases are released when an actor has QUEUED its kernels; drain the device so nothing that
    // still reads this slot's old weights is in flight (the reference's model call is synchronous).
    py::module_::import("torch").attr("cuda").attr("synchronize")(execDevice);
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
  std::atomic<int> numAction_{0};
  Kind kind_ = kFF;
  bool kindKnown_ = false;
  double eta_ = 0.9;
  std::mutex m_;
  std::condition_variable cv_;
};

// =====================================================================================
// One Python replay object = one PARTITION per ModelLocker that feeds it.
//
// The reference builds one ModelLocker per act device in ONE process and deals the actor threads round-robin onto
// them, all inserting into the single host-RAM replay (pyrela/main.py:131-136,155,166).  Here a replay partition
// lives in the HBM of the GPU whose actors fill it, so the one Python object owns one partition per locker (= per act
// device in the reference's wiring; two lockers on one device give two partitions on it, which is how a one-GPU box
// tests the path): capacity / G slots each, generator seed + g.  sample(B) draws B / G rows from every partition --
// each bit-identical to a reference PrioritizedReplay(capacity / G, seed + g) fed that partition's insertion stream
// and asked for B / G -- concatenates them on the requested device (peer copies over xGMI between GPUs) and
// normalises the importance weights over ALL partitions:  w_i = (N_total p_i / (G sum_g))^-beta / max  (every
// partition contributes exactly B / G of the B draws, so item i of partition g is drawn with probability
// p_i / (G sum_g); equal to prioritized_replay.h:320-322 when G = 1, where the library's own weights are returned
// untouched).  update_priority routes slice g of the priorities to partition g.  Context.start() announces the
// lockers (plan); a replay nobody planned has one partition.
// =====================================================================================
static std::tuple<torch::Tensor, torch::Tensor, int> lastSampleRaw(rela_replay* h, int device, int n);

class ReplayParts {
 public:
  struct Part {
    rela_replay* h = nullptr;
    int device = -1;
    const void* key = nullptr;
  };

  ReplayParts(int capacity, int seed, float alpha, float beta, int prefetch)
      : capacity_(capacity), seed_(seed), alpha_(alpha), beta_(beta), prefetch_(prefetch) {}
  ~ReplayParts() {
    for (auto& p : parts_) rela_replay_destroy(p.h);
  }

  void plan(const std::vector<const void*>& keys) {
    std::lock_guard<std::mutex> lk(m_);
    if (!parts_.empty()) return;  // (a second Context on the same replay: the partitions exist already)
    planned_ = keys;
  }

  // the partition of the locker `key` on `device`, created by its first actor; setSchema(h) fixes the row layout.
  // Partition g = the g-th locker in the order the Context met them (thread order), whichever thread acts first:
  // its generator seed is seed + g.
  template <class F>
  rela_replay* handle(const void* key, int device, F&& setSchema) {
    std::lock_guard<std::mutex> lk(m_);
    if (planned_.empty()) planned_.push_back(key);  // nobody planned: one partition, the first locker's
    const int G = (int)planned_.size();
    if (parts_.empty()) parts_.resize(G);
    int g = -1;
    for (int i = 0; i < G; ++i)
      if (planned_[i] == key) g = i;
    if (g < 0)
      throw std::runtime_error("replay: an actor of a ModelLocker the Context did not announce (one partition per "
                               "locker; all actors must be pushed before Context.start())");
    Part& p = parts_[g];
    if (p.h) {
      if (p.device != device) throw std::runtime_error("replay partition: one ModelLocker lives on one device");
      return p.h;
    }
    if (capacity_ / G < 1) throw std::runtime_error("replay: capacity smaller than the number of partitions");
    rela_replay* h = nullptr;
    check(rela_replay_create(&h, capacity_ / G, seed_ + g, alpha_, beta_, prefetch_, device), "rela_replay_create");
    try {
      setSchema(h);
      // actors and the Python sampler are independent threads here: keep inserts off the sample chain's stream
      check(rela_replay_set_decoupled_insert(h, 1), "rela_replay_set_decoupled_insert");
    } catch (...) {
      rela_replay_destroy(h);
      throw;
    }
    p.key = key;
    p.device = device;
    p.h = h;
    return h;
  }

  // One partition: its size.  Several: G x the smallest (0 until every planned partition exists) -- a training loop
  // gates its first sample on size() >= burn_in (pyrela/main.py:206), and sample() draws B / G from EVERY partition:
  // the sum over partitions would open the gate while one act device has not inserted yet.
  int size() const {
    std::lock_guard<std::mutex> lk(m_);
    const int G = (int)planned_.size();
    if (G <= 1) {
      int n = 0;
      for (auto& p : parts_)
        if (p.h) n += rela_replay_size(p.h);
      return n;
    }
    int least = INT_MAX;
    for (int g = 0; g < G; ++g) least = std::min(least, g < (int)parts_.size() && parts_[g].h ? rela_replay_size(parts_[g].h) : 0);
    return least * G;
  }
  int numAdd() const {
    std::lock_guard<std::mutex> lk(m_);
    int64_t n = 0;
    for (auto& p : parts_)
      if (p.h) n += rela_replay_num_add(p.h);
    return (int)n;
  }
  void shutdown() {
    std::lock_guard<std::mutex> lk(m_);
    for (auto& p : parts_)
      if (p.h) rela_replay_shutdown(p.h);
  }
  // the partitions created so far, in partition order
  std::vector<Part> parts() const {
    std::lock_guard<std::mutex> lk(m_);
    std::vector<Part> out;
    for (auto& p : parts_)
      if (p.h) out.push_back(p);
    return out;
  }
  int expected() const {
    std::lock_guard<std::mutex> lk(m_);
    return planned_.empty() ? 1 : (int)planned_.size();
  }

  // importance weights of a batch drawn over G partitions (see the class comment); raw[g] / sum[g] on `target`
  torch::Tensor globalWeights(const std::vector<torch::Tensor>& raw, const std::vector<torch::Tensor>& sum,
                              const std::vector<int>& size) const {
    const int G = (int)raw.size();
    double total = 0;
    for (int n : size) total += n;
    std::vector<torch::Tensor> w;
    for (int g = 0; g < G; ++g) w.push_back((raw[g] * (float)total / (sum[g] * (float)G)).pow(-beta_));
    auto all = torch::cat(w, 0);
    return all / all.max();
  }

  void updatePriority(const torch::Tensor& priority, const std::vector<int>& counts, const char* where) {
    auto ps = parts();
    auto p = priority.detach().to(torch::kFloat32).contiguous();
    int64_t total = 0;
    for (int c : counts) total += c;
    if (counts.size() != ps.size() || p.numel() != total)
      throw std::runtime_error(std::string(where) + ": expected the priorities of the last batch");
    int64_t off = 0;
    keep_.clear();
    for (size_t g = 0; g < ps.size(); ++g) {
      auto chunk = ps.size() == 1 ? p : p.narrow(0, off, counts[g]);
      off += counts[g];
      if (chunk.is_cuda()) {
        if (chunk.device().index() != ps[g].device) chunk = chunk.to(torch::Device(torch::kCUDA, (c10::DeviceIndex)ps[g].device));
        chunk = chunk.contiguous();
        check(rela_replay_update_priority(ps[g].h, (int)chunk.numel(), chunk.data_ptr<float>(), 1, torchCurrentStream(ps[g].device)), where);
        keep_.push_back(chunk);  // consumed asynchronously on the replay's stream
      } else {
        chunk = chunk.contiguous();
        check(rela_replay_update_priority(ps[g].h, (int)chunk.numel(), chunk.data_ptr<float>(), 0, nullptr), where);
      }
    }
  }

  // ---- native partition exchange (include/rela_amd.h, rela_amd/parallel.py): this process owns ONE partition that a
  // learner in another process maps through HIP IPC and gathers from itself
  py::bytes exportIpc() const {
    auto ps = parts();
    if (ps.size() != 1) throw std::runtime_error("export_ipc: the replay must hold exactly one partition");
    rela_replay_ipc_desc d;
    check(rela_replay_export_ipc(ps[0].h, &d), "rela_replay_export_ipc");
    return py::bytes(reinterpret_cast<const char*>(&d), sizeof(d));
  }
  // ... for a partition of any size (fields above the chunk size travel as file descriptors, include/rela_amd.h:
  // rela_replay_export_chunks): -> (descriptor bytes, [fd, ...]); the caller sends the descriptors with SCM_RIGHTS
  // and closes them (rela_amd/parallel.py: _FdServer)
  std::pair<py::bytes, std::vector<int>> exportChunks() const {
    auto ps = parts();
    if (ps.size() != 1) throw std::runtime_error("export_chunks: the replay must hold exactly one partition");
    rela_replay_chunk_desc d;
    std::vector<int> fds(RELA_IPC_MAX_FDS, -1);
    check(rela_replay_export_chunks(ps[0].h, &d, fds.data(), RELA_IPC_MAX_FDS), "rela_replay_export_chunks");
    fds.resize(d.nfds);
    return {py::bytes(reinterpret_cast<const char*>(&d), sizeof(d)), fds};
  }
  // sample WITHOUT gathering: ids, raw weights and eviction only; -> (raw weights [n], float sum [1], size)
  std::tuple<torch::Tensor, torch::Tensor, int> sampleIds(int n) {
    auto ps = parts();
    if (ps.size() != 1) throw std::runtime_error("sample_ids: the replay must hold exactly one partition");
    auto opt = torch::TensorOptions().dtype(torch::kFloat32).device(torch::Device(torch::kCUDA, (c10::DeviceIndex)ps[0].device));
    auto scratch = torch::empty({n}, opt);
    check(rela_replay_sample(ps[0].h, n, nullptr, scratch.data_ptr<float>(), torchCurrentStream(ps[0].device)), "sample_ids");
    return lastSampleRaw(ps[0].h, ps[0].device, n);
  }

  const int capacity_, seed_;
  const float alpha_, beta_;
  const int prefetch_;

 private:
  mutable std::mutex m_;
  std::vector<Part> parts_;
  std::vector<const void*> planned_;
  std::vector<torch::Tensor> keep_;
};

// what a PARTITION contributes to the importance weights of a batch drawn over several: the un-normalised
// weights w_i of its last sample, the float sum they were drawn against and the size its weights used
// (prioritized_replay.h:289,261,312)
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

static torch::Device deviceOf(const std::string& device) {
  const int want = parseDevice(device);
  return want < 0 ? torch::Device(torch::kCPU) : torch::Device(torch::kCUDA, (c10::DeviceIndex)want);
}

// =====================================================================================
// FFPrioritizedReplay (rela/prioritized_replay.h:173-348 as bound in pybind.cc:37-47)
// =====================================================================================
class FFPrioritizedReplay {
 public:
  FFPrioritizedReplay(int capacity, int seed, float alpha, float beta, int prefetch)
      : core_(capacity, seed, alpha, beta, prefetch) {}

  ReplayParts& core() { return core_; }

  // the partition of `lockerKey`, created lazily by the first actor that knows the device and the action count
  rela_replay* handle(const void* lockerKey, int device, int numAction) {
    {  // (several cohort leaders call this concurrently on their first act: ADVICE r4)
      int seen = 0;
      if (!numAction_.compare_exchange_strong(seen, numAction) && seen != numAction)
        throw std::runtime_error("FFPrioritizedReplay: actors disagree on the action count");
    }
    return core_.handle(lockerKey, device, [&](rela_replay* h) {
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
        check(rela_replay_set_schema_dedup(h, 10, rb, 0, 1, kObsBytes / ups, ups, guard), "rela_replay_set_schema_dedup");
      } else {
        check(rela_replay_set_schema(h, 10, rb), "rela_replay_set_schema");
      }
    });
  }

  int size() const { return core_.size(); }
  int numAdd() const { return core_.numAdd(); }
  void shutdown() { core_.shutdown(); }

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

  // B rows of one partition, on the partition's own device
  std::tuple<FFTransition, torch::Tensor> samplePart(const ReplayParts::Part& p, int B_) {
    const auto dev = torch::Device(torch::kCUDA, (c10::DeviceIndex)p.device);
    auto u8 = torch::TensorOptions().dtype(torch::kUInt8).device(dev);
    auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
    const int64_t B = B_, A = numAction_;
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
    check(rela_replay_sample(p.h, B_, rows, weight.data_ptr<float>(), torchCurrentStream(p.device)),
          "FFPrioritizedReplay.sample");
    return std::make_tuple(std::move(b), weight);
  }

  std::tuple<FFTransition, torch::Tensor> sampleNow(int batchsize, const std::string& device) {
    auto parts = core_.parts();
    if (parts.empty()) throw std::runtime_error("FFPrioritizedReplay.sample: the replay is empty");
    const int G = (int)parts.size();
    if (G != core_.expected())
      throw std::runtime_error("FFPrioritizedReplay.sample: only " + std::to_string(G) + " of " +
                               std::to_string(core_.expected()) + " partitions have received data yet");
    if (batchsize % G != 0)
      throw std::runtime_error("FFPrioritizedReplay.sample: the batch must split evenly over the " + std::to_string(G) +
                               " partitions (one per ModelLocker)");
    const auto target = deviceOf(device);
    lastBatch_ = batchsize;
    lastDevice_ = device;
    lastCounts_.assign(G, batchsize / G);
    if (G == 1) {
      FFTransition b;
      torch::Tensor weight;
      std::tie(b, weight) = samplePart(parts[0], batchsize);
      if (target != torch::Device(torch::kCUDA, (c10::DeviceIndex)parts[0].device)) {  // learner elsewhere: types.cc:34-43
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
    std::vector<FFTransition> sub(G);
    std::vector<torch::Tensor> raw, sum;
    std::vector<int> size;
    for (int g = 0; g < G; ++g) {
      torch::Tensor w;
      std::tie(sub[g], w) = samplePart(parts[g], batchsize / G);
      auto r = lastSampleRaw(parts[g].h, parts[g].device, batchsize / G);
      raw.push_back(std::get<0>(r).to(target));
      sum.push_back(std::get<1>(r).to(target));
      size.push_back(std::get<2>(r));
    }
    auto catOf = [&](auto get) {
      std::vector<torch::Tensor> v;
      for (int g = 0; g < G; ++g) v.push_back(get(sub[g]).to(target));
      return torch::cat(v, 0);
    };
    FFTransition b;
    for (const char* k : {"s", "eps", "legal_move"}) {
      b.obs[k] = catOf([&](FFTransition& t) { return t.obs.at(k); });
      b.nextObs[k] = catOf([&](FFTransition& t) { return t.nextObs.at(k); });
    }
    b.action["a"] = catOf([](FFTransition& t) { return t.action.at("a"); });
    b.reward = catOf([](FFTransition& t) { return t.reward; });
    b.terminal = catOf([](FFTransition& t) { return t.terminal; });
    b.bootstrap = catOf([](FFTransition& t) { return t.bootstrap; });
    return std::make_tuple(std::move(b), core_.globalWeights(raw, sum, size));
  }

  // native partition exchange (not in the reference's surface): see ReplayParts::exportIpc / sampleIds
  py::bytes exportIpc() { return core_.exportIpc(); }
  std::pair<py::bytes, std::vector<int>> exportChunks() { return core_.exportChunks(); }
  std::tuple<torch::Tensor, torch::Tensor, int> sampleIds(int n) {
    auto r = core_.sampleIds(n);
    lastBatch_ = n;
    lastCounts_.assign(1, n);
    lastDevice_.clear();  // (no device-side prefetch behind the next update_priority: the rows stay where they are)
    return r;
  }

  std::tuple<torch::Tensor, torch::Tensor, int> lastSampleRaw_() {
    auto parts = core_.parts();
    if (parts.size() != 1) throw std::runtime_error("last_sample_raw: this replay has several partitions of its own");
    return lastSampleRaw(parts[0].h, parts[0].device, lastBatch_);
  }

  void updatePriority(const torch::Tensor& priority) {
    if (lastBatch_ == 0) throw std::runtime_error("FFPrioritizedReplay.update_priority: nothing was sampled");
    if (priority.dim() != 1) throw std::invalid_argument("update_priority expects a 1-D tensor");  // :236
    core_.updatePriority(priority, lastCounts_, "FFPrioritizedReplay.update_priority");
    if (core_.prefetch_ > 0 && !lastDevice_.empty() && core_.size() >= lastBatch_) {
      prefetched_.emplace(sampleNow(lastBatch_, lastDevice_));
      prefetchedBatch_ = lastBatch_;
      prefetchedDevice_ = lastDevice_;
    }
  }

 private:
  ReplayParts core_;
  std::atomic<int> numAction_{0};
  int lastBatch_ = 0;
  std::vector<int> lastCounts_;
  std::string lastDevice_;
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
      : core_(capacity, seed, alpha, beta, prefetch) {}

  ReplayParts& core() { return core_; }

  rela_replay* handle(const void* lockerKey, int device, int numAction, int T) {
    {  // (several cohort leaders call this concurrently on their first act: ADVICE r4)
      int64_t seen = 0;
      const int64_t mine = ((int64_t)numAction << 32) | (uint32_t)T;
      if (!shape_.compare_exchange_strong(seen, mine) && seen != mine)
        throw std::runtime_error("RNNPrioritizedReplay: actors disagree on action count / window length");
      numAction_ = numAction;
      T_ = T;
    }
    return core_.handle(lockerKey, device, [&](rela_replay* h) {
      const int64_t A = numAction, t = T;
      const int64_t rb[10] = {t * kObsBytes, t * 4, t * 4 * A, t * 8, t * 4, t, t * 4, 2048, 2048, 4};
      const int32_t st[10] = {T, T, T, T, T, T, T, 1, 1, 1};
      check(rela_replay_set_schema_seq(h, 10, rb, st), "rela_replay_set_schema_seq");
    });
  }

  int size() const { return core_.size(); }
  int numAdd() const { return core_.numAdd(); }
  void shutdown() { core_.shutdown(); }

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

  std::tuple<RNNTransition, torch::Tensor> samplePart(const ReplayParts::Part& p, int B_) {
    const auto dev = torch::Device(torch::kCUDA, (c10::DeviceIndex)p.device);
    auto opt = [&](torch::ScalarType t) { return torch::TensorOptions().dtype(t).device(dev); };
    const int64_t B = B_, A = numAction_, T = T_;
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
    check(rela_replay_sample(p.h, B_, rows, weight.data_ptr<float>(), torchCurrentStream(p.device)),
          "RNNPrioritizedReplay.sample");
    return std::make_tuple(std::move(b), weight);
  }

  std::tuple<RNNTransition, torch::Tensor> sampleNow(int batchsize, const std::string& device) {
    auto parts = core_.parts();
    if (parts.empty()) throw std::runtime_error("RNNPrioritizedReplay.sample: the replay is empty");
    const int G = (int)parts.size();
    if (G != core_.expected())
      throw std::runtime_error("RNNPrioritizedReplay.sample: only " + std::to_string(G) + " of " +
                               std::to_string(core_.expected()) + " partitions have received data yet");
    if (batchsize % G != 0)
      throw std::runtime_error("RNNPrioritizedReplay.sample: the batch must split evenly over the " + std::to_string(G) +
                               " partitions (one per ModelLocker)");
    const auto target = deviceOf(device);
    lastBatch_ = batchsize;
    lastDevice_ = device;
    lastCounts_.assign(G, batchsize / G);
    if (G == 1) {
      RNNTransition b;
      torch::Tensor weight;
      std::tie(b, weight) = samplePart(parts[0], batchsize);
      if (target != torch::Device(torch::kCUDA, (c10::DeviceIndex)parts[0].device)) {
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
    std::vector<RNNTransition> sub(G);
    std::vector<torch::Tensor> raw, sum;
    std::vector<int> size;
    for (int g = 0; g < G; ++g) {
      torch::Tensor w;
      std::tie(sub[g], w) = samplePart(parts[g], batchsize / G);
      auto r = lastSampleRaw(parts[g].h, parts[g].device, batchsize / G);
      raw.push_back(std::get<0>(r).to(target));
      sum.push_back(std::get<1>(r).to(target));
      size.push_back(std::get<2>(r));
    }
    auto catOf = [&](auto get, int dim) {  // time-major fields stack along their batch axis (types.cc:140-182: dim 1)
      std::vector<torch::Tensor> v;
      for (int g = 0; g < G; ++g) v.push_back(get(sub[g]).to(target));
      return torch::cat(v, dim);
    };
    RNNTransition b;
    for (const char* k : {"s", "eps", "legal_move"}) b.obs[k] = catOf([&](RNNTransition& t) { return t.obs.at(k); }, 1);
    b.action["a"] = catOf([](RNNTransition& t) { return t.action.at("a"); }, 1);
    b.reward = catOf([](RNNTransition& t) { return t.reward; }, 1);
    b.terminal = catOf([](RNNTransition& t) { return t.terminal; }, 1);
    b.bootstrap = catOf([](RNNTransition& t) { return t.bootstrap; }, 1);
    for (const char* k : {"h0", "c0"}) b.h0[k] = catOf([&](RNNTransition& t) { return t.h0.at(k); }, 1);
    b.seqLen = catOf([](RNNTransition& t) { return t.seqLen; }, 0);
    return std::make_tuple(std::move(b), core_.globalWeights(raw, sum, size));
  }

  void updatePriority(const torch::Tensor& priority) {
    if (lastBatch_ == 0) throw std::runtime_error("RNNPrioritizedReplay.update_priority: nothing was sampled");
    if (priority.dim() != 1) throw std::invalid_argument("update_priority expects a 1-D tensor");
    core_.updatePriority(priority, lastCounts_, "RNNPrioritizedReplay.update_priority");
    if (core_.prefetch_ > 0 && !lastDevice_.empty() && core_.size() >= lastBatch_) {
      prefetched_.emplace(sampleNow(lastBatch_, lastDevice_));
      prefetchedBatch_ = lastBatch_;
      prefetchedDevice_ = lastDevice_;
    }
  }

  // native partition exchange (not in the reference's surface): see ReplayParts::exportIpc / sampleIds
  py::bytes exportIpc() { return core_.exportIpc(); }
  std::pair<py::bytes, std::vector<int>> exportChunks() { return core_.exportChunks(); }
  std::tuple<torch::Tensor, torch::Tensor, int> sampleIds(int n) {
    auto r = core_.sampleIds(n);
    lastBatch_ = n;
    lastCounts_.assign(1, n);
    lastDevice_.clear();  // (no device-side prefetch behind the next update_priority: the rows stay where they are)
    return r;
  }

  std::tuple<torch::Tensor, torch::Tensor, int> lastSampleRaw_() {
    auto parts = core_.parts();
    if (parts.size() != 1) throw std::runtime_error("last_sample_raw: this replay has several partitions of its own");
    return lastSampleRaw(parts[0].h, parts[0].device, lastBatch_);
  }

 private:
  ReplayParts core_;
  std::atomic<int64_t> shape_{0};  // numAction << 32 | T, set once
  std::atomic<int> numAction_{0}, T_{0};
  int lastBatch_ = 0;
  std::vector<int> lastCounts_;
  std::string lastDevice_;
  std::optional<std::tuple<RNNTransition, torch::Tensor>> prefetched_;
  int prefetchedBatch_ = 0;
  std::string prefetchedDevice_;
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
    const int dev = locker_->execDevice;
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
    void* slot = nullptr;
    bool planes = false;
    {
      std::unique_lock<std::mutex> lk(m_);
      if (draining_) return drained();
      if (!created_) create(A);
      // (the slot moves only inside the leader's work, which every member of the round has left by now)
      slot = lstm_ ? rela_r2d2_actor_obs_slot(hr_) : rela_apex_actor_obs_slot(h_);
      // a VectorEnv whose envs all slide their frame stack marks its batch (rela/env.h): only plane 3 of every row is
      // new; the very first observation has no predecessor on the device and goes up whole
      planes = planeUpload_ && !firstRound_ && obs.count("__stack_restart") != 0;
    }
    // this member's rows: frames go to the HBM history slot on the upload stream -- outside the cohort's lock, the
    // rows of different members are disjoint -- the per-env constants to the host staging under it
    auto sc = s.contiguous();
    const int dev = locker_->execDevice;
    uint8_t* dst = static_cast<uint8_t*>(slot) + (int64_t)member * K_ * kObsBytes;
    uint8_t* flags = restartAll_.data_ptr<uint8_t>() + (int64_t)member * K_;
    if (planes) {
      // the newest plane of each of this member's rows, packed into the cohort's page-locked staging (a strided 2-D
      // DMA of K planes costs the issuing thread 2-3 x a plain copy), then ONE 1-D copy to the shard's plane stage
      constexpr int64_t kPlane = 84 * 84;
      uint8_t* host = planeHost_.data_ptr<uint8_t>() + (int64_t)member * K_ * kPlane;
      const uint8_t* src = sc.data_ptr<uint8_t>() + 3 * kPlane;
      for (int i = 0; i < K_; ++i) std::memcpy(host + i * kPlane, src + (int64_t)i * kObsBytes, (size_t)kPlane);
      check(rela_memcpy_h2d_async(static_cast<uint8_t*>(planeStage_) + (int64_t)member * K_ * kPlane, host, (int64_t)K_ * kPlane,
                                  upload_, dev),
            "rela_memcpy_h2d_async");
      std::memcpy(flags, obs.at("__stack_restart").data_ptr<uint8_t>(), (size_t)K_);
    } else {
      check(rela_memcpy_h2d_async(dst, sc.data_ptr(), (int64_t)K_ * kObsBytes, upload_, dev), "rela_memcpy_h2d_async");
      std::memset(flags, 2, (size_t)K_);  // 2 = the row went up whole: slide_stacks leaves it alone
    }
    auto e = eps.reshape({K_}).to(torch::kFloat32).contiguous();
    auto l = legal.to(torch::kFloat32).contiguous();
    std::unique_lock<std::mutex> lk(m_);
    if (draining_) return drained();
    keepObs_[member] = sc;
    if (planes) planesThisRound_ = true;
    float* ed = epsAll_.data_ptr<float>() + (int64_t)member * K_;
    float* ld = legalAll_.data_ptr<float>() + (int64_t)member * K_ * A;
    if (!constsValid_ || std::memcmp(ed, e.data_ptr(), e.nbytes()) != 0 || std::memcmp(ld, l.data_ptr(), l.nbytes()) != 0) {
      std::memcpy(ed, e.data_ptr(), e.nbytes());
      std::memcpy(ld, l.data_ptr(), l.nbytes());
      constsDirty_ = true;
    }
    if (gStats.on) gStats.actPrep += ThreadedStats::now() - tPrep;
    rendezvous(lk, [&] {
      check(rela_stream_wait_stream(compute_, upload_, dev), "rela_stream_wait_stream");
      if (planesThisRound_) {  // complete the stacks on the device (atari/game_state.h:53-82) before the forward
        const uint8_t* f = restartAll_.data_ptr<uint8_t>();
        check(lstm_ ? rela_r2d2_actor_slide_stacks(hr_, f, compute_) : rela_apex_actor_slide_stacks(h_, f, compute_),
              "slide_stacks");
      }
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
      planesThisRound_ = false;
      firstRound_ = false;
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
      check(rela_stream_wait_stream(upload_, compute_, locker_->execDevice), "rela_stream_wait_stream");
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
    const int dev = locker_->execDevice;
    static std::atomic<uint64_t> counter{0};
    check(rela_stream_create(&compute_, dev), "rela_stream_create");
    check(rela_stream_create(&upload_, dev), "rela_stream_create");
    if (lstm_) {
      // one pop of the shard commits the sequences of all members as ONE block, in row (= member)
      // order; the reference would issue one block per thread (only the float block-sum grouping of
      // sum_ differs, far below the fp tolerance of the priorities themselves)
      rela_replay* rep = rnnReplay_->handle(locker_.get(), dev, A, burnin_ + seqLen_ + n_);
      check(rela_r2d2_actor_create(&hr_, T_ * K_, K_, A, n_, gamma_, seqLen_, burnin_, locker_->eta(), rep,
                                   0xC2B2AE3D27D4EB4Full * (++counter), dev),
            "rela_r2d2_actor_create");
    } else {
      rela_replay* rep = replay_->handle(locker_.get(), dev, A);
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
    restartAll_ = pin(torch::full({R}, 2, torch::kUInt8));
    planeHost_ = pin(torch::zeros({R, 84 * 84}, torch::kUInt8));
    planeStage_ = lstm_ ? rela_r2d2_actor_plane_stage(hr_) : rela_apex_actor_plane_stage(h_);
    if (!planeStage_) throw std::runtime_error("ActorCohort: could not allocate the plane stage");
    keepObs_.resize(T_);
    // RELA_PLANE_UPLOAD=0: always upload whole frame stacks (A/B switch of the sliding-stack path)
    const char* pu = std::getenv("RELA_PLANE_UPLOAD");
    planeUpload_ = !(pu && pu[0] == '0');
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
  torch::Tensor actionAll_, epsAll_, legalAll_, rewardAll_, terminalAll_, restartAll_, planeHost_;
  void* planeStage_ = nullptr;  // device [R][7056]: the newest plane of every row
  std::vector<torch::Tensor> keepObs_;
  std::vector<std::atomic<int64_t>> numAct_;
  bool constsValid_ = false, constsDirty_ = false, draining_ = false;
  bool planeUpload_ = true, firstRound_ = true, planesThisRound_ = false;
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
    if (stream_) rela_stream_destroy(stream_, locker_->execDevice);
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
      rela_replay* rep = replay_ ? replay_->handle(locker_.get(), locker_->execDevice, A) : nullptr;
      check(rela_stream_create(&stream_, locker_->execDevice), "rela_stream_create");
      check(rela_apex_actor_create(&h_, batchsize_, batchsize_, A, multiStep_, gamma_, rep,
                                   0x9E3779B97F4A7C15ull * (++counter), locker_->execDevice),
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
    if (stream_) rela_stream_destroy(stream_, locker_->execDevice);
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
      rela_replay* rep = replay_ ? replay_->handle(locker_.get(), locker_->execDevice, A, T) : nullptr;
      check(rela_stream_create(&stream_, locker_->execDevice), "rela_stream_create");
      check(rela_r2d2_actor_create(&h_, batchsize_, batchsize_, A, multiStep_, gamma_, seqLen_, burnin_,
                                   locker_->eta(), rep, 0xD1B54A32D192ED03ull * (++counter), locker_->execDevice),
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
  // RELA_COHORT_SPLIT=0 opts out.
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
    // RELA_COHORT_SPLIT (default 2): a bucket of >= 16 threads becomes that many cohorts of consecutive threads, each
    // with its own device shard and streams, so that one cohort's env stepping and upload (host, PCIe) overlap the
    // other's batched forward (GPU): with ONE cohort the tick is host phase + device phase, and each idles in turn.
    int split = 2;
    if (const char* e = std::getenv("RELA_COHORT_SPLIT")) split = std::max(1, std::atoi(e));
    for (auto& b : buckets) {
      if (b.size() < 2) continue;
      const size_t parts = std::max<size_t>(1, std::min<size_t>((size_t)split, b.size() / 8));
      for (size_t p = 0; p < parts; ++p) {
        const size_t lo = b.size() * p / parts, hi = b.size() * (p + 1) / parts;
        auto cohort = make(*b[lo], (int)(hi - lo));
        for (size_t i = lo; i < hi; ++i) b[i]->joinCohort(cohort, (int)(i - lo));
      }
    }
  }

  // one replay partition per ModelLocker that feeds the replay (ReplayParts): announce them before any actor acts
  template <class ActorT>
  void planPartitionsOf() {
    std::vector<std::pair<decltype(std::declval<ActorT>().replay()), std::vector<const void*>>> seen;
    for (auto& l : loops_) {
      auto a = std::dynamic_pointer_cast<ActorT>(l->actor());
      if (!a || !a->trainable()) continue;
      auto rep = a->replay();
      auto it = std::find_if(seen.begin(), seen.end(), [&](auto& e) { return e.first.get() == rep.get(); });
      if (it == seen.end()) {
        seen.push_back({rep, {}});
        it = seen.end() - 1;
      }
      if (std::find(it->second.begin(), it->second.end(), a->lockerKey()) == it->second.end())
        it->second.push_back(a->lockerKey());
    }
    for (auto& e : seen) e.first->core().plan(e.second);
  }

  void formCohorts() {
    planPartitionsOf<DQNActor>();
    planPartitionsOf<R2D2Actor>();
    if (const char* e = std::getenv("RELA_COHORT_SPLIT"))
      if (std::atoi(e) <= 0) return;  // RELA_COHORT_SPLIT=0: no cohorts, one private shard per actor thread
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
  // the module's users run an independent sampler / learner thread next to the actor cohorts: keep a few CUs out of the
  // persistent forward kernels' grids (include/rela_amd.h: rela_runtime_set_cu_reserve; RELA_CU_RESERVE overrides)
  (void)rela_runtime_set_cu_reserve(8);

  m.def("set_replay_chunk_bytes", [](int64_t bytes) {
          if (rela_runtime_set_replay_chunk_bytes(bytes) != 0) throw std::runtime_error(rela_last_error());
        }, py::arg("bytes"),
        "field arrays of replays created AFTER this call that are larger than `bytes` are allocated as chunks of that size, "
        "so that a partition of any size can be exported to a learner process (include/rela_amd.h: "
        "rela_replay_set_chunk_bytes); 0 = off; not in the reference");
  m.def("threaded_stats", [](bool reset) { return gStats.snapshot(reset); }, py::arg("reset") = false,
        "RELA_THREADED_STATS=1: where the actor threads' wall time went since the last reset (diagnostic)");

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
      .def("last_sample_raw", &FFPrioritizedReplay::lastSampleRaw_)  // partition exchange only
      .def("export_ipc", &FFPrioritizedReplay::exportIpc)            // native exchange: this partition's IPC descriptor
      .def("export_chunks", &FFPrioritizedReplay::exportChunks)      // ... of a partition with chunked fields (any size)
      .def("sample_ids", &FFPrioritizedReplay::sampleIds);           // ... and a sample that leaves the rows in place

  py::class_<RNNPrioritizedReplay, std::shared_ptr<RNNPrioritizedReplay>>(m, "RNNPrioritizedReplay")
      .def(py::init<int, int, float, float, int>())
      .def("size", &RNNPrioritizedReplay::size)
      .def("num_add", &RNNPrioritizedReplay::numAdd)
      .def("sample", &RNNPrioritizedReplay::sample)
      .def("update_priority", &RNNPrioritizedReplay::updatePriority)
      .def("last_sample_raw", &RNNPrioritizedReplay::lastSampleRaw_)  // partition exchange only
      .def("export_ipc", &RNNPrioritizedReplay::exportIpc)
      .def("export_chunks", &RNNPrioritizedReplay::exportChunks)
      .def("sample_ids", &RNNPrioritizedReplay::sampleIds);

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
