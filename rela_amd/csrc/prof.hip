// prof.hip -- see prof.h.  Events are recorded on the same stream the kernel is launched on, so
// the measured interval is that kernel's execution (plus the inter-kernel gap before it drains).
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "prof.h"

namespace rela_amd {
namespace {
std::atomic<int> g_on{0};
std::mutex g_m;
struct Rec {
  const char* name;
  hipEvent_t a, b;
};
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
std::vector<std::string> g_filter;  // non-empty: only these kernel names are timed
std::atomic<int> g_filtered{0};

std::atomic<int> g_count_on{0};
std::map<std::string, long> g_counts;

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

ProfScope::ProfScope(const char* name, hipStream_t s) : slot(-1), stream(s) {
  if (!g_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_m);
  if (g_filtered.load(std::memory_order_relaxed)) {
    bool keep = false;
    for (const auto& f : g_filter) keep = keep || f == name;
    if (!keep) return;
  }
  Rec r{name, get_event(), get_event()};
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
  slot = (int)g_recs.size() - 1;
}

ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_m);
  if (slot < (int)g_recs.size()) (void)hipEventRecord(g_recs[slot].b, stream);
}

void note_launch(const char* kernel) {
  if (!g_count_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_m);
  g_counts[kernel] += 1;
}

}  // namespace rela_amd

using namespace rela_amd;

extern "C" int rela_prof_count_enable(int on) {
  std::lock_guard<std::mutex> lk(g_m);
  if (on && !g_count_on.load()) g_counts.clear();
  g_count_on.store(on ? 1 : 0);
  return RELA_OK;
}

// {"kernel": launches, ...} since rela_prof_count_enable(1) or the last call; clears the counters.
extern "C" int rela_prof_counts_json(char* out, int64_t cap) {
  RELA_CHECK(out && cap > 2, RELA_EINVAL, "rela_prof_counts_json: bad arguments");
  std::lock_guard<std::mutex> lk(g_m);
  std::string s = "{";
  bool first = true;
  for (auto& kv : g_counts) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s\"%s\":%ld", first ? "" : ",", kv.first.c_str(), kv.second);
    s += buf;
    first = false;
  }
  s += "}";
  g_counts.clear();
  RELA_CHECK((int64_t)s.size() + 1 <= cap, RELA_EINVAL, "rela_prof_counts_json: buffer too small");
  memcpy(out, s.c_str(), s.size() + 1);
  return RELA_OK;
}

extern "C" int rela_prof_enable(int on) {
  g_on.store(on ? 1 : 0);
  return RELA_OK;
}

// Restricts the timing to a comma-separated list of kernel names (NULL or "" = all kernels): two
// event records per kernel are not free on the launch stream (0.37 ms per bench step for ~100 kernels).
extern "C" int rela_prof_set_filter(const char* names) {
  std::lock_guard<std::mutex> lk(g_m);
  g_filter.clear();
  if (names) {
    std::string cur;
    for (const char* p = names;; ++p) {
      if (*p == ',' || *p == 0) {
        if (!cur.empty()) g_filter.push_back(cur);
        cur.clear();
        if (*p == 0) break;
      } else {
        cur += *p;
      }
    }
  }
  g_filtered.store(g_filter.empty() ? 0 : 1);
  return RELA_OK;
}

// Synchronises the device, aggregates and clears the recorded intervals.  Writes a JSON object
// {"kernel": {"count": n, "total_ms": t}, ...} into out (NUL terminated).
extern "C" int rela_prof_summary_json(char* out, int64_t cap) {
  RELA_CHECK(out && cap > 2, RELA_EINVAL, "rela_prof_summary_json: bad arguments");
  RELA_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(g_m);
  std::map<std::string, std::pair<long, double>> agg;
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      auto& e = agg[r.name];
      e.first += 1;
      e.second += ms;
    }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  std::string s = "{";
  bool first = true;
  for (auto& kv : agg) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s\"%s\":{\"count\":%ld,\"total_ms\":%.6f}", first ? "" : ",", kv.first.c_str(),
             kv.second.first, kv.second.second);
    s += buf;
    first = false;
  }
  s += "}";
  RELA_CHECK((int64_t)s.size() + 1 <= cap, RELA_EINVAL, "rela_prof_summary_json: buffer too small");
  memcpy(out, s.c_str(), s.size() + 1);
  return RELA_OK;
}

extern "C" int rela_stream_create(void** out, int device) {
  RELA_CHECK(out, RELA_EINVAL, "rela_stream_create: bad arguments");
  DeviceGuard g(device);
  RELA_CHECK(g.ok, RELA_ENODEV, "rela_stream_create: HIP device %d not available; there is no CPU path", device);
  hipStream_t s = nullptr;
  RELA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *out = s;
  return RELA_OK;
}

extern "C" void rela_stream_destroy(void* stream, int device) {
  if (!stream) return;
  DeviceGuard g(device);
  (void)hipStreamSynchronize((hipStream_t)stream);
  (void)hipStreamDestroy((hipStream_t)stream);
}

extern "C" int rela_stream_synchronize(void* stream, int device) {
  DeviceGuard g(device);
  RELA_HIP(hipStreamSynchronize((hipStream_t)stream));
  return RELA_OK;
}

extern "C" int rela_stream_wait_stream(void* waiter, void* signaler, int device) {
  DeviceGuard g(device);
  hipEvent_t ev = nullptr;
  RELA_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  RELA_HIP(hipEventRecord(ev, (hipStream_t)signaler));
  RELA_HIP(hipStreamWaitEvent((hipStream_t)waiter, ev, 0));
  RELA_HIP(hipEventDestroy(ev));  // released once the recorded work has completed
  return RELA_OK;
}

extern "C" int rela_memcpy_h2d_async(void* dst_dev, const void* src_host, int64_t bytes, void* stream, int device) {
  RELA_CHECK(dst_dev && src_host && bytes >= 0, RELA_EINVAL, "rela_memcpy_h2d_async: bad arguments");
  DeviceGuard g(device);
  RELA_HIP(hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return RELA_OK;
}
