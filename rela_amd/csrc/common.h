// common.h -- shared host helpers for librela_amd.so (error plumbing, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/rela_amd.h"

namespace rela_amd {

void set_last_error(const char* fmt, ...);

#define RELA_HIP(call)                                                                      \
  do {                                                                                      \
    hipError_t _e = (call);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      ::rela_amd::set_last_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e),     \
                                 __FILE__, __LINE__);                                       \
      return (_e == hipErrorOutOfMemory) ? RELA_ENOMEM : RELA_ENODEV;                       \
    }                                                                                       \
  } while (0)

#define RELA_CHECK(cond, code, ...)            \
  do {                                         \
    if (!(cond)) {                             \
      ::rela_amd::set_last_error(__VA_ARGS__); \
      return (code);                           \
    }                                          \
  } while (0)

#define RELA_LAUNCH_CHECK() RELA_HIP(hipGetLastError())

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// RAII device selection for entry points called from arbitrary host threads
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};


// Page-locked staging for the small host->device uploads of the hot path (index plans, RNG draws,
// host-side priorities).  hipMemcpyAsync from pageable memory is only lifetime-safe because the
// runtime happens to stage it synchronously, and that staging drains the stream from the host; here
// the source is copied into a pinned arena first, so the upload is truly asynchronous and the caller's
// buffer may die at once.  The arena is a ring of kSeg segments: one segment per API call
// (begin ... end), reused only after the event recorded at its end() has completed.
struct HostStage {
  static constexpr int kSeg = 4;
  uint8_t* base = nullptr;
  size_t seg_bytes = 0, off = 0;
  hipEvent_t ev[kSeg] = {};
  bool pending[kSeg] = {};
  int cur = 0;
  bool open = false;

  int init(size_t bytes_per_call) {
    seg_bytes = (bytes_per_call + 255) & ~(size_t)255;
    if (hipHostMalloc(reinterpret_cast<void**>(&base), seg_bytes * kSeg, hipHostMallocDefault) != hipSuccess) {
      base = nullptr;
      return RELA_ENOMEM;
    }
    for (int i = 0; i < kSeg; ++i)
      if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return RELA_ENODEV;
    return RELA_OK;
  }
  void destroy() {
    for (int i = 0; i < kSeg; ++i)
      if (ev[i]) (void)hipEventDestroy(ev[i]);
    if (base) (void)hipHostFree(base);
    base = nullptr;
  }
  void begin() {
    cur = (cur + 1) % kSeg;
    if (pending[cur]) {
      (void)hipEventSynchronize(ev[cur]);
      pending[cur] = false;
    }
    off = 0;
    open = true;
  }
  // upload `bytes` from src to dst_dev on stream s (asynchronous when the segment has room)
  hipError_t h2d(void* dst_dev, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    const size_t need = (bytes + 15) & ~(size_t)15;
    if (!open || base == nullptr || off + need > seg_bytes) {  // fallback: still lifetime-safe
      hipError_t e = hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, s);
      if (e != hipSuccess) return e;
      return hipStreamSynchronize(s);
    }
    uint8_t* p = base + (size_t)cur * seg_bytes + off;
    memcpy(p, src, bytes);
    off += need;
    return hipMemcpyAsync(dst_dev, p, bytes, hipMemcpyHostToDevice, s);
  }
  void end(hipStream_t s) {
    if (!open) return;
    if (off > 0 && hipEventRecord(ev[cur], s) == hipSuccess) pending[cur] = true;
    open = false;
  }
};

// Two small device-to-device copies as ONE kernel launch.  A tick copies a few KB four times (eps / legal moves into their
// history slots, rewards / terminals into theirs); as hipMemcpyAsync these are blit kernels of the runtime, and the kernel
// that follows one starts ~25 us late (profiles/r05_trace_gaps.txt: "after __amd_rocclr_copyBuffer before conv12_s3").
__global__ static void dev_copy2_kernel(uint8_t* __restrict__ d0, const uint8_t* __restrict__ s0, size_t n0,
                                        uint8_t* __restrict__ d1, const uint8_t* __restrict__ s1, size_t n1) {
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  uint8_t* d = blockIdx.y ? d1 : d0;
  const uint8_t* s = blockIdx.y ? s1 : s0;
  const size_t n = blockIdx.y ? n1 : n0;
  if ((((uintptr_t)d | (uintptr_t)s | n) & 15) == 0) {
    for (size_t i = i0; i < (n >> 4); i += stride) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
  } else if ((((uintptr_t)d | (uintptr_t)s | n) & 3) == 0) {
    for (size_t i = i0; i < (n >> 2); i += stride) reinterpret_cast<uint32_t*>(d)[i] = reinterpret_cast<const uint32_t*>(s)[i];
  } else {
    for (size_t i = i0; i < n; i += stride) d[i] = s[i];
  }
}
inline hipError_t dev_copy2(void* d0, const void* s0, size_t n0, void* d1, const void* s1, size_t n1, hipStream_t s) {
  const size_t m = n0 > n1 ? n0 : n1;
  const size_t want = (m / 16 + 255) / 256;  // one 16-byte unit per thread up to 2,048 blocks (a 6.5 MB LSTM state: 8 waves per CU)
  const int gx = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
  hipLaunchKernelGGL(dev_copy2_kernel, dim3(gx, 2), dim3(256), 0, s, (uint8_t*)d0, (const uint8_t*)s0, n0, (uint8_t*)d1,
                     (const uint8_t*)s1, n1);
  return hipGetLastError();
}

// agent_ops.hip: completes sliding frame stacks of an observation slot from the previous slot (see there)
int slide_stacks(uint8_t* cur_slot, const uint8_t* prev_slot, const uint8_t* fresh_planes, const uint8_t* restart_dev, int rows,
                 hipStream_t s);

}  // namespace rela_amd
