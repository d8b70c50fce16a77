// common.h -- shared host helpers for librela_amd.so (error plumbing, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

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

}  // namespace rela_amd
