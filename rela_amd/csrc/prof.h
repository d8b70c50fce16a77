// prof.h -- optional per-kernel timing with HIP events on the launch stream (bench.py's live
// roofline measurement).  Disabled by default: a ProfScope is then two relaxed atomic loads.
#pragma once
#include <hip/hip_runtime.h>

namespace rela_amd {

struct ProfScope {
  ProfScope(const char* name, hipStream_t stream);
  ~ProfScope();
  int slot;
  hipStream_t stream;
};

}  // namespace rela_amd
