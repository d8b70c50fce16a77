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

// Launch census (tests): when enabled (rela_prof_count_enable) every launch site that calls note_launch adds one to
// the counter of the kernel it REALLY launched (the ProfScope labels are shared by the f32 and the split-bf16
// kernels of a layer); rela_prof_counts_json reads and clears.  Off: one relaxed atomic load.
void note_launch(const char* kernel);

}  // namespace rela_amd
