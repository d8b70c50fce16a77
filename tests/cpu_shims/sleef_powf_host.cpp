// CPU shim: the host build of rela_amd/csrc/sleef_powf_core.h (the same source the device compiles) plus the
// scalar tail rule of ATen's CPU pow kernel, for tests/test_sleef_powf_host.py.  TEST INFRASTRUCTURE.
#include <math.h>

#include "../../rela_amd/csrc/sleef_powf_core.h"

extern "C" void shim_aten_pow(const float* x, int n, float ex, float* out) {
  const int nv = n - (n & 31);
  for (int i = 0; i < n; ++i) {
    if (ex == 1.0f) out[i] = x[i];
    else if (ex == -1.0f) out[i] = 1.0f / x[i];
    else if (i < nv) out[i] = rela_amd::sleef::powf_u10(x[i], ex);
    else out[i] = (float)pow((double)x[i], (double)ex);
  }
}
