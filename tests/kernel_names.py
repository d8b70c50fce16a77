"""Names under which the launch census (rela_prof_count_enable) reports the conv1 -> conv2 kernel of the split-bf16
mode: conv1 on the int8 matrix cores (csrc/ffnet.hip: conv12_i8) unless RELA_CONV12=bf16 selects the half-frame bf16
kernel -- the library reads the variable once per process, and so do the tests."""
import os

_BF16 = os.environ.get("RELA_CONV12") == "bf16"
CONV12 = "conv12_bf16s" if _BF16 else "conv12_i8"
CONV12_JOBS = "conv12_bf16s_jobs" if _BF16 else "conv12_i8_jobs"
