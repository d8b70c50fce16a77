"""Names under which the launch census (rela_prof_count_enable) reports the conv1 -> conv2 kernel of the split-bf16
mode: conv1 on the int8 matrix cores fused with conv2 through LDS (csrc/ffnet.hip: conv12_i8; the half-frame bf16
generations were removed in r4)."""
CONV12 = "conv12_i8"
CONV12_JOBS = "conv12_i8_jobs"
