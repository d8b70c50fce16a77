#!/bin/bash
# round 4: the threaded drop-in path after the VectorEnv / plane-upload / cohort-split rework
set -o pipefail
mkdir -p gpurun_out/r4b
python -m pytest tests/test_e2e_gpu.py tests/test_rela_module_cpu.py -x -q -m "gpu or not gpu" > gpurun_out/r4b/e2e_tests.log 2>&1; echo "e2e tests rc=$?"
tail -4 gpurun_out/r4b/e2e_tests.log
run() {  # name, env vars..., then benchmark args
  name=$1; shift
  env RELA_THREADED_STATS=1 "$@" > gpurun_out/r4b/$name.log 2>&1; echo "$name rc=$?"
  grep -E "act rate:|threaded stats" gpurun_out/r4b/$name.log | tail -2
}
B="python rela_amd/pyrela/benchmark.py --grid 64x100 --epoch_sec 2 --num_epoch 2 --replay_buffer_size 4194304 --burn_in_frames 20000"
run fresh_split2            $B --env fresh
run fresh_split1            RELA_COHORT_SPLIT=1 $B --env fresh
run sliding_split2          $B --env sliding
run sliding_split2_noplane  RELA_PLANE_UPLOAD=0 $B --env sliding
run sliding_split4          RELA_COHORT_SPLIT=4 $B --env sliding
run null_split2             $B --env null
run null_split4             RELA_COHORT_SPLIT=4 $B --env null
run sliding_split2_fast     RELA_PRECISION=bf16x2 $B --env sliding
run null_split2_fast        RELA_PRECISION=bf16x2 $B --env null
run sliding_128x50          python rela_amd/pyrela/benchmark.py --grid 128x50 --epoch_sec 2 --num_epoch 2 --replay_buffer_size 4194304 --burn_in_frames 20000 --env sliding
