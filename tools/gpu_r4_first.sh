#!/bin/bash
# round 4, first GPU pass: new parity tests, the restructured bench line, and where the threaded path's host time goes
set -o pipefail
mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q -k "k128 or cohort_in_fast or reuses" > gpurun_out/r4a/new_tests.log 2>&1; echo "new tests rc=$?" 
tail -3 gpurun_out/r4a/new_tests.log
python bench.py --steps 60 --warmup 10 --repeats 3 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; echo "bench rc=$?"
tail -c 2000 gpurun_out/r4a/bench.json
RELA_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 20 --warmup 5 --repeats 2 --replay-cap 262144 > gpurun_out/r4a/bench_2rank.json 2> gpurun_out/r4a/bench_2rank.err; echo "bench 2-rank self-spawn rc=$?"
tail -c 600 gpurun_out/r4a/bench_2rank.json
for env in fresh sliding null; do
  RELA_THREADED_STATS=1 python rela_amd/pyrela/benchmark.py --env $env --grid 64x100 --epoch_sec 3 --num_epoch 2 --replay_buffer_size 2097152 --burn_in_frames 20000 > gpurun_out/r4a/threaded_$env.log 2>&1; echo "threaded $env rc=$?"
  grep -E "act rate:|threaded stats" gpurun_out/r4a/threaded_$env.log | tail -3
done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; ls /sys/class/drm/ 2>/dev/null | head; ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>/dev/null | head -40
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/gpu_tests.log 2>&1; echo "all gpu tests rc=$?"; tail -3 gpurun_out/r4a/gpu_tests.log
