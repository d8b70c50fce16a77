#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4f
python -m pytest tests -m gpu -x -q > gpurun_out/r4f/gpu_tests.log 2>&1; echo "all gpu tests rc=$?"; tail -3 gpurun_out/r4f/gpu_tests.log
run() { name=$1; shift; env RELA_THREADED_STATS=1 "$@" > gpurun_out/r4f/$name.log 2>&1; echo "$name rc=$?"; grep -E "act rate:" gpurun_out/r4f/$name.log | tail -1; }
B="--epoch_sec 1.5 --num_epoch 3 --replay_buffer_size 4194304 --burn_in_frames 20000"
run null_64x100 python rela_amd/pyrela/benchmark.py --grid 64x100 $B --env null
grep -E "sample: epoch" gpurun_out/r4f/null_64x100.log
for prio in 1 2; do
  RELA_BENCH_PRIO=$prio python bench.py --steps 100 --repeats 3 --no-threaded --no-cpu-baseline > gpurun_out/r4f/bench_prio$prio.json 2> /dev/null; echo "bench prio $prio rc=$?"
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4f/bench_prio$prio.json").read().strip().splitlines()[-1])
print("prio $prio", d["value"], d["summary"])
PY
done
python bench.py --algo r2d2 --steps 40 --warmup 10 --repeats 3 > gpurun_out/r4f/bench_r2d2.json 2> gpurun_out/r4f/bench_r2d2.err; echo "bench r2d2 rc=$?"; tail -c 1500 gpurun_out/r4f/bench_r2d2.json
