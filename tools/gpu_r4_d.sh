#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4d
for cfg in "0.0043 weight" "1.0 weight" "0.0043 ones" "0.37 weight"; do set -- $cfg
  echo "== PRIO=$1 FEED=$2" >> gpurun_out/r4d/null2.log
  PRIO=$1 FEED=$2 timeout -k 10 120 python tools/null_collapse2.py >> gpurun_out/r4d/null2.log 2>&1; echo "null2 $1 $2 rc=$?"
done
grep -E "^==|round\": (0|1|2|5|11)," gpurun_out/r4d/null2.log | cut -c1-400
run() { name=$1; shift; env RELA_THREADED_STATS=1 "$@" > gpurun_out/r4d/$name.log 2>&1; echo "$name rc=$?"; grep -E "act rate:" gpurun_out/r4d/$name.log | tail -1; }
B="--epoch_sec 1.5 --num_epoch 3 --replay_buffer_size 4194304 --burn_in_frames 20000"
run sliding_64x100 python rela_amd/pyrela/benchmark.py --grid 64x100 $B --env sliding
run sliding_32x200 python rela_amd/pyrela/benchmark.py --grid 32x200 $B --env sliding
run sliding_16x400 python rela_amd/pyrela/benchmark.py --grid 16x400 $B --env sliding
run fresh_32x200 python rela_amd/pyrela/benchmark.py --grid 32x200 $B --env fresh
run r2d2_sliding_32x100 python rela_amd/pyrela/benchmark.py --algo r2d2 --grid 32x100 --epoch_sec 1.5 --num_epoch 3 --replay_buffer_size 16384 --burn_in_frames 200 --env sliding
run r2d2_fresh_32x100 python rela_amd/pyrela/benchmark.py --algo r2d2 --grid 32x100 --epoch_sec 1.5 --num_epoch 3 --replay_buffer_size 16384 --burn_in_frames 200 --env fresh
grep -E "sample: epoch|thread-us" gpurun_out/r4d/sliding_64x100.log | cut -c1-330
grep -E "sample: epoch|thread-us" gpurun_out/r4d/sliding_32x200.log | cut -c1-330
