#!/bin/bash
# GPU pass: full gpu tests, default bench, rocprof of the bench, threaded benchmark.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -15 $O/gpu_tests.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; }
cut -c1-1500 $O/bench.json
timeout -k 10 400 python rela_amd/pyrela/benchmark.py --grid 64x100 --epoch_sec 1.5 --num_epoch 4 --replay_buffer_size 2097152 --burn_in_frames 20000 > $O/threaded_benchmark.log 2>&1 || { tail -5 $O/threaded_benchmark.log; }
tail -4 $O/threaded_benchmark.log
