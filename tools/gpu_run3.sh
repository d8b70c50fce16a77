#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
timeout -k 10 600 python bench.py --algo r2d2 --steps 200 --warmup 10 > $O/bench_r2d2.json 2> $O/bench_r2d2.err; echo "r2d2 bench rc=$?"; tail -5 $O/bench_r2d2.err; cut -c1-3000 $O/bench_r2d2.json
timeout -k 10 300 python -m pytest tests/test_e2e_gpu.py -q -k "benchmark_driver or training_entry" > $O/e2e_subset.log 2>&1; echo "pytest rc=$?"; tail -8 $O/e2e_subset.log
