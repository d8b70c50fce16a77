#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_replay_gpu.py tests/test_replay_seq_gpu.py tests/test_e2e_gpu.py -q > $O/replay_tests.log 2>&1; echo "replay tests rc=$?"; tail -30 $O/replay_tests.log
timeout -k 10 120 python tools/time_sample.py > $O/time_sample.json 2>&1; cat $O/time_sample.json | tail -2
