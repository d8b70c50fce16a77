#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_replay_gpu.py tests/test_e2e_gpu.py -q -k "partitions_match or multi_process" > $O/multi.log 2>&1; echo "pytest rc=$?"; tail -30 $O/multi.log
