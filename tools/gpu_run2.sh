#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_r2d2_learner_gpu.py -q -x > $O/r2d2_learner.log 2>&1; echo "r2d2 learner rc=$?"; tail -30 $O/r2d2_learner.log
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_r2d2_learner_gpu.py > $O/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -25 $O/gpu_tests.log
