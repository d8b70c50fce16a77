#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4e
for cfg in "1048576 65536 6400 weight" "1048576 65536 6400 ones" "262144 65536 6400 weight" "1048576 4096 4096 weight"; do set -- $cfg
  echo "== CAP=$1 ROWS=$2 ADD=$3 FEED=$4" >> gpurun_out/r4e/null2.log
  CAP=$1 ROWS=$2 ADD=$3 FEED=$4 ROUNDS=8 timeout -k 10 200 python tools/null_collapse2.py >> gpurun_out/r4e/null2.log 2>&1; echo "null2 $cfg rc=$?"
done
grep -E "^==|filled|round" gpurun_out/r4e/null2.log | cut -c1-330
python -m pytest tests -m gpu -x -q > gpurun_out/r4e/gpu_tests.log 2>&1; echo "all gpu tests rc=$?"; tail -3 gpurun_out/r4e/gpu_tests.log
python bench.py > gpurun_out/r4e/bench.json 2> gpurun_out/r4e/bench.err; echo "bench rc=$?"; tail -c 1900 gpurun_out/r4e/bench.json
