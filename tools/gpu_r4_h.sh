#!/bin/bash
mkdir -p gpurun_out/r4h
python -m pytest tests/test_ipc_gpu.py -x -q > gpurun_out/r4h/ipc_tests.log 2>&1; echo "ipc tests rc=$?"; tail -3 gpurun_out/r4h/ipc_tests.log | cut -c1-300
for ex in native packed; do
RELA_BENCH_REHEARSAL=1 python bench.py --gpus 2 --layout reference --exchange $ex --steps 40 --warmup 20 --repeats 3 --replay-cap 262144 > gpurun_out/r4h/bench_ref_$ex.json 2> gpurun_out/r4h/bench_ref_$ex.err; echo "bench ref layout $ex rc=$?"; tail -1 gpurun_out/r4h/bench_ref_$ex.json | cut -c1-900; tail -3 gpurun_out/r4h/bench_ref_$ex.err | cut -c1-300
done
