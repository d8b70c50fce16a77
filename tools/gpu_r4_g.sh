#!/bin/bash
mkdir -p gpurun_out/r4g
python -X faulthandler -m pytest tests/test_replay_gpu.py -v -x -k "grouped or partitions or full_size or scan_running or protocol" > gpurun_out/r4g/replay_tail.log 2>&1; echo "replay tail rc=$?"
grep -E "PASSED|FAILED|Aborted|Error" gpurun_out/r4g/replay_tail.log | tail -15
python tools/ipc_probe.py > gpurun_out/r4g/ipc_probe.log 2>&1; echo "probe rc=$?"; tail -2 gpurun_out/r4g/ipc_probe.log | cut -c1-600
python -m pytest tests/test_ipc_gpu.py -x -q > gpurun_out/r4g/ipc_tests.log 2>&1; echo "ipc tests rc=$?"; tail -15 gpurun_out/r4g/ipc_tests.log | cut -c1-300
python -m pytest tests/test_replay_seq_gpu.py tests/test_replay_gpu.py -x -q > gpurun_out/r4g/replay_all.log 2>&1; echo "replay all rc=$?"; tail -3 gpurun_out/r4g/replay_all.log | cut -c1-200
