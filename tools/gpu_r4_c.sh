#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4c
python -m pytest tests/test_e2e_gpu.py tests/test_replay_gpu.py tests/test_replay_seq_gpu.py tests/test_dedup_gpu.py tests/test_agent_ops_gpu.py tests/test_r2d2_actor_gpu.py -x -q -m gpu > gpurun_out/r4c/tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r4c/tests.log
timeout -k 10 300 python tools/null_collapse.py > gpurun_out/r4c/null_collapse.log 2>&1; echo "null_collapse rc=$?"; tail -4 gpurun_out/r4c/null_collapse.log | cut -c1-900
run() { name=$1; shift; env RELA_THREADED_STATS=1 "$@" > gpurun_out/r4c/$name.log 2>&1; echo "$name rc=$?"; grep -E "act rate:|threaded stats" gpurun_out/r4c/$name.log | tail -2; }
B="python rela_amd/pyrela/benchmark.py --grid 64x100 --epoch_sec 1.5 --num_epoch 3 --replay_buffer_size 4194304 --burn_in_frames 20000"
run fresh    $B --env fresh
run sliding  $B --env sliding
run sliding_noplane RELA_PLANE_UPLOAD=0 $B --env sliding
run null     $B --env null
run sliding_fast RELA_PRECISION=bf16x2 $B --env sliding
run fresh_fast RELA_PRECISION=bf16x2 $B --env fresh
run sliding_32x200 python rela_amd/pyrela/benchmark.py --grid 32x200 --epoch_sec 1.5 --num_epoch 3 --replay_buffer_size 4194304 --burn_in_frames 20000 --env sliding
