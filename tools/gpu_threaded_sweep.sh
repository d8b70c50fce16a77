#!/bin/bash
# threaded drop-in (rela.Context + C++ actor threads + host envs + H2D) under several env / thread shapes
run() { echo "== $*"; env "$@" timeout -k 10 200 python rela_amd/pyrela/benchmark.py --grid $GRID --epoch_sec 2 --num_epoch 2 --replay_buffer_size 2097152 2>&1 | grep -E "act rate:|Error|error" | tail -2; }
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
GRID=64x100 run RELA_SYNTH_SLIDING=0
GRID=64x100 run RELA_SYNTH_SLIDING=1
GRID=16x400 run RELA_SYNTH_SLIDING=1
GRID=128x50 run RELA_SYNTH_SLIDING=1
GRID=64x100 run RELA_SYNTH_SLIDING=1 RELA_REPLAY_DEDUP=plane
