#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -6 $O/gpu_tests.log
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02e/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','grad_steps_per_s','replay_sample_scan_ms')}, d['no_reuse']['env_steps_per_s'], d['roofline']['frac'], d['roofline_hbm'])
print({k:round(v,4) for k,v in d['kernels_ms_per_step'].items() if k.startswith('seq_') or k.startswith('replay_')})
PY
timeout -k 10 400 python rela_amd/pyrela/benchmark.py --grid 64x100 --epoch_sec 1.5 --num_epoch 3 --replay_buffer_size 2097152 --burn_in_frames 20000 > $O/threaded_benchmark.log 2>&1 || { tail -5 $O/threaded_benchmark.log; }
tail -9 $O/threaded_benchmark.log
