#!/bin/bash
# Round-end evidence pass on the GPU box.  Outputs under gpurun_out/final/ (copied to profiles/ by hand):
#   gpu_tests.log            full `pytest -m gpu`
#   pmc_*/, pmc_table.md     one rocprofv3 --pmc pass per counter set over tools/profile_forward.py
#   traffic.json             HBM bytes per launch from those passes (-> profiles/r01_traffic.json)
#   iso_stats/               rocprofv3 --kernel-trace --stats of the same isolated script
#   bench.json               default `python bench.py` (with cpu_baseline; reads the traffic file)
#   prof_bench/              rocprofv3 --kernel-trace --stats of bench.py
#   forward_small_n.log      per-kernel forward timings at N = 80 / 512 / 6400
#   threaded_benchmark.log   rela_amd/pyrela/benchmark.py, 64 threads x 100 envs, replay 2^21
#   threaded_benchmark_r2d2.log  the same driver with --algo r2d2, 32 threads x 100 envs
#   r2d2_actor_tick.json     tools/time_r2d2_tick.py, 3200 envs, seq 80 / burn 40 / n 3
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1 || { tail -5 $O/gpu_tests.log; exit 1; }
tail -1 $O/gpu_tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iso_stats -- python3 $R/tools/profile_forward.py > $O/iso_stats.log 2>&1 || exit 4
echo "iso stats done"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/profile_forward.py > $O/pmc_fetch.log 2>&1 || exit 5
echo "pmc fetch done"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/profile_forward.py > $O/pmc_write.log 2>&1 || exit 6
echo "pmc write done"
timeout -k 10 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/tools/profile_forward.py > $O/pmc_sq.log 2>&1 || exit 7
echo "pmc sq done"
cd $R
python tools/pmc_table.py $O $O/traffic.json > $O/pmc_table.md || exit 8
cp $O/traffic.json $R/profiles/r01_traffic.json
cat $O/pmc_table.md
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 2; }
cat $O/bench.json
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err || exit 3
echo "bench prof done"
cd $R
for n in 80 512 6400; do TAG=N=$n N=$n timeout -k 10 120 python tools/time_forward.py 2>&1 | grep conv2; done > $O/forward_small_n.log
cat $O/forward_small_n.log
timeout -k 10 400 python rela_amd/pyrela/benchmark.py --grid 64x100 --epoch_sec 1.5 --num_epoch 4 --replay_buffer_size 2097152 --burn_in_frames 20000 > $O/threaded_benchmark.log 2>&1 || { tail -5 $O/threaded_benchmark.log; exit 9; }
tail -4 $O/threaded_benchmark.log
timeout -k 10 300 python rela_amd/pyrela/benchmark.py --algo r2d2 --grid 32x100 --epoch_sec 2 --num_epoch 4 --replay_buffer_size 8192 --burn_in_frames 200 --episode_len 400 > $O/threaded_benchmark_r2d2.log 2>&1 || { tail -5 $O/threaded_benchmark_r2d2.log; exit 11; }
tail -3 $O/threaded_benchmark_r2d2.log
ROWS=3200 TICKS=260 timeout -k 10 300 python tools/time_r2d2_tick.py 2> $O/r2d2_actor_tick.err | tail -1 > $O/r2d2_actor_tick.json || exit 10
cut -c1-300 $O/r2d2_actor_tick.json
find $O -name "*.csv" -size +8M -delete
