#!/bin/bash
# Round-end evidence pass on the GPU box (round 2).  Outputs under gpurun_out/final/ (copied to profiles/r02_* by hand):
#   gpu_tests.log            full `pytest -m gpu`
#   pmc_*/, pmc_table.md     one rocprofv3 --pmc pass per counter set over tools/profile_forward.py (bf16x2 mode)
#   traffic.json             HBM bytes per launch from those passes (-> profiles/r02_traffic.json)
#   iso_stats/               rocprofv3 --kernel-trace --stats of the same isolated script
#   bench.json               default `python bench.py` (with cpu_baseline; reads the traffic file)
#   prof_bench/              rocprofv3 --kernel-trace --stats of bench.py
#   bench_r2d2.json          `python bench.py --algo r2d2`
#   bench_dedup_2p23.json    `python bench.py --dedup plane --replay-cap 8388608` (BASELINE C5's replay on ONE GPU)
#   forward_modes.log        per-kernel forward timings: N = 512 / 6400, f32 and bf16x2, fused and unfused conv1 -> conv2
#   time_sample.json         tools/time_sample.py (isolated sample path, ring 1,310,720)
#   r2d2_learner.log         tools/time_r2d2_learner.py (isolated R2D2 learner step: bf16x2 / f32, persistent vs per-step launches)
#   bench_rehearsal_2ranks.json   `bench.py --gpus 2` with both ranks on the one card over gloo (N > 1 code path)
#   bench_only_{learner,actor}.json   one side of the step alone (diagnosis: RELA_BENCH_ONLY)
#   lds_conflicts.txt        tools/lds_conflicts.py (bank-conflict model of conv12_bf16s; compare SQ_LDS_BANK_CONFLICT)
#   threaded_benchmark*.log  rela_amd/pyrela/benchmark.py through the C++ actor threads
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1 || { tail -5 $O/gpu_tests.log; exit 1; }
  tail -1 $O/gpu_tests.log
fi
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
cp $O/traffic.json $R/profiles/r02_traffic.json
cat $O/pmc_table.md
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 2; }
cut -c1-600 $O/bench.json
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err || exit 3
echo "bench prof done"
cd $R
timeout -k 10 400 python bench.py --algo r2d2 --no-cpu-baseline > $O/bench_r2d2.json 2> $O/bench_r2d2.err || { tail -5 $O/bench_r2d2.err; exit 12; }
cut -c1-400 $O/bench_r2d2.json
timeout -k 10 600 python bench.py --dedup plane --replay-cap 8388608 --steps 300 --no-cpu-baseline > $O/bench_dedup_2p23.json 2> $O/bench_dedup_2p23.err || { tail -5 $O/bench_dedup_2p23.err; exit 13; }
cut -c1-400 $O/bench_dedup_2p23.json
{
  for n in 512 6400; do TAG="N=$n f32" N=$n PRECISION=f32 timeout -k 10 120 python tools/time_forward.py 2>&1 | tail -1; done
  for n in 1024 6400; do TAG="N=$n bf16x2 fused" N=$n PRECISION=bf16x2 timeout -k 10 120 python tools/time_forward.py 2>&1 | tail -1; done
  TAG="N=6400 bf16x2 separate conv1, conv2" RELA_FUSE12=0 N=6400 PRECISION=bf16x2 timeout -k 10 120 python tools/time_forward.py 2>&1 | tail -1
  TAG="N=6400 bf16x2 fused, layer-specialised waves" RELA_FUSE12=2 N=6400 PRECISION=bf16x2 timeout -k 10 120 python tools/time_forward.py 2>&1 | tail -1
} > $O/forward_modes.log
cat $O/forward_modes.log
timeout -k 10 300 python tools/time_sample.py > $O/time_sample.json 2> $O/time_sample.err || exit 14
tail -2 $O/time_sample.json | cut -c1-400
{ TAG="bf16x2 (bench default), persistent" PRECISION=bf16x2 timeout -k 10 200 python tools/time_r2d2_learner.py 2>&1 | tail -1; TAG="f32, persistent" timeout -k 10 200 python tools/time_r2d2_learner.py 2>&1 | tail -1; RELA_R2D2_REC=steps TAG="f32, per-step launches" timeout -k 10 200 python tools/time_r2d2_learner.py 2>&1 | tail -1; } > $O/r2d2_learner.log
cut -c1-300 $O/r2d2_learner.log
timeout -k 10 400 python rela_amd/pyrela/benchmark.py --grid 64x100 --epoch_sec 1.5 --num_epoch 4 --replay_buffer_size 2097152 --burn_in_frames 20000 > $O/threaded_benchmark.log 2>&1 || { tail -5 $O/threaded_benchmark.log; exit 9; }
tail -4 $O/threaded_benchmark.log
timeout -k 10 300 python rela_amd/pyrela/benchmark.py --algo r2d2 --grid 32x100 --epoch_sec 2 --num_epoch 4 --replay_buffer_size 8192 --burn_in_frames 200 --episode_len 400 > $O/threaded_benchmark_r2d2.log 2>&1 || { tail -5 $O/threaded_benchmark_r2d2.log; exit 11; }
tail -3 $O/threaded_benchmark_r2d2.log
RELA_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 200 --warmup 5 --replay-cap 262144 --no-cpu-baseline > $O/bench_rehearsal_2ranks.json 2> $O/bench_rehearsal_2ranks.err || { tail -5 $O/bench_rehearsal_2ranks.err; exit 15; }
cut -c1-300 $O/bench_rehearsal_2ranks.json
RELA_BENCH_ONLY=learner timeout -k 10 200 python bench.py --no-cpu-baseline --steps 300 > $O/bench_only_learner.json 2> /dev/null
RELA_BENCH_ONLY=actor timeout -k 10 200 python bench.py --no-cpu-baseline --steps 300 > $O/bench_only_actor.json 2> /dev/null
python tools/lds_conflicts.py > $O/lds_conflicts.txt
find $O -name "*.csv" -size +8M -delete
