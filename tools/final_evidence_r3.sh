#!/bin/bash
# Round-3 evidence pass on the GPU box.  Outputs under gpurun_out/final3/ (tools/copy_evidence_r3.sh copies them to
# profiles/r03_*).  Each step is bounded by its own timeout and the script stops at the first failure.
#   pmc_{fetch,write,sq}/   one rocprofv3 --pmc pass per counter set over tools/profile_forward.py (bf16x2, N = 6400)
#   pmc_table.md, traffic.json   tools/pmc_table.py over those passes (HBM bytes per launch -> roofline.traffic)
#   bench.json              default `python bench.py` (threaded leg + cpu_baseline included)
#   prof_bench/             rocprofv3 --kernel-trace --stats of `bench.py --steps 60 --warmup 5 --repeats 2`
#   bench_r2d2.json         `python bench.py --algo r2d2`
#   bench_only_{learner,actor}.json, forward_modes.log, time_sample.json, r2d2_learner.log
#   bench_layout_reference_rehearsal.json   `--gpus 2 --layout reference`, both ranks on the one card over gloo
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/profile_forward.py > $O/pmc_fetch.log 2>&1 || exit 5
echo "pmc fetch done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/profile_forward.py > $O/pmc_write.log 2>&1 || exit 6
echo "pmc write done"
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/tools/profile_forward.py > $O/pmc_sq.log 2>&1 || exit 7
echo "pmc sq done"
cd $R
python tools/pmc_table.py $O $O/traffic.json > $O/pmc_table.md || exit 8
cat $O/pmc_table.md
cp $O/traffic.json $R/profiles/r03_traffic.json
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 2; }
cut -c1-400 $O/bench.json
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 60 --warmup 5 --repeats 2 --no-cpu-baseline --no-threaded > $O/bench_prof.json 2> $O/bench_prof.err || exit 3
echo "bench prof done"
cd $R
timeout -k 10 400 python bench.py --algo r2d2 --no-cpu-baseline > $O/bench_r2d2.json 2> $O/bench_r2d2.err || { tail -5 $O/bench_r2d2.err; exit 12; }
cut -c1-300 $O/bench_r2d2.json
RELA_BENCH_ONLY=learner timeout -k 10 200 python bench.py --no-cpu-baseline --no-threaded > $O/bench_only_learner.json 2> /dev/null
RELA_BENCH_ONLY=actor timeout -k 10 200 python bench.py --no-cpu-baseline --no-threaded > $O/bench_only_actor.json 2> /dev/null
{
  for n in 512 6400; do TAG="N=$n f32" N=$n PRECISION=f32 timeout -k 10 120 python tools/time_forward.py 2>&1 | tail -1; done
  for n in 512 1024 6400; do TAG="N=$n bf16x2" N=$n PRECISION=bf16x2 timeout -k 10 120 python tools/time_forward.py 2>&1 | tail -1; done
  TAG="N=6400 bf16x2, conv1 on bf16 MFMA in half frames (RELA_CONV12=bf16: the kernel up to mid r3)" RELA_CONV12=bf16 N=6400 PRECISION=bf16x2 timeout -k 10 120 python tools/time_forward.py 2>&1 | tail -1
  TAG="N=6400 bf16x2 (again)" N=6400 PRECISION=bf16x2 timeout -k 10 120 python tools/time_forward.py 2>&1 | tail -1
} > $O/forward_modes.log
{ echo "== tools/conv12_phases.py (conv12_i8)"; timeout -k 10 100 python tools/conv12_phases.py 2>/dev/null; echo "== RELA_CONV12=bf16 tools/conv12_phases.py (conv12_bf16s)"; RELA_CONV12=bf16 timeout -k 10 100 python tools/conv12_phases.py 2>/dev/null; echo "== tools/conv3_phases.py"; timeout -k 10 100 python tools/conv3_phases.py 2>/dev/null; echo "== tools/fc_phases.py"; timeout -k 10 100 python tools/fc_phases.py 2>/dev/null; } > $O/phase_stamps.log
cat $O/forward_modes.log
timeout -k 10 300 python tools/time_sample.py > $O/time_sample.json 2> $O/time_sample.err || exit 14
{ TAG="bf16x2 (bench default), persistent" PRECISION=bf16x2 timeout -k 10 200 python tools/time_r2d2_learner.py 2>&1 | tail -1; TAG="bf16x2, online trunk f32 (r2)" RELA_R2D2_ONLINE_F32=1 PRECISION=bf16x2 timeout -k 10 200 python tools/time_r2d2_learner.py 2>&1 | tail -1; TAG="f32, persistent" timeout -k 10 200 python tools/time_r2d2_learner.py 2>&1 | tail -1; } > $O/r2d2_learner.log
cut -c1-300 $O/r2d2_learner.log
RELA_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --layout reference --steps 40 --warmup 20 --repeats 3 > $O/bench_layout_reference_rehearsal.json 2> $O/bench_layout_reference_rehearsal.err || { tail -5 $O/bench_layout_reference_rehearsal.err; exit 15; }
cut -c1-300 $O/bench_layout_reference_rehearsal.json
RELA_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 100 --warmup 5 --repeats 3 --replay-cap 262144 --no-cpu-baseline > $O/bench_rehearsal_2ranks.json 2> $O/bench_rehearsal_2ranks.err || { tail -5 $O/bench_rehearsal_2ranks.err; exit 16; }
cut -c1-300 $O/bench_rehearsal_2ranks.json
python3 tools/per_shape_stats.py $O/prof_bench $O/bench_kernel_per_shape.csv || exit 17  # (before the large traces go)
find $O -name "*.csv" -size +8M -delete
echo "evidence pass complete"
