#!/bin/bash
# Round-5 evidence, part B (one gpurun call): PMC passes (one counter set per pass, --kernel-trace only) of the isolated
# forward in BOTH precision modes + the replay loop, and of the R2D2 actor tick; the threaded benchmark's full
# 6 x 30 s protocol (pyrela/benchmark.py:73-109) for the sliding-stack env.
O=gpurun_out/r5_final; mkdir -p $O
R=$PWD
pmc() {  # tag, env..., -- script
  tag=$1; shift
  for set in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
    name=${set%%:*}; counters=${set#*:}
    (cd /tmp && export TMPDIR=/tmp && env "$@" timeout -k 10 300 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $R/$O/pmc_$tag/pmc_$name -- python3 $R/$SCRIPT > $R/$O/pmc_${tag}_$name.log 2>&1); echo "pmc $tag $name rc=$?"
  done
}
SCRIPT=tools/profile_forward.py pmc f32x3 PRECISION=f32x3 ITERS=8
SCRIPT=tools/profile_forward.py pmc f32 PRECISION=f32 ITERS=8
SCRIPT=tools/profile_forward.py pmc fast PRECISION=bf16x2 ITERS=8
SCRIPT=tools/time_r2d2_tick.py pmc r2d2 ROWS=3200 TICKS=12 PRECISION=f32x3
for tag in f32x3 f32 fast r2d2; do python tools/pmc_table.py $O/pmc_$tag $O/traffic_$tag.json > $O/pmc_table_$tag.md 2>&1; echo "table $tag rc=$?"; cat $O/pmc_table_$tag.md | cut -c1-200; done
find $O -name "*kernel_trace.csv" -size +3M -delete; find $O -name "*.db" -delete 2>/dev/null
RELA_PRECISION=f32x3 RELA_THREADED_STATS=1 python rela_amd/pyrela/benchmark.py --grid 64x100 --epoch_sec 30 --num_epoch 6 --replay_buffer_size 4194304 --burn_in_frames 20000 --env sliding > $O/threaded_protocol_sliding_f32x3.log 2>&1; echo "threaded protocol rc=$?"
grep -E "act rate|sample: epoch" $O/threaded_protocol_sliding_f32x3.log | cut -c1-160
