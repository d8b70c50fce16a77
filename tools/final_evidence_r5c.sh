#!/bin/bash
# Round-5 evidence, part C: the R2D2 actor tick under PMC in the f32x3 mode (the x part of the gate GEMM on the three-part
# kernel) and the R2D2 line once more (its roofline now prices that kernel); replaces part B's r2d2 records.
O=gpurun_out/r5_final; mkdir -p $O
R=$PWD
rm -rf $O/pmc_r2d2
for set in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  name=${set%%:*}; counters=${set#*:}
  (cd /tmp && export TMPDIR=/tmp && env ROWS=3200 TICKS=12 PRECISION=f32x3 timeout -k 10 300 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $R/$O/pmc_r2d2/pmc_$name -- python3 $R/tools/time_r2d2_tick.py > $R/$O/pmc_r2d2_$name.log 2>&1); echo "pmc r2d2 $name rc=$?"
done
python tools/pmc_table.py $O/pmc_r2d2 $O/traffic_r2d2.json > $O/pmc_table_r2d2.md 2>&1; cat $O/pmc_table_r2d2.md | cut -c1-200
find $O -name "*kernel_trace.csv" -size +3M -delete; find $O -name "*.db" -delete 2>/dev/null
python bench.py --algo r2d2 --steps 60 --warmup 10 --repeats 3 > $O/bench_r2d2.json 2> /dev/null; echo "bench r2d2 rc=$?"; tail -c 1700 $O/bench_r2d2.json
cp gpurun_out/bench_detail_r2d2_n1.json $O/bench_r2d2_detail.json 2>/dev/null
