#!/bin/bash
# Round-5 evidence, part A (one gpurun call): full GPU tests, the default bench line, the R2D2 line, the bench under
# rocprofv3 --kernel-trace --stats (summary + per-shape table).  Everything lands in gpurun_out/r5_final/.
O=gpurun_out/r5_final; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
python bench.py > $O/bench.json 2> $O/bench_detail.stderr; echo "bench rc=$?"; tail -c 1900 $O/bench.json
cp gpurun_out/bench_detail_apex_n1.json $O/bench_detail.json 2>/dev/null
python bench.py --algo r2d2 --steps 60 --warmup 10 --repeats 3 > $O/bench_r2d2.json 2> /dev/null; echo "bench r2d2 rc=$?"; tail -c 1500 $O/bench_r2d2.json
cp gpurun_out/bench_detail_r2d2_n1.json $O/bench_r2d2_detail.json 2>/dev/null
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- python3 $R/bench.py --steps 60 --warmup 5 --repeats 2 --no-cpu-baseline --no-threaded > $R/$O/bench_under_rocprof.json 2> /dev/null); echo "rocprof bench rc=$?"
python tools/per_shape_stats.py $O/prof_bench $O/bench_kernel_per_shape.csv; echo "per-shape rc=$?"
f=$(ls $O/prof_bench/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
head -12 $O/bench_kernel_stats.csv | cut -c1-160
find $O/prof_bench -name "*.csv" -size +3M -delete; find $O/prof_bench -name "*.db" -delete 2>/dev/null
