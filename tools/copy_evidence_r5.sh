#!/bin/bash
# gpurun_out/r5_final (tools/final_evidence_r5a.sh + r5b.sh) -> profiles/r05_* (tracked).  Traffic: the f32x3, f32, fast
# and R2D2 PMC records merged into ONE profiles/r05_traffic.json (bench.py looks `label` / `label_f32x3` up there).
O=gpurun_out/r5_final; P=profiles
cp $O/gpu_tests.log $P/r05_gpu_tests.log
tail -n 1 $O/bench.json > $P/r05_bench_line.json
cp $O/bench_detail.json $P/r05_bench_detail.json
tail -n 1 $O/bench_r2d2.json > $P/r05_bench_r2d2_line.json
cp $O/bench_r2d2_detail.json $P/r05_bench_r2d2_detail.json
cp $O/bench_kernel_stats.csv $P/r05_bench_kernel_stats.csv
cp $O/bench_kernel_per_shape.csv $P/r05_bench_kernel_per_shape.csv
tail -n 1 $O/bench_under_rocprof.json > $P/r05_bench_under_rocprof.json
for tag in f32x3 f32 fast r2d2; do
  [ -f $O/pmc_table_$tag.md ] && cp $O/pmc_table_$tag.md $P/r05_pmc_table_$tag.md
  mkdir -p $P/r05_pmc_$tag
  for set in fetch write sq; do
    f=$(ls $O/pmc_$tag/pmc_$set/*/*_counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && cp $f $P/r05_pmc_$tag/pmc_${set}_counter_collection.csv
  done
done
python3 - <<PY
import json, os
out = {"source": "rocprofv3 --pmc passes (FETCH_SIZE x2 / WRITE_SIZE in KB, one counter set per pass, --kernel-trace only) of "
                 "tools/profile_forward.py at N = 6400 in the THREE precision modes (f32x3 records carry the suffix _f32x3) and of "
                 "tools/time_r2d2_tick.py at 3200 rows (tools/final_evidence_r5b.sh, tools/pmc_table.py)", "kernels": {}}
for tag in ("r2d2", "fast", "f32", "f32x3"):
    p = "$O/traffic_%s.json" % tag
    if not os.path.exists(p):
        continue
    for k, v in json.load(open(p))["kernels"].items():
        if tag == "f32" and k in ("conv3_mfma", "fc_mfma"):
            k += "_f32"  # (the bare labels belong to the fast mode's kernels, as in r02 / r03)
        if tag == "f32x3" and not k.endswith("_f32x3") and k in out["kernels"]:
            continue
        out["kernels"][k] = v
json.dump(out, open("$P/r05_traffic.json", "w"), indent=1)
print(sorted(out["kernels"]))
PY
cp $O/threaded_protocol_sliding_f32x3.log $P/r05_threaded_protocol_sliding_f32x3.log 2>/dev/null
ls -la $P | grep r05_ | wc -l
