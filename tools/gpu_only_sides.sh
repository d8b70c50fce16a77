#!/bin/bash
mkdir -p gpurun_out/r4k
for prec in f32 bf16x2; do for side in learner actor; do
RELA_BENCH_ONLY=$side python bench.py --precision $prec --steps 200 --repeats 3 --no-cpu-baseline --no-threaded > gpurun_out/r4k/only_${side}_$prec.json 2>/dev/null
cp gpurun_out/bench_detail_apex_n1.json gpurun_out/r4k/detail_${side}_$prec.json
python - <<PY
import json
d=json.load(open("gpurun_out/r4k/detail_${side}_$prec.json"))
k=d["kernels_ms_per_step"]
print("$side $prec ms/step", round(d["ms_per_step"],4), "sum kernels", round(sum(k.values()),3))
print("  ", ", ".join("%s %.3f"%(n,v) for n,v in sorted(k.items(), key=lambda kv:-kv[1])[:22]))
PY
done; done
