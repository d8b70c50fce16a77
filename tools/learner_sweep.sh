#!/bin/bash
# learner-only step time for several values of one environment knob:  tools/learner_sweep.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  env $var=$v RELA_BENCH_ONLY=learner python bench.py --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-threaded > gpurun_out/sweep_${var}_${v}.json 2> gpurun_out/sweep_${var}_${v}.err
  python - "$var" "$v" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sweep_%s_%s.json"%(sys.argv[1],sys.argv[2])))
k=d["kernels_ms_per_step"]
print(sys.argv[1],sys.argv[2],"ms/step %.4f"%d["ms_per_step"], {x:round(k[x]*1e3,1) for x in k if x.startswith("learner_wgrad") or x.startswith("learner_dgrad")})
PY
done
