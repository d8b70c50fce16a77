#!/bin/bash
mkdir -p gpurun_out/r4i
for cfg in "legacy 0" "new 0" "new 1" "legacy 0"; do set -- $cfg
RELA_DBG_INSERT=$1 RELA_DBG_COPY_PRIO=$2 python bench.py --precision bf16x2 --steps 150 --repeats 3 --no-threaded --no-cpu-baseline > gpurun_out/r4i/bench_$1_$2.json 2> /dev/null; echo "bench insert=$1 copyprio=$2 rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r4i/bench_$1_$2.json").read().strip().splitlines()[-1])
print("insert=$1 copyprio=$2", d["value"], d["summary"])
dd=json.load(open("gpurun_out/bench_detail_apex_n1.json"))
k=dd["kernels_ms_per_step"]
print({n: round(k[n],3) for n in ("replay_scatter_rows","learner_col2im","learner_wgrad_conv2","learner_colsum","seq_chain","conv12_fused","replay_update") if n in k})
PY
done
