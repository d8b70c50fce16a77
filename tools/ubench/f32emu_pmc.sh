#!/bin/bash
# PMC passes over the f32emu probe (one counter set per pass; --kernel-trace only, program directly after --).
BIN=${1:-tools/ubench/f32emu_probe_occ1}; TAG=${2:-occ1}
R=$PWD; O=$R/gpurun_out/emu/pmc_$TAG; mkdir -p $O
for set in "l2:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "fetch:FETCH_SIZE" "sq:GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "ta:TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  name=${set%%:*}; counters=${set#*:}
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $O/$name -- $R/$BIN 6554 2 > $O/$name.log 2>&1); echo "pmc $name rc=$?"
done
python3 - <<PY
import csv, glob, collections
for name in ("l2","fetch","sq","ta"):
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % name, recursive=True)
    if not fs: print(name, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "gemm_f32emu" not in k: continue
        x6 = (", 6," in k) or ("Li6E" in k)
        short = ("conv2" if "ProbConv2" in k else "conv3" if "ProbConv3" in k else "fc") + ("_x6" if x6 else "_x9")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(name, k, {c: round(sum(v)/len(v)) for c, v in acc[k].items()})
PY
