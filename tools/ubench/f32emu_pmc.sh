#!/bin/bash
# PMC passes over the f32emu probe (one counter set per pass; --kernel-trace only, program directly after --).
BIN=${1:-tools/ubench/f32emu_probe}; TAG=${2:-occ1}
R=$PWD; O=$R/gpurun_out/emu/pmc_$TAG; mkdir -p $O
for set in "l2:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "fetch:FETCH_SIZE" "act:GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "ins:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_CVT SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "wait:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_VMEM_RD" "fifo:SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU"; do
  name=${set%%:*}; counters=${set#*:}
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $O/$name -- $R/$BIN 6554 2 > $O/$name.log 2>&1); echo "pmc $name rc=$?"
done
python3 - <<PY
import csv, glob, collections
for name in ("l2","fetch","act","ins","wait","fifo"):
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
