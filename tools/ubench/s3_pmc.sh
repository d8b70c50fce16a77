#!/bin/bash
# PMC passes over the s3 probe (one counter set per pass; --kernel-trace only, program directly after --).
#   tools/ubench/s3_pmc.sh [BIN] [TAG] [N]        BIN = "py": the isolated forward of tools/time_forward.py instead
#   (PRECISION / N / ITERS from the environment)
BIN=${1:-tools/ubench/s3_probe}; TAG=${2:-v}; N=${3:-6554}
if [ "$BIN" = py ]; then CMD="python3 $PWD/tools/time_forward.py"; export N ITERS=${ITERS:-3}; else CMD="$PWD/$BIN $N 2"; fi
R=$PWD; O=$R/gpurun_out/s3/pmc_$TAG; mkdir -p $O
for set in "act:GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "ins:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "wait:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_VMEM_RD" "fifo:SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "l2:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${set%%:*}; counters=${set#*:}
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $O/$name -- $CMD > $O/$name.log 2>&1); echo "pmc $name rc=$?"
done
python3 - <<PY > $R/gpurun_out/s3/pmc_$TAG.txt
import csv, glob, collections, re
for name in ("act","ins","wait","fifo","l2","fetch","write"):
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % name, recursive=True)
    if not fs: print(name, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "gemm_s3" not in k and "_img_" not in k and "conv12" not in k: continue
        short = re.sub(r"rela_amd::s3::|<|>|\(.*", "", k)[:48]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(name, k, {c: round(sum(v)/len(v)) for c, v in acc[k].items()})
PY
find $O -name "*.csv" -size +2M -delete
cat $R/gpurun_out/s3/pmc_$TAG.txt
