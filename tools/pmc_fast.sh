#!/bin/bash
# PMC pass over the isolated forward (fast mode): SQ counters per kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_fast
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PRECISION=${PRECISION:-bf16x2} N=6400 ITERS=5 timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/time_forward.py > $O/sq.log 2>&1
PRECISION=${PRECISION:-bf16x2} N=6400 ITERS=5 timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/sq2 -- python3 $R/tools/time_forward.py > $O/sq2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections, os
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "pmc_fast")
for sub in ("sq","sq2"):
    files=glob.glob(O+"/%s/**/*counter_collection.csv"%sub, recursive=True)
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            name=row["Kernel_Name"]
            short=None
            for k in ("conv1_persist","conv1_bf16x3","conv_bf16s","fc_bf16s","conv_mfma_bstat","conv_mfma","gemm_mfma"):
                if k in name: short=k+("<Conv3>" if "ConvFastCfg<64" in name else ""); break
            if short is None: continue
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in acc.items():
        print(sub,k,{c:round(sum(x)/len(x)) for c,x in v.items()})
PY
