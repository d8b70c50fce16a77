#!/bin/bash
# Two quick PMC passes over the isolated forward (tools/profile_forward.py, no replay): HBM fetch bytes and LDS activity
# per kernel -> gpurun_out/pmcq_<tag>.txt.   tools/pmc_quick.sh TAG [VAR=value ...]
tag=$1; shift
for kv in "$@"; do export "$kv"; done
export WITH_REPLAY=0 ITERS=6
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcq_$tag
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/profile_forward.py > $O/fetch.log 2>&1 || exit 5
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/lds -- python3 $R/tools/profile_forward.py > $O/lds.log 2>&1 || exit 6
python3 - $O > $R/gpurun_out/pmcq_$tag.txt <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
for name in ("fetch", "lds"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(glob.glob("%s/%s/*/*_counter_collection.csv" % (root, name))[0])):
        import re
        m = re.search(r"(conv12_i8|conv12_bf16s|conv_bf16s|fc_bf16s|heads_duel|fc_reduce)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:30]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(name, k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
find $O -name "*.csv" -size +2M -delete
cat $R/gpurun_out/pmcq_$tag.txt
