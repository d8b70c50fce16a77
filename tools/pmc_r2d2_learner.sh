#!/bin/bash
# PMC pass over the isolated R2D2 learner step (bf16x2 mode): MFMA busy / waits / LDS conflicts per kernel.
#   bash tools/pmc_r2d2_learner.sh   (on the GPU box; output under gpurun_out/pmc_r2d2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_r2d2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ITERS=3 PRECISION=bf16x2 timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/time_r2d2_learner.py > $O/sq.log 2>&1 || exit 1
ITERS=3 PRECISION=bf16x2 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/time_r2d2_learner.py > $O/fetch.log 2>&1 || exit 2
cd $R
python3 - <<PY
import csv, glob, collections
def load(d):
    f = sorted(glob.glob("$O/%s/*/*_counter_collection.csv" % d))[-1]
    return list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in load("sq"):
    n = r["Kernel_Name"]
    key = n.split("(")[0][-60:]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[key] += 1
for r in load("fetch"):
    key = r["Kernel_Name"].split("(")[0][-60:]
    agg[key]["FETCH_SIZE"] += float(r["Counter_Value"])
print("kernel | launches | Mcycles/launch | MFMA busy | LDS active | LDS conflict | fetch MB/launch (x2 corr.)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:16]:
    c = max(cnt[k], 1); cyc = v["GRBM_GUI_ACTIVE"] / 8 / c
    print("%s | %d | %.3f | %.1f%% | %.1f%% | %.1f%% | %.1f" % (k, c, cyc / 1e6, 100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / c / (cyc * 1024 + 1),
          100 * v["SQ_LDS_IDX_ACTIVE"] / c / (cyc * 256 + 1), 100 * v["SQ_LDS_BANK_CONFLICT"] / c / (cyc * 256 + 1), v["FETCH_SIZE"] / c * 2 * 1024 / 1e6 / 16))
PY
