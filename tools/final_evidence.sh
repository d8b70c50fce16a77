#!/bin/bash
# Round-end evidence pass on the GPU box: full GPU tests, default bench, rocprofv3 stats of the bench,
# isolated-forward stats and one PMC pass per counter set.  Outputs under gpurun_out/final/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1 || { tail -5 $O/gpu_tests.log; exit 1; }
tail -1 $O/gpu_tests.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 2; }
cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err || exit 3
echo "bench prof done"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iso_stats -- python3 $R/tools/profile_forward.py > $O/iso_stats.log 2>&1 || exit 4
echo "iso stats done"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/tools/profile_forward.py > $O/pmc_fetch.log 2>&1 || exit 5
echo "pmc fetch done"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/tools/profile_forward.py > $O/pmc_write.log 2>&1 || exit 6
echo "pmc write done"
timeout -k 10 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/tools/profile_forward.py > $O/pmc_sq.log 2>&1 || exit 7
echo "pmc sq done"
find $O -name "*.csv" -size +8M -delete
ls -R $O | head -60
