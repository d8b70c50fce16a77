#!/bin/bash
# Copies the outputs of tools/final_evidence.sh (gpurun_out/final, newest file of each kind) into profiles/r02_*.
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/final
newest() { ls -t $1 2>/dev/null | head -1; }
cp $F/pmc_table.md profiles/r02_pmc_table.md
cp $F/bench.json profiles/r02_bench_latest.json
cp $F/bench_prof.json profiles/r02_bench_under_rocprof.json
cp $F/bench_r2d2.json profiles/r02_bench_r2d2_latest.json
cp $F/bench_dedup_2p23.json profiles/r02_bench_dedup_plane_2p23.json
cp $F/forward_modes.log profiles/r02_forward_modes.log
cp $F/time_sample.json profiles/r02_time_sample_isolated.json
cp $F/r2d2_learner.log profiles/r02_r2d2_learner_isolated.log
cp $F/threaded_benchmark.log profiles/r02_threaded_benchmark_final.log
cp $F/threaded_benchmark_r2d2.log profiles/r02_threaded_benchmark_r2d2_final.log
[ -f $F/bench_rehearsal_2ranks.json ] && cp $F/bench_rehearsal_2ranks.json profiles/r02_bench_rehearsal_2ranks_one_gpu.json || true
[ -f $F/bench_only_learner.json ] && cp $F/bench_only_learner.json profiles/r02_bench_only_learner.json || true
[ -f $F/bench_only_actor.json ] && cp $F/bench_only_actor.json profiles/r02_bench_only_actor.json || true
[ -f $F/lds_conflicts.txt ] && cp $F/lds_conflicts.txt profiles/r02_lds_conflict_model_conv12.txt || true
cp "$(newest "$F/prof_bench/*/*_kernel_stats.csv")" profiles/r02_bench_kernel_stats.csv
cp "$(newest "$F/iso_stats/*/*_kernel_stats.csv")" profiles/r02_isolated_kernel_stats.csv
python3 tools/per_shape_stats.py $F/prof_bench profiles/r02_bench_kernel_per_shape.csv
mkdir -p profiles/r02_pmc
for d in pmc_fetch pmc_write pmc_sq; do cp "$(newest "$F/$d/*/*_counter_collection.csv")" profiles/r02_pmc/${d}_counter_collection.csv; done
[ -f $F/gpu_tests.log ] && cp $F/gpu_tests.log profiles/r02_gpu_tests_final.log || true
python3 - <<'PY'
import json
d = json.load(open("profiles/r02_bench_latest.json"))
r = d["roofline"]
print("bench: %.3f M env-steps/s, %.0f grad-steps/s, %.3f ms/step, %.2f forwards/tick; reuse_next_only %.3f M; no_reuse %.3f M; f32_mode %.3f M" % (
    d["value"] / 1e6, d["grad_steps_per_s"], d["ms_per_step"], d["forwards_per_tick"],
    d["reuse_next_only"]["env_steps_per_s"] / 1e6, d["no_reuse"]["env_steps_per_s"] / 1e6,
    d["f32_mode"]["env_steps_per_s"] / 1e6))
print("roofline: %s %s frac %.3f (mfma %.3f, hbm %.3f), %.1f us live, traffic %s" % (
    r["kernel"], r["bound"], r["frac"], r["mfma"]["frac"], r["hbm"]["frac"], r["avg_launch_ms"] * 1e3, r["traffic"]))
print("cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"].get("best_shape", {}).get("value"))
r2 = json.load(open("profiles/r02_bench_r2d2_latest.json"))
print("r2d2: %.1f k env-steps/s, %.1f grad-steps/s" % (r2["value"] / 1e3, r2["grad_steps_per_s"]))
dd = json.load(open("profiles/r02_bench_dedup_plane_2p23.json"))
print("dedup 2^23: %.3f M" % (dd["value"] / 1e6))
PY
grep -m3 "conv12_bf16s\|conv_bf16s\|fc_bf16s" profiles/r02_bench_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
