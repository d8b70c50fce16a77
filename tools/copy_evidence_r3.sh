#!/bin/bash
# Copies the outputs of tools/final_evidence_r3.sh (gpurun_out/final3) into profiles/r03_*.
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/final3
newest() { ls -t $1 2>/dev/null | head -1; }
cp $F/pmc_table.md profiles/r03_pmc_table.md
cp $F/traffic.json profiles/r03_traffic.json
cp $F/bench.json profiles/r03_bench_latest.json
cp $F/bench_prof.json profiles/r03_bench_under_rocprof.json
cp $F/bench_r2d2.json profiles/r03_bench_r2d2_latest.json
cp $F/forward_modes.log profiles/r03_forward_modes.log
[ -f $F/phase_stamps.log ] && cp $F/phase_stamps.log profiles/r03_phase_stamps.log || true
cp $F/time_sample.json profiles/r03_time_sample_isolated.json
cp $F/r2d2_learner.log profiles/r03_r2d2_learner_isolated.log
cp $F/bench_only_learner.json profiles/r03_bench_only_learner.json
cp $F/bench_only_actor.json profiles/r03_bench_only_actor.json
cp $F/bench_layout_reference_rehearsal.json profiles/r03_bench_layout_reference_rehearsal_2ranks_one_gpu.json
cp $F/bench_rehearsal_2ranks.json profiles/r03_bench_rehearsal_2ranks_one_gpu.json
cp "$(newest "$F/prof_bench/*/*_kernel_stats.csv")" profiles/r03_bench_kernel_stats.csv
[ -f $F/bench_kernel_per_shape.csv ] && cp $F/bench_kernel_per_shape.csv profiles/r03_bench_kernel_per_shape.csv || true
mkdir -p profiles/r03_pmc
for d in pmc_fetch pmc_write pmc_sq; do cp "$(newest "$F/$d/*/*_counter_collection.csv")" profiles/r03_pmc/${d}_counter_collection.csv; done
python3 - <<'PY'
import json
d = json.load(open("profiles/r03_bench_latest.json"))
r = d["roofline"]
print("bench: %.3f M env-steps/s, %.0f grad-steps/s, %.3f ms/step, %.2f forwards/tick; no_reuse %.3f M; f32_mode %.3f M; strict %.3f M" % (
    d["value"] / 1e6, d["grad_steps_per_s"], d["ms_per_step"], d["forwards_per_tick"], d["no_reuse"]["env_steps_per_s"] / 1e6,
    d["f32_mode"]["env_steps_per_s"] / 1e6, d["strict"]["env_steps_per_s"] / 1e6))
print("roofline: %s %s frac %.3f, %.1f us live, traffic %s" % (r["kernel"], r["bound"], r["frac"], r["avg_launch_ms"] * 1e3, r["traffic"]))
print("threaded:", d.get("threaded", {}).get("without_sampler"), d.get("threaded", {}).get("with_sampler"))
print("cpu_baseline:", d["cpu_baseline"]["value"])
PY
