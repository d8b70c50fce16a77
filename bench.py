#!/usr/bin/env python3
"""bench.py -- Ape-X hot path throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): Ape-X DQN, 80 actor threads x 80 games = 6,400 envs,
one MI355X running actors + learner, replay 2^20 (ring 1,310,720 slots, device resident),
synthetic 84x84x4 uint8 frames already in HBM, A = 18 actions, n = 3, gamma = 0.997,
alpha = 0.6, beta = 0.4, learner batch 512 (pyrela/main.py:28-65 defaults).

One STEP = what the reference does per env-step of every actor thread plus one learner update:
  actor tick over 6,400 envs  (rela/thread_loop.h:74-105 -> rela/dqn_actor.h:153-203)
      act: 1 trunk forward + eps-greedy; post_step: n-step return, TD priority from online(s_t),
      online(s_t+n), target(s_t+n), replay insert of 6,400 transitions.
      online(s_t+n) is the forward act() just ran on the same observation with the same weights and online(s_t)
      the one act() ran n ticks ago (dqn_actor.h:84,161 + apex.py:38,41): each is reused bit-identically when the
      online weights were not re-loaded in between, so every observation gets ONE online and ONE target forward
      per weight version: 2 trunk forwards on most ticks, 3-4 on the n + 1 ticks after a weight publish (every 20
      learner steps: 2.2 on average).  `forwards_per_tick` reports what ran; `no_reuse` carries the same step
      measured with the reuse switched off (all 4 forwards of the reference) and `reuse_next_only` with only the
      same-tick reuse (3 forwards, the headline of rounds 1-2).  Nothing else is cached or skipped.
  learner step                (pyrela/main.py:206-251)
      replay.sample(512) [exact sequential-sum scan + gather] -> ApexAgent.loss -> backward
      -> clip 40 -> RMSprop -> update_priority, hand-written HIP (csrc/learner.hip; RELA_BENCH_LEARNER=torch
      runs PyTorch autograd instead); actor weights re-published every 20 steps, target net every 2,500.

Arithmetic: three modes, all timed by ONE run as regions of the same K steps (`summary`), `value` = the one `--precision`
names.  `f32x3` (the default): f32 results at f32 accuracy from the bf16 matrix cores -- conv2 / conv3 / fc of the
actors' forwards (and the learner's conv forwards and weight-gradient GEMMs) take every f32 operand as THREE exact bf16
parts and six products with f32 accumulation (csrc/gemm_f32emu.h); against an f64 evaluation its Q-values are as close
as the exact-f32-MFMA kernels' and as torch CPU f32's (tests/test_ffnet_gpu.py::test_ffnet_f32x3_is_f32_accurate).
`f32`: exact f32 MFMA (v_mfma_f32_16x16x4_f32) for the actors' forwards and the whole learner step.  `bf16x2`: the fast
mode (two bf16 parts per operand = 16-bit significands, conv1 on the int8 matrix cores fused with conv2 through LDS;
|dQ| < 2e-5 max|Q| against the f32 path) -- a narrower arithmetic, never `value` unless asked for.
Regions: f32x3_mode, f32x3_strict (all 4 forwards of the reference), f32_mode, strict, fast_mode, fast_no_reuse.

Output (rank 0): ONE compact JSON line on stdout (<= 1,900 characters: the driver keeps the last 2,000 of stdout) with the
contract's keys, `roofline` (live HIP events around the dominant forward kernel inside the timed region, both the
algorithmic and the issued-MFMA fraction where they differ, the sampled shader clock), `cpu_baseline` (the reference's
own CPU-thread actor path from oracle/_ref, pinned to the cgroup's CPU share) and `summary` (the four regions and the
threaded leg through rela.Context); the full record -- per-kernel table, every region's repeats and rooflines, the HBM
rooflines of the replay sample path and insert -- goes to gpurun_out/bench_detail_<tag>.json and to stderr.
`python bench.py --gpus N` without a launcher spawns its N ranks itself (before any GPU call).
`--algo r2d2` is the second line: config C4's per-GPU shape with the HIP R2D2 learner.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

T_THREADS, K_GAMES = 80, 80
ROWS = T_THREADS * K_GAMES
NUM_ACTION = 18
MULTI_STEP, GAMMA = 3, 0.997
ALPHA, BETA = 0.6, 0.4
BATCH = 512
REPLAY_CAP = 1 << 20
SEED = 10002

# algorithmic FLOPs per sample-forward: 2 * MACs
FLOP = {"conv1_bf16x3": 2 * 400 * 32 * 256, "conv2_mfma": 2 * 81 * 64 * 512, "conv3_mfma": 2 * 49 * 64 * 576,
        "fc_mfma": 2 * 3136 * 512, "heads_mfma": 2 * 512 * 19,
        # bf16x2 mode: conv1 and conv2 are ONE kernel (conv1's output stays in LDS)
        "conv12_fused": 2 * 400 * 32 * 256 + 2 * 81 * 64 * 512}
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA
# algorithmic HBM bytes per sample-forward of each kernel: input tensor + output tensor (activations are 4 B per
# element in both precision modes: f32, or bf16 hi + lo), u8 frames in, f32 h out
BYTES_PER_SAMPLE = {"conv1_bf16x3": 28224 + 400 * 32 * 4, "conv2_mfma": 400 * 32 * 4 + 81 * 64 * 4,
                    "conv3_mfma": 81 * 64 * 4 + 49 * 64 * 4, "fc_mfma": 49 * 64 * 4 + 512 * 4, "heads_mfma": 512 * 4 + 32 * 4,
                    "conv12_fused": 28224 + 81 * 64 * 4}
# f32x3 mode from 512 rows (csrc/gemm_s3.h): activations travel between the layers as split3 records, 6 B per element
BYTES_PER_SAMPLE_F32X3 = {"conv12_fused": 28224 + 81 * 64 * 6, "conv3_mfma": 81 * 64 * 6 + 49 * 64 * 6, "fc_mfma": 49 * 64 * 6 + 512 * 4}
PEAK_HBM_GBS = 8000.0
DTYPE_F32X3 = ("f32 (conv2 / conv3 / fc of the actors' forwards: every f32 operand split EXACTLY in three bf16 parts, six "
               "products each on v_mfma_f32_16x16x32_bf16, f32 accumulation -- error against f64 <= the f32 MFMA kernels' "
               "and torch CPU f32's, tests/test_ffnet_gpu.py::test_ffnet_f32x3_is_f32_accurate; everything else f32)")
DTYPE_NOTE = ("f32 results from split-bf16 MFMA (bf16 hi + lo operands = 16 significant bits, 3 bf16 products per product, "
              "f32 accumulate; conv1: u8 frames x 24-bit fixed-point weights as three int8 digit products, exact i32 sums): "
              "|dQ| < 2e-5 * max|Q| against the exact f32 mode (tests/test_ffnet_gpu.py)")
# what the HIP Ape-X learner step computes in, per --precision (csrc/learner.hip)
LEARNER_PRECISION_NOTE = {
    "f32": "f32 throughout (exact f32 MFMA forwards, f32 MFMA GEMM backward, f32 clip + RMSprop)",
    "f32x3": "f32 throughout: conv2 / conv3 of the three forwards on the f32-accurate three-part bf16 kernels (csrc/gemm_f32emu.h), "
             "conv1 / fc / heads, f32 MFMA GEMM backward, clip + RMSprop as in the f32 mode",
    "bf16x2": "mixed: td_err's two gradient-free forwards (online(s'), target(s')) with conv trunk on split-bf16 MFMA; "
              "conv1 weight gradient and conv2 / conv3 data gradients on bf16 MFMA (hi + lo operands, f32 accumulate); "
              "online(s) forward whose activations / ReLU masks feed the backward, fc and head GEMMs, conv2 / conv3 weight "
              "gradients, loss, clip, RMSprop in f32",
}


def generate_eps(base_eps, alpha, num_actor):
    """eps_i = base ** (1 + i/(N-1) * alpha)   (pyrela/utils.py:88-96)."""
    if num_actor == 1:
        return [base_eps]
    return [base_eps ** (1 + i / (num_actor - 1) * alpha) for i in range(num_actor)]


def host_cores():
    """CPUs this process may really use: the scheduler affinity capped by the cgroup quota (`cpu.max` of cgroup v2 or
    cfs_quota / cfs_period of v1).  -> (cores, {"affinity": n, "cgroup_quota": q or None})"""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    cores = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return cores, {"affinity": aff, "cgroup_quota": quota}


def pin_to_cores(n):
    """-> preexec_fn that pins a CHILD process (it never touches the GPU) to the first n CPUs of this process's affinity
    set, so that `cpu_baseline.cores` is what its threads really ran on"""
    cpus = sorted(os.sched_getaffinity(0))[:n]

    def fn():
        os.sched_setaffinity(0, cpus)
    return fn


class ClockSampler:
    """Samples the GPU's shader clock and socket power from sysfs (hwmon freq1_input / power1_average, else the starred
    level of pp_dpm_sclk) on a background thread while a timed region runs: the same kernel takes 156-204 us on
    different boxes of the pool, and without the sustained clock two rounds' lines cannot be compared.  Best effort:
    summary() is {} when the files are not readable."""

    def __init__(self, pci_bus_id=None, period=0.02):
        import glob
        import threading

        self.freq_file = self.dpm_file = self.power_file = None
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        if pci_bus_id:
            want = [c for c in cards if os.path.basename(os.path.realpath(c)).lower() == pci_bus_id.lower()]
            cards = want or cards
        for c in cards:
            f = sorted(glob.glob(os.path.join(c, "hwmon", "hwmon*", "freq1_input")))
            pw = sorted(glob.glob(os.path.join(c, "hwmon", "hwmon*", "power1_average"))
                        + glob.glob(os.path.join(c, "hwmon", "hwmon*", "power1_input")))
            dpm = os.path.join(c, "pp_dpm_sclk")
            if f or os.path.exists(dpm):
                self.freq_file = f[0] if f else None
                self.dpm_file = dpm if os.path.exists(dpm) else None
                self.power_file = pw[0] if pw else None
                break
        self.period, self.mhz, self.watts = period, [], []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        try:
            if self.freq_file:
                self.mhz.append(int(open(self.freq_file).read()) / 1e6)
            elif self.dpm_file:
                for ln in open(self.dpm_file).read().splitlines():
                    if ln.rstrip().endswith("*"):
                        self.mhz.append(float(ln.split(":")[1].strip().split("M")[0]))
            if self.power_file:
                self.watts.append(int(open(self.power_file).read()) / 1e6)
        except (OSError, ValueError, IndexError):
            pass

    def _run(self):
        while not self._stop.is_set():
            self._read()
            self._stop.wait(self.period)

    def __enter__(self):
        if self.freq_file or self.dpm_file:
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th.is_alive():
            self._th.join()

    def summary(self):
        out = {}
        if self.mhz:
            out["sclk_mhz"] = round(float(np.median(self.mhz)), 0)
            out["sclk_mhz_min_max"] = [round(min(self.mhz)), round(max(self.mhz))]
        if self.watts:
            out["power_w"] = round(float(np.median(self.watts)), 0)
        return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks as fresh child
    processes of THIS process, which has not touched the GPU (no torch.cuda call yet, nothing re-executes a GPU process),
    hand them RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would, and exit with the worst code.
    Rank 0 prints the JSON line on the inherited stdout."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RELA_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in procs:  # one rank died: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            p.kill()
    return rc


LINE_BUDGET = 1900  # the driver stores the last 2,000 characters of stdout: the whole JSON line must fit


def rnd(x, sig=5):
    """floats to `sig` significant digits (keeps the JSON line short); passes everything else through"""
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    if isinstance(x, dict):
        return {k: rnd(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [rnd(v, sig) for v in x]
    return x


def emit(line, detail, tag):
    """Prints the ONE compact JSON line (rank 0) and keeps the long record next to it: the driver stores only the last
    2,000 characters of stdout, so every number that must be seen goes into the compact line (<= LINE_BUDGET
    characters, optional keys dropped from the end of `optional` until it fits) and the full record -- per-kernel
    table, every region's repeats, notes -- goes to gpurun_out/bench_detail_<tag>.json and to stderr."""
    path = None
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "bench_detail_%s.json" % tag)
        with open(path, "w") as f:
            json.dump(detail, f, indent=1)
        line["detail"] = os.path.relpath(path, ROOT)
    except OSError:
        pass
    print(json.dumps(detail), file=sys.stderr, flush=True)
    line = rnd(line)
    for k in ("detail", "clock", "grad_steps_per_s", "train_samples_per_s"):  # dropped in this order if it ever gets too long
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
        line.pop(k, None)
    s = json.dumps(line)
    assert len(s) <= LINE_BUDGET, "bench line is %d characters: the driver would cut it" % len(s)
    print(s, flush=True)
    return s


# rela_prof label -> substring of the HIP kernel name in the rocprofv3 CSVs
PMC_KERNEL_OF = {"conv12_fused": "conv12_", "conv3_mfma": "conv_bf16s<", "fc_mfma": "fc_bf16s",
                 "conv1_bf16x3": "conv1_bf16x3", "conv2_mfma": "ConvCfg<32", "replay_scatter_rows": "replay_scatter_rows",
                 "lstm_gates_mfma": "GemmCfg<3648", "lstm_gates_x_bf16": "gemm_rec64_nt"}


def traffic_from_profiles(label, precision=None):
    """HBM-side bytes per launch of the kernel behind a rela_prof label, from the PMC passes TRACKED under profiles/
    (PMC counters cannot be read from inside this process): first a summary profiles/r0N_traffic.json, else straight
    from profiles/r0N_pmc/pmc_{fetch,write}_counter_collection.csv -- separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE runs of the isolated forward at the same N = 6400 (tools/final_evidence.sh).  Corrections as
    MI355X_MICROARCH.md prescribes: both counters are in KB; FETCH_SIZE reports half the bytes of wide coalesced
    reads on gfx950 and is doubled.  Newest round first.  -> (bytes, source) or (None, None)."""
    import csv
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            kernels = json.load(open(path))["kernels"]
            # (the f32x3 mode's kernels share the rela_prof labels of the f32 mode's: their PMC records carry a suffix)
            rec = kernels.get(label + "_f32x3") if precision == "f32x3" else None
            if rec is None:
                if precision == "f32x3" and label in ("conv12_fused", "conv2_mfma", "conv3_mfma", "fc_mfma"):
                    continue  # never report another kernel's traffic for this one
                rec = kernels.get(label)
        except (OSError, ValueError, KeyError):
            continue
        if rec is not None:
            return rec["read_bytes"] + rec["write_bytes"], "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, not this run)" % os.path.basename(path)
    needle = PMC_KERNEL_OF.get(label)
    for d in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc")), reverse=True):
        tot = {}
        for kind, counter, scale in (("fetch", "FETCH_SIZE", 2 * 1024.0), ("write", "WRITE_SIZE", 1024.0)):
            try:
                rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(d, "pmc_%s_counter_collection.csv" % kind)))
                        if needle and needle in r["Kernel_Name"] and r["Counter_Name"] == counter]
            except (OSError, ValueError, KeyError):
                rows = []
            if rows:
                tot[kind] = scale * sum(rows) / len(rows)
        if len(tot) == 2:
            return tot["fetch"] + tot["write"], ("profiles/%s/pmc_{fetch,write}_counter_collection.csv (rocprofv3 --pmc, separate "
                                                 "passes, FETCH_SIZE x2 per the gfx950 correction; not this run)" % os.path.basename(d))
    return None, None


def threaded_leg(seconds=1.0, epochs=3, threads=64, games=100, precision="f32"):
    """The metric as the reference defines it (pyrela/benchmark.py:73-109): sum of DQNActor.num_act() deltas per
    second through rela.Context + BasicThreadLoop + DQNActor + FFPrioritizedReplay of the drop-in `rela` module --
    C++ actor threads stepping HOST envs, per-step host -> HBM observation upload, the headline's arithmetic
    (RELA_PRECISION = --precision: f32x3 runs conv2 / conv3 of the 3,200-row cohorts on the three-part bf16 kernels, their
    fc -- below 4,096 rows -- and everything else in exact f32) -- without and with a
    concurrent unthrottled B = 512 sample / update_priority loop, in a child process on this box's host cores.  Bounded:
    the reference protocol is 6 x 30 s windows per mode, this leg runs `epochs` x `seconds` (windows without a sampler
    end before the 2^21 replay's ring is full -- nothing evicts in that mode); tools/threaded_protocol.sh
    runs the full protocol once per round (profiles/).  Two env flavours: `fresh` = four new LCG planes per env-step
    (SURVEY 8d), `sliding` = Atari's frame stacking, ONE new plane per step (atari/game_state.h:53-82), for which only
    that plane crosses PCIe."""
    out = {"metric": "env-steps/s = d(sum of DQNActor.num_act())/dt through rela.Context / BasicThreadLoop / DQNActor "
                     "(pyrela/benchmark.py:73-109), host envs + H2D upload included", "unit": "env-steps/s",
           "threads": threads, "games_per_thread": games, "host_cores": host_cores()[0],
           "host_cores_source": host_cores()[1], "window_s": seconds, "windows": epochs,
           "sampler": "unthrottled B=512 sample + update_priority loop on the Python thread", "RELA_PRECISION": precision,
           "note": "bounded sample of the reference's 6 x 30 s protocol; mean of the last half of the valid windows"}
    t0 = time.time()
    for env in ("fresh", "sliding"):
        # r5: the sliding-stack env -- the drop-in's headline -- gets ONE window of the reference
        # protocol's own length, 30 s, without and with the sampler (1 s windows moved 1.6 <-> 2.1 M between two passes); the
        # fresh-LCG env (PCIe-bound) keeps the short windows
        long_ = env == "sliding" and os.environ.get("RELA_BENCH_THREADED_LONG", "1") != "0"
        cmd = [sys.executable, os.path.join(ROOT, "rela_amd", "pyrela", "benchmark.py"), "--grid", "%dx%d" % (threads, games),
               "--epoch_sec", "30" if long_ else str(seconds), "--num_epoch", "1" if long_ else str(epochs),
               "--replay_buffer_size", str(1 << (22 if long_ else 21)), "--env", env] + (["--burn_in_frames", "20000"] if long_ else [])
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, RELA_PRECISION=precision))
            line = [l for l in res.stdout.splitlines() if l.startswith("act rate: without sample:")][-1]
            without, with_ = (float(x.split(":")[-1]) for x in line[len("act rate: "):].split(","))
        except Exception as e:  # noqa: BLE001  (reported, never fatal for the headline)
            out[env] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            continue
        out[env] = {"without_sampler": without, "with_sampler": with_, "window_s": 30.0 if long_ else seconds,
                    "windows": 1 if long_ else epochs}
    out["wall_s"] = time.time() - t0
    return out


def cpu_baseline_reference():
    """The REAL reference's CPU-thread actor path (oracle/_ref/rela*.so, compiled from the reference's
    sources where they exist; only the built module travels) timed on this box's host cores by
    oracle/ref_actor_bench.py in a child process: first at the headline's shape (80 threads x 80 envs, the
    threads oversubscribing the cores as the reference's README does with 80 threads on 40 cores), then
    with one thread per core x 20 envs (its best shape on few cores, reported as `best_shape`).
    None if the prebuilt module is absent or fails."""
    import glob

    if not glob.glob(os.path.join(ROOT, "oracle", "_ref", "rela*.so")):
        return None
    # cores = the scheduler affinity capped by the cgroup's CPU quota (a box shows 256 CPUs and grants 16); the child is
    # PINNED to that many CPUs, so the label is what its threads ran on
    cores, core_info = host_cores()
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")

    def run(threads, games, seconds, warmup):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_actor_bench.py"), "--threads",
                              str(threads), "--games", str(games), "--seconds", str(seconds), "--warmup", str(warmup),
                              "--num_action", str(NUM_ACTION)], env=env, capture_output=True, text=True, timeout=240,
                             preexec_fn=pin_to_cores(cores))
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])

    try:
        rec = run(T_THREADS, K_GAMES, 14.0, 6.0)
    except Exception:  # noqa: BLE001  (any failure -> fall back to the port)
        return None
    res = {"value": rec["env_steps_per_s"], "unit": "env-steps/s", "cores": cores, "kind": "reference",
           "cores_source": dict(core_info, pinned=True),
           "sample": "the reference's own C++ actor threads (oracle/_ref/rela, g++ -O2) with a CPU TorchScript Ape-X "
                     "agent at the headline's shape: %d threads x %d synthetic envs on %d host cores, OMP_NUM_THREADS=1, "
                     "A=%d, n=3, %.1f s window after 6 s warm-up, light B=32 sampler evicting the overflow "
                     "(pyrela/benchmark.py protocol)"
                     % (rec["threads"], rec["games"], cores, rec["num_action"], rec["seconds"])}
    try:
        best = run(cores, 20, 10.0, 4.0)
        res["best_shape"] = {"value": best["env_steps_per_s"], "threads": best["threads"], "games": best["games"],
                             "seconds": best["seconds"]}
    except Exception:  # noqa: BLE001
        pass
    return res



def cpu_baseline_reference_r2d2():
    """The REAL reference's R2D2 CPU-thread actor path (oracle/_ref/h6/rela*.so: the reference compiled from a scratch copy
    with SURVEY H6's one-line torch-version fix) timed by oracle/ref_actor_bench.py --algo r2d2 in a child process pinned
    to the box's CPU share.  Bounded shape: one thread per core x 20 envs (every env holds a window of 123 frame stacks on
    the host; C4's 3,200 envs would need 11 GB).  None if the prebuilt module is absent or fails."""
    import glob

    if not glob.glob(os.path.join(ROOT, "oracle", "_ref", "h6", "rela*.so")):
        return None
    cores, core_info = host_cores()
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_actor_bench.py"), "--algo", "r2d2",
                              "--threads", str(cores), "--games", "20", "--seconds", "14", "--warmup", "6",
                              "--num_action", str(NUM_ACTION)], env=env, capture_output=True, text=True, timeout=240,
                             preexec_fn=pin_to_cores(cores))
        rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:  # noqa: BLE001  (reported, never fatal)
        return {"value": None, "unit": "env-steps/s", "cores": cores, "kind": "reference",
                "sample": "failed: %s: %s" % (type(e).__name__, str(e)[:200])}
    return {"value": rec["env_steps_per_s"], "unit": "env-steps/s", "cores": cores, "kind": "reference",
            "cores_source": dict(core_info, pinned=True),
            "sample": "the reference's own C++ R2D2 actor threads (oracle/_ref/h6/rela, g++ -O2, H6 fix) with a CPU "
                      "TorchScript R2D2 agent: %d threads x %d synthetic envs on %d host cores, OMP_NUM_THREADS=1, A=%d, "
                      "seq 80 / burn-in 40 / n 3, %.1f s window after 6 s warm-up"
                      % (rec["threads"], rec["games"], cores, rec["num_action"], rec["seconds"])}


def cpu_baseline(budget_s=15.0):
    """The oracle (plain-C port of the same path) timed on this box's host cores: one actor tick
    (act + n-step + TD priority + replay insert) over a bounded number of envs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth import synth_obs, synth_params

    odir = os.path.join(ROOT, "oracle")
    so = os.path.join(odir, "liboracle_native.so")
    srcs = [os.path.join(odir, f) for f in ("replay_oracle.c", "mt19937.c", "nstep_oracle.c", "dqn_oracle.c")]
    subprocess.run(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-ffp-contract=off", "-shared", "-o", so] + srcs
                   + ["-lm", "-lpthread"], check=True)
    lib = C.CDLL(so)
    cores = host_cores()[0]  # affinity capped by the cgroup quota: the GPU box's CPU share for one GPU
    lib.oracle_set_threads.argtypes = [C.c_int]
    lib.oracle_set_threads(cores)
    A = NUM_ACTION

    class Net(C.Structure):
        _fields_ = [("num_action", C.c_int)] + [(n, C.POINTER(C.c_float)) for n in
                                                ("c1w", "c1b", "c2w", "c2b", "c3w", "c3b", "l1w", "l1b", "vw", "vb",
                                                 "aw", "ab")]

    keys = ["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias",
            "linear.0.weight", "linear.0.bias", "fc_v.weight", "fc_v.bias", "fc_a.weight", "fc_a.bias"]

    def mk(seed):
        p = synth_params(A, seed)
        net = Net()
        net.num_action = A
        keep = []
        for (f, _), k in zip(Net._fields_[1:], keys):
            a = np.ascontiguousarray(p[k], np.float32)
            keep.append(a)
            setattr(net, f, a.ctypes.data_as(C.POINTER(C.c_float)))
        net._keep = keep
        return net

    on, tg = mk(1), mk(2)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    up = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    lib.oracle_replay_new.restype = C.c_void_p
    lib.oracle_replay_new.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float]
    lib.oracle_replay_add.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_float)]
    lib.oracle_apex_priority.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_float, C.c_void_p,
                                                                                             C.c_void_p]
    rep = lib.oracle_replay_new(1 << 16, SEED, ALPHA, BETA)

    def tick(n_env):
        s, ns = synth_obs(n_env, 1), synth_obs(n_env, 2)
        legal = np.ones((n_env, A), np.float32)
        q = np.zeros((n_env, A), np.float32)
        act = np.zeros(n_env, np.int64)
        rh = np.zeros((MULTI_STEP + 1, n_env), np.float32)
        th = np.zeros((MULTI_STEP + 1, n_env), np.uint8)
        r, b, prio = np.zeros(n_env, np.float32), np.zeros(n_env, np.float32), np.zeros(n_env, np.float32)
        t = np.zeros(n_env, np.uint8)
        t0 = time.perf_counter()
        for g0 in range(0, n_env, K_GAMES):  # one reference actor thread = K envs per call
            g1 = min(n_env, g0 + K_GAMES)
            m = g1 - g0
            lib.oracle_ffnet_forward(C.byref(on), m, up(s[g0:g1]), fp(legal[g0:g1]), fp(q[g0:g1]))
            lib.oracle_greedy(m, A, fp(q[g0:g1]), fp(legal[g0:g1]), ip(act[g0:g1]))
            lib.oracle_nstep_pop(MULTI_STEP, m, C.c_float(GAMMA), fp(np.ascontiguousarray(rh[:, g0:g1])),
                                 up(np.ascontiguousarray(th[:, g0:g1])), fp(r[g0:g1]), fp(b[g0:g1]), up(t[g0:g1]))
            lib.oracle_apex_priority(C.byref(on), C.byref(tg), m, up(s[g0:g1]), fp(legal[g0:g1]), ip(act[g0:g1]),
                                     fp(r[g0:g1]), fp(b[g0:g1]), up(ns[g0:g1]), fp(legal[g0:g1]),
                                     C.c_float(GAMMA ** MULTI_STEP), None, fp(prio[g0:g1]))
            lib.oracle_replay_add(rep, m, ip(np.arange(g0, g1, dtype=np.int64)), fp(prio[g0:g1]))
        return time.perf_counter() - t0

    probe_n = max(cores, 8)
    t_probe = tick(probe_n)
    per_env = t_probe / probe_n
    n_env = int(min(ROWS, max(K_GAMES, (budget_s / max(per_env, 1e-6)) // K_GAMES * K_GAMES)))
    dt = tick(n_env)
    return {"value": n_env / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "one actor tick (act + n-step + TD priority with 3 forwards + replay insert) over %d envs in "
                      "groups of K=%d, oracle/ plain-C port, gcc -O3 -march=native, %d pthreads, %.1f s"
                      % (n_env, K_GAMES, cores, dt)}


# ---- R2D2 (BASELINE config C4's shapes on one GPU per rank) ------------------------------------------
R2_ROWS, R2_K = 3200, 80            # 40 actor threads x 80 envs per GPU
R2_SEQ, R2_BURN, R2_ETA = 80, 40, 0.9  # pyrela/scripts/ref_run_r2d2.sh:12-18
R2_BATCH, R2_ALPHA, R2_BETA = 64, 0.9, 0.6
R2_REPLAY_CAP = 8192                # sequences of 123 steps x 28,224 B = 3.47 MB each (ring 10,240: 35.5 GB)
R2_EPISODE = 400
FLOP_LSTM_GATES = 2 * 3648 * 2048   # [x | h] x W per env-step of one LSTM step


def bench_r2d2(args, world, rank, device):
    """One STEP = one R2D2 actor tick over 3,200 envs (act: trunk + LSTM step + eps-greedy; post_step: n-step
    return, per-step priority from online(s_t, h_t) / target(s_t+n, h_t+n) [online.act(s_t+n) is act's own step,
    reused], sequence windows, eta-aggregated priority, insert of the finished sequences) + one learner update
    (sample 64 sequences of 123 steps -> R2D2Agent.loss -> BPTT -> clip -> Adam -> update_priority, hand-written
    HIP, csrc/learner_r2d2.hip).  Frames are device-resident; rewards / terminals come from the host because the
    window bookkeeping branches on them (rela/r2d2_actor.h:29-87)."""
    import torch.distributed as dist

    from rela_amd import _capi as capi
    from rela_amd.engine import LSTMNetHandle, R2D2ActorEngine
    from rela_amd.learner import HipR2D2Learner
    from rela_amd.pyrela.net import AtariLSTMNet
    from rela_amd.pyrela.r2d2 import R2D2Agent
    from rela_amd.replay import RNNReplay

    assert R2_BATCH % world == 0
    B_LOCAL = R2_BATCH // world
    T = R2_BURN + R2_SEQ + MULTI_STEP
    torch.manual_seed(SEED + 2 + rank)
    agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, NUM_ACTION), "cpu", MULTI_STEP, GAMMA, R2_ETA, R2_SEQ, R2_BURN,
                      0).to(device)
    if world > 1:
        for p in agent.parameters():
            dist.broadcast(p.data, 0)
    learner = HipR2D2Learner.from_agent(agent, B_LOCAL, lr=6.25e-5, eps=1.5e-4, grad_clip=40.0)
    online, target = LSTMNetHandle(NUM_ACTION, device), LSTMNetHandle(NUM_ACTION, device)
    online.load_state_dict(agent.online_net.state_dict())
    target.load_state_dict(agent.target_net.state_dict())
    # --precision bf16x2 (default): the conv trunks of the actors' nets and of the learner's target net on split-bf16
    # MFMA; the LSTM gate GEMMs, the heads and the learner's online pass stay f32
    online.set_precision(args.precision)
    target.set_precision(args.precision)
    learner.set_precision(args.precision)
    replay = RNNReplay(args.replay_cap, SEED + rank, R2_ALPHA, R2_BETA, 0, NUM_ACTION, T, device)
    eps_all = generate_eps(0.4, 7, R2_ROWS * world)
    engine = R2D2ActorEngine(R2_ROWS, R2_K, NUM_ACTION, MULTI_STEP, GAMMA, R2_SEQ, R2_BURN, R2_ETA, replay,
                             eps_all[rank * R2_ROWS:(rank + 1) * R2_ROWS], device, seed=SEED + rank)
    g = torch.Generator(device=device)
    g.manual_seed(SEED + 7 + rank)
    rng = np.random.default_rng(SEED + rank)
    reward = rng.integers(-1, 2, R2_ROWS).astype(np.float32)
    phase = rng.integers(0, R2_EPISODE, R2_ROWS)
    actor_stream = torch.cuda.Stream(device=device)
    # (the learner's stream: high priority unless RELA_BENCH_PRIO=2 -- its persistent chain kernels need every block
    # resident before the first grid barrier releases)
    main_stream = (torch.cuda.Stream(device=device) if os.environ.get("RELA_BENCH_PRIO", "1") == "2"
                   else torch.cuda.Stream(device=device, priority=-1))
    tick_idx, step_idx = [0], [0]

    def actor_tick():
        k = tick_idx[0]
        if k <= MULTI_STEP:  # the n+1 history slots get their static synthetic frames on the first pass
            engine.next_obs_slot().copy_(torch.randint(0, 256, (R2_ROWS, 4, 84, 84), dtype=torch.uint8, device=device,
                                                       generator=g))
            actor_stream.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(actor_stream):
            engine.act(online)
            term = ((k + phase) % R2_EPISODE == R2_EPISODE - 1).astype(np.uint8)
            engine.post_step(reward, term, online, target, nonblocking=True)
        tick_idx[0] += 1

    R2_PIPE = os.environ.get("RELA_BENCH_PIPELINE", "1") == "1"
    r2_pending, r2_slot = [None], [0]
    if R2_PIPE:
        replay.set_deferred_wait(True)

    def one_step():
        with torch.cuda.stream(main_stream):
            k = step_idx[0]
            if k % 2500 == 0:
                learner.sync_target_with_online()
            if k % 20 == 0:
                main_stream.wait_stream(actor_stream)
                learner.publish(online, target)
                actor_stream.wait_stream(main_stream)
            if r2_pending[0] is None:
                r2_pending[0] = replay.sample(B_LOCAL, slot=r2_slot[0])
                r2_slot[0] ^= 1 if R2_PIPE else 0
            batch, weight = r2_pending[0]
            r2_pending[0] = None
            actor_tick()
            if R2_PIPE:
                # as the Ape-X step below: update_priority and the next sample (0.44 ms: the time-major gather of
                # 64 x 3.47 MB) are queued between the forward and the backward half and run next to BPTT
                replay.wait()
                loss, prio, _ = learner.loss(batch, weight)
                replay.update_priority(prio)
                r2_pending[0] = replay.sample(B_LOCAL, slot=r2_slot[0])
                r2_slot[0] ^= 1
                learner.grad()
                if world > 1:
                    g = learner.flat()[1]
                    dist.all_reduce(g, op=dist.ReduceOp.SUM)
                    g.div_(world)
                learner.apply()
            else:
                loss, prio = learner.step(batch, weight, world_size=world)
                replay.update_priority(prio)
            main_stream.wait_stream(actor_stream)
        step_idx[0] += 1

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    while replay.size() < 4 * R2_BATCH:  # untimed: actors alone until the replay can serve batches
        actor_tick()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        one_step()
    sync_all()
    buf = C.create_string_buffer(1 << 16)

    def prof_summary():
        capi.check(capi.lib.rela_prof_summary_json(buf, len(buf)), "rela_prof_summary_json")
        return json.loads(buf.value.decode())

    cur_prec = [args.precision]

    def set_all_precision(mode):
        if mode == cur_prec[0]:  # (setting a mode bumps the nets' weight version: memoised forwards would be recomputed)
            return
        cur_prec[0] = mode
        online.set_precision(mode)
        target.set_precision(mode)
        learner.set_precision(mode)

    try:
        pr = torch.cuda.get_device_properties(torch.device(device))
        pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:  # noqa: BLE001
        pci = None

    def bracketed(k):
        """EXACTLY k steps between barrier + synchronize on both sides; MAX over ranks; -> seconds"""
        sync_all()
        t0 = time.perf_counter()
        for _ in range(k):
            one_step()
        sync_all()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed_region(precision, reuse, settle_steps):
        set_all_precision(precision)
        engine.set_reuse(reuse)
        for _ in range(settle_steps):
            one_step()
        sync_all()
        # the roofline's kernel only (the actors' gate GEMM in whichever form the mode runs it; r5: the f32x3 label was missing
        # here, and the line priced the h-only kernel of that mode with the whole gate GEMM's FLOPs)
        capi.lib.rela_prof_set_filter(b"lstm_gates_mfma,lstm_gates_x_bf16,lstm_gates_x_f32x3")
        capi.lib.rela_prof_enable(1)
        add0 = replay.num_add()
        with ClockSampler(pci) as clk:
            ms = [bracketed(args.steps) / args.steps * 1e3 for _ in range(args.repeats)]
        capi.lib.rela_prof_enable(0)
        adds = replay.num_add() - add0
        if world > 1:
            a = torch.tensor([float(adds)], device=device, dtype=torch.float64)
            dist.all_reduce(a, op=dist.ReduceOp.SUM)
            adds = float(a.item())
        med = float(np.median(ms))
        return {"precision": precision, "reuse": reuse, "ms": ms, "ms_per_step": med, "prof": prof_summary(),
                "adds_per_s": adds / (sum(ms) * 1e-3 * args.steps), "clock": clk.summary(),
                "env_steps_per_s": R2_ROWS * world / (med * 1e-3)}

    head = timed_region(args.precision, 1, 0)
    capi.lib.rela_prof_set_filter(None)
    capi.lib.rela_prof_enable(1)
    k_all = max(5, args.steps // 10)
    for _ in range(k_all):
        one_step()
    sync_all()
    capi.lib.rela_prof_enable(0)
    prof_all = prof_summary()
    # further regions of the same run, as in the Ape-X line: the reference's work (all forwards recomputed) in the
    # headline's arithmetic, and the other arithmetic with the forwards memoised
    # (f32x3 for the recurrent net: conv2 / conv3 of the trunks on the three-part bf16 kernels, everything else exact f32)
    name_of = {("f32", 1): "f32_mode", ("f32", 0): "strict", ("bf16x2", 1): "fast_mode", ("bf16x2", 0): "fast_no_reuse",
               ("f32x3", 1): "f32x3_mode", ("f32x3", 0): "f32x3_strict"}
    regions = {name_of[(args.precision, 1)]: head}
    settle = 2 * (MULTI_STEP + 1)
    regions[name_of[(args.precision, 0)]] = timed_region(args.precision, 0, settle)
    for other in ("f32x3", "f32", "bf16x2"):
        if other != args.precision:
            regions[name_of[(other, 1)]] = timed_region(other, 1, settle)
    set_all_precision(args.precision)
    engine.set_reuse(1)
    st = replay.debug_state()
    assert st["dev_error"] == 0
    learner.check()  # raises if a grid barrier of the persistent recurrent kernels ever gave up (updates were skipped)

    def roofline_of(region):
        # the actors' gate GEMM: one f32 MFMA kernel over [x | h] (f32 mode), or (bf16x2 mode, >= 1,024 rows) the x part as a
        # split-bf16 GEMM (three bf16 MFMAs per product) followed by the f32 kernel over h only: the roofline is the x part's
        prof = region["prof"]
        split_gates = "lstm_gates_x_bf16" in prof
        x3_gates = "lstm_gates_x_f32x3" in prof  # r5, f32x3: the x part as a three-part GEMM over a3's records (six products)
        roof_name = "lstm_gates_x_bf16" if split_gates else ("lstm_gates_x_f32x3" if x3_gates else "lstm_gates_mfma")
        rec = prof.get(roof_name, {"total_ms": 0.0, "count": 1})
        avg_ms = rec["total_ms"] / max(rec["count"], 1)
        flops = (2 * 3136 * 2048 if (split_gates or x3_gates) else FLOP_LSTM_GATES) * R2_ROWS
        products = 3 if split_gates else (6 if x3_gates else 0)
        peak = PEAK_BF16_MFMA_TFLOPS / (6 if x3_gates else 1) if products else PEAK_F32_MFMA_TFLOPS
        ach = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else None
        traffic, src = traffic_from_profiles(roof_name)
        roof = {"kernel": roof_name, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": None if ach is None else ach / peak, "traffic": traffic, "traffic_source": src,
                "avg_launch_ms": avg_ms, "launches": rec["count"], "algorithmic_flop_per_launch": flops,
                "instruction": ("v_mfma_f32_16x16x32_bf16 x3 (split operands)" if split_gates else
                                "v_mfma_f32_16x16x32_bf16, six products per f32 product (three exact bf16 parts per operand): "
                                "peak = 2,500 / 6 TFLOP/s" if x3_gates else "v_mfma_f32_16x16x4_f32")}
        if split_gates and ach is not None:
            roof["frac_issued"] = 3 * ach / peak  # three bf16 products issued per algorithmic product
        if x3_gates and ach is not None:  # the three readings of the same launch, side by side
            roof["frac_issued"] = ach / peak
            roof["frac_algorithmic_bf16"] = ach / PEAK_BF16_MFMA_TFLOPS
            roof["frac_of_f32_mfma_peak"] = ach / PEAK_F32_MFMA_TFLOPS
        roof.update(region["clock"])
        return roof

    if rank == 0:
        roof = roofline_of(head)
        sample_ms = sum(v["total_ms"] for k, v in prof_all.items() if k.startswith("seq_") or k.startswith("replay_gather")
                        or k in ("replay_targets", "replay_search", "replay_pop", "replay_is_weights")) / k_all
        sample_bytes = 4 * st["safe_size"] + B_LOCAL * sum(replay.row_bytes)
        learner_ms = sum(v["total_ms"] for k, v in prof_all.items() if k.startswith("learner_")) / k_all
        ms_med = head["ms_per_step"]
        dtype = {"f32": "f32", "f32x3": "f32 (every trunk forward and the input half of every forward gate GEMM with f32 operands as "
                 "three exact bf16 parts on the bf16 MFMA, csrc/gemm_s3.h: f32 accuracy; recurrences, heads, backward in exact f32)"}.get(
            args.precision,
            "f32 results from split-bf16 MFMA (bf16 hi+lo operands, f32 accumulate): the actors' conv trunks and gate GEMM "
            "(h, c, Q within 4e-6), the learner's trunks, LSTM GEMMs and conv gradients; recurrences, cells, heads in f32")
        detail = {
            "metric": "env-steps/s (R2D2 Atari 84x84x4, seq 80 / burn-in 40 / n 3, actor tick + learner grad-step)",
            "value": head["env_steps_per_s"], "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_med, "repeats": args.repeats, "ms_per_step_repeats": head["ms"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "R2D2 LSTM (BASELINE config C4's shapes), 40 threads x 80 games (3200 envs) per GPU, "
                                   "actor + learner on one MI355X, sequence replay of %d x 3.47 MB device-resident, A=18, "
                                   "seq 80 / burn-in 40 / n 3, ONE learner batch of 64 sequences per step for the whole "
                                   "job; device-resident static frames, rewards / terminals from the host"
                                   % args.replay_cap,
                       "envs_per_gpu": R2_ROWS, "replay_capacity": args.replay_cap, "learner_batch": R2_BATCH,
                       "learner_batch_per_gpu": B_LOCAL, "parallelism": "single" if world == 1 else
                       "actor-shards%d+replay-partitions+grad-allreduce" % world},
            "grad_steps_per_s": 1e3 / ms_med, "train_samples_per_s": R2_BATCH * 1e3 / ms_med,
            "buffer_add_per_s": head["adds_per_s"], "learner": "hip (csrc/learner_r2d2.hip)",
            "learner_kernel_ms_per_step": learner_ms,
            # R2D2 grad-step (B = 64) ~ 0.80 TFLOP algorithmic
            "learner_tflops": 0.80 * (B_LOCAL / 64.0) / (learner_ms * 1e-3) if learner_ms > 0 else None,
            "regions": {k: {"precision": r["precision"], "reuse": r["reuse"], "ms_per_step": r["ms_per_step"],
                            "ms_per_step_repeats": r["ms"], "env_steps_per_s": r["env_steps_per_s"],
                            "grad_steps_per_s": 1e3 / r["ms_per_step"], "clock": r["clock"], "roofline": roofline_of(r)}
                        for k, r in regions.items()},
            "kernels_ms_per_step": {k: v["total_ms"] / k_all for k, v in sorted(prof_all.items())},
            "roofline": roof,
            "roofline_hbm": None if sample_ms <= 0 else {
                "kernel": "rela_replay_sample (scan + time-major gather of 64 x 3.47 MB)", "bound": "hbm", "unit": "GB/s",
                "peak": PEAK_HBM_GBS, "achieved": sample_bytes / (sample_ms * 1e-3) / 1e9,
                "frac": sample_bytes / (sample_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "algorithmic_bytes_per_call": sample_bytes, "ms_per_call": sample_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            replay.close()
            torch.cuda.empty_cache()
            detail["cpu_baseline"] = cpu_baseline_reference_r2d2()
        line = {"metric": detail["metric"], "value": detail["value"], "unit": "env-steps/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_med, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None,
                "dtype": {"f32": "f32", "f32x3": "f32 (trunk conv2 / conv3: f32 operands as 3 exact bf16 parts on bf16 MFMA; "
                          "f32-accurate)"}.get(args.precision, "f32 results from split-bf16 MFMA (16-bit significands)"),
                "data": "synthetic",
                "config": {"workload": "R2D2 LSTM 40x80 envs/GPU (C4 shapes), seq 80 / burn-in 40 / n 3, B=64, A=18; "
                                       "device-resident static frames", "parallelism": detail["config"]["parallelism"]},
                "grad_steps_per_s": 1e3 / ms_med,
                "roofline": {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                      "avg_launch_ms", "frac_issued", "sclk_mhz") if roof.get(k) is not None
                             or k == "traffic"}}
        cb = detail.get("cpu_baseline")
        if cb is not None:
            line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                    "sample": cb["sample"][:150]}
        line["summary"] = {k: round(r["env_steps_per_s"]) for k, r in regions.items()}
        line["summary"]["grad_steps_per_s"] = {k: 1e3 / r["ms_per_step"] for k, r in regions.items()}
        emit(line, detail, "r2d2_n%d" % world)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ---- the reference's own multi-GPU layout (pyrela/main.py:131-136,155,166; BASELINE C3 / C4) ---------------------
def bench_reference_layout(args, world, rank, device, rehearsal):
    """`--layout reference`: ONE learner (rank 0, its own GPU) + world - 1 actor-only ranks, every actor rank with
    6,400 envs and a replay PARTITION (capacity / G) fed by its own actors -- rela_amd/parallel.py's exchange:
    per learner step ONE packed gather of B / G rows per partition (issued asynchronously right after the previous
    update_priority, so it lands under the backward half), one priority scatter, IS weights normalised over the
    partitions (SUM of sizes, MAX of the maximum), the flat weight buffers broadcast every 20 steps.  Command words (a
    host synchronisation on every rank) only with those publishes.  Actors tick freely, as the reference's actor
    threads do: env-steps/s = the ticks all actor ranks completed inside the timed window x 6,400 / its duration;
    the window is K learner steps on rank 0 between two publishes, at which every actor rank records its tick count."""
    import threading

    import torch.distributed as dist

    from rela_amd import _capi as capi
    from rela_amd.engine import ApexActorEngine, FFNetHandle
    from rela_amd.learner import HipApexLearner, ffnet_flat_layout, load_net_from_flat
    from rela_amd.parallel import (FFPartition, NativePartitionedReplay, NativePartitionServer, PartitionedReplay,
                                   PartitionServer, ff_batch_namespace, ff_field_specs)
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet
    from rela_amd.replay import FFReplay

    G = world - 1
    assert G >= 1 and BATCH % G == 0, "--layout reference needs >= 2 ranks and a batch that splits over the actor ranks"
    ctrl = dist.new_group(backend="gloo")  # counters / barriers of the main threads only (the exchange has its own order)
    # RCCL moves device buffers; the one-GPU gloo rehearsal exchanges through host memory
    exch = "cpu" if dist.get_backend() == "gloo" else device
    specs = ff_field_specs(NUM_ACTION)
    _, total = ffnet_flat_layout(NUM_ACTION)
    K, R = args.steps, args.repeats
    cycle = 20  # actor_sync_freq, pyrela/main.py:213-215
    # rows per partition.  Native exchange: the learner maps every partition's ring.  One hipIpcMemHandle_t per field stopped
    # working above ~24 GB per field on this pool (r4: partitions were held to 2^19 rows); since r5 the actor ranks create
    # their partition in 8 GB chunks, which travel as file descriptors at any size (include/rela_amd.h:
    # rela_replay_export_chunks; tests/test_ipc_gpu.py maps a 2^20-row partition) -- no cap
    part_cap = args.replay_cap // G
    if args.exchange == "native" and rank != 0:
        from rela_amd import _capi as capi

        capi.check(capi.lib.rela_runtime_set_replay_chunk_bytes(8 << 30), "rela_runtime_set_replay_chunk_bytes")
    torch.manual_seed(SEED + 2)
    agent = ApexAgent(lambda: AtariFFNet(NUM_ACTION), MULTI_STEP, GAMMA).to(device)
    for p_ in agent.parameters():
        dist.broadcast(p_.data, 0)
    marks = []  # actor ranks: (publish index, monotonic time, ticks so far)
    if rank == 0:
        learner = HipApexLearner.from_agent(agent, BATCH, lr=6.25e-5, eps=1.5e-4, grad_clip=40.0)
        learner.set_precision(args.precision)
        flat_on, flat_tg = learner.flat()[0], learner.flat_target()
        # native data plane (default; --exchange packed = r3's packed gather through the collective): partitions mapped
        # through HIP IPC, rows gathered by the learner's own kernel over xGMI, weights read by the actors directly
        if args.exchange == "native":
            rep = NativePartitionedReplay(specs, BATCH, BETA, exch, scheduled=True, flats=(flat_on, flat_tg), data_device=device)
        else:
            rep = PartitionedReplay(specs, BATCH, BETA, exch, scheduled=True)
        on_exch = (lambda t: t) if (exch == device or args.exchange == "native") else (lambda t: t.to(exch))
        times = []

        def run(n_steps, pending):
            """n_steps learner steps; publishes at its start and every `cycle` steps; -> pending sample"""
            done = 0
            while done < n_steps:
                seg = min(cycle, n_steps - done)
                rep.publish(on_exch(flat_on), on_exch(flat_tg), steps=seg)
                times.append(time.perf_counter())
                if pending is None:
                    pending = rep.sample(async_op=True)
                for i in range(seg):
                    fields, weight = pending.wait()
                    if exch != device:
                        fields, weight = {k: v.to(device) for k, v in fields.items()}, weight.to(device)
                    loss, prio = learner.loss(ff_batch_namespace(fields), weight)
                    rep.update_priority(prio)
                    # the next gather runs under this step's backward half (not across a publish: that needs a command word)
                    pending = rep.sample(async_op=True) if i + 1 < seg else None
                    learner.grad()
                    learner.apply()
                done += seg
            return pending

        warm = max(cycle, (args.warmup + cycle - 1) // cycle * cycle)
        pending = run(warm, None)
        windows = []
        for _ in range(R):
            torch.cuda.synchronize()
            i0, t0 = len(times), time.perf_counter()
            pending = run(K, pending)
            torch.cuda.synchronize()
            windows.append((i0, len(times), t0, time.perf_counter()))
        rep.publish(on_exch(flat_on), on_exch(flat_tg), steps=0)  # the closing mark of the last window
        times.append(time.perf_counter())
        rep.stop()
    else:
        g = rank - 1
        online, target = FFNetHandle(NUM_ACTION, device), FFNetHandle(NUM_ACTION, device)
        online.load_state_dict(agent.online_net.state_dict())
        target.load_state_dict(agent.target_net.state_dict())
        online.set_precision(args.precision)
        target.set_precision(args.precision)
        part = FFReplay(part_cap, SEED + rank, ALPHA, BETA, 0, NUM_ACTION, device)
        eps_all = generate_eps(0.4, 7, ROWS * G)
        engine = ApexActorEngine(ROWS, K_GAMES, NUM_ACTION, MULTI_STEP, GAMMA, part, eps_all[g * ROWS:(g + 1) * ROWS], device,
                                 seed=SEED + rank)
        gen = torch.Generator(device=device)
        gen.manual_seed(SEED + 7 + rank)
        for h in range(MULTI_STEP + 1):
            engine.obs_hist[h].copy_(torch.randint(0, 256, engine.obs_hist[h].shape, dtype=torch.uint8, device=device,
                                                   generator=gen))
        reward = torch.randint(-1, 2, (ROWS,), device=device, generator=gen).float()
        term = (torch.rand((ROWS,), device=device, generator=gen) < 0.005).to(torch.uint8)
        lock = threading.Lock()
        ticks, stopped = [0], [False]
        actor_stream = torch.cuda.Stream(device=device)

        def tick():
            with lock, torch.cuda.stream(actor_stream):
                engine.act(online)
                engine.post_step(reward, term, online, target, nonblocking=True)
            ticks[0] += 1

        while part.size() < BATCH:  # the partition must be able to serve before the learner asks
            tick()
        torch.cuda.synchronize()

        def on_weights(on_flat, tg_flat):
            marks.append((len(marks), time.perf_counter(), ticks[0]))
            with lock, torch.cuda.stream(server_stream):
                server_stream.wait_stream(actor_stream)
                load_net_from_flat(online, on_flat.to(device, copy=True), NUM_ACTION)
                load_net_from_flat(target, tg_flat.to(device, copy=True), NUM_ACTION)
                actor_stream.wait_stream(server_stream)

        server_stream = torch.cuda.Stream(device=device)

        failure = []

        def serve():
            try:
                torch.cuda.set_device(device)
                with torch.cuda.stream(server_stream):
                    if args.exchange == "native":
                        srv = NativePartitionServer(part, specs, BATCH, BETA, exch, on_weights=on_weights, scheduled=True,
                                                    data_device=device)
                    else:
                        srv = PartitionServer(FFPartition(part), specs, BATCH, BETA, exch, flat_sizes=(total, total),
                                              on_weights=on_weights, scheduled=True)
                    srv.serve_forever()
            except BaseException as e:  # noqa: BLE001  (re-raised on the main thread)
                failure.append(e)
            finally:
                stopped[0] = True

    # both constructors create the actor subgroup collectively: rank 0 did it in PartitionedReplay(...), the actor ranks
    # do it in PartitionServer(...) on their server threads
    if rank != 0:
        th = threading.Thread(target=serve, daemon=True)
        th.start()
        while not stopped[0]:
            tick()
            if ticks[0] % 8 == 0:
                actor_stream.synchronize()  # (bounds the launch queue; the reference's actor threads block on act() too)
        th.join()
        torch.cuda.synchronize()
        if failure:
            raise failure[0]
    # gather the actor ranks' marks on rank 0
    mine = torch.zeros(4096, 2, dtype=torch.float64)
    for i, t, c in marks:
        mine[i, 0], mine[i, 1] = t, c
    allm = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, allm, dst=0, group=ctrl)
    if rank == 0:
        ms, envs = [], []
        for i0, i1, t0, t1 in windows:
            # publish i0 opened the window, publish i1 (the first of the next run / the closing mark) ended it
            n = sum(float(allm[r][i1, 1] - allm[r][i0, 1]) for r in range(1, world)) * ROWS
            ms.append((t1 - t0) / K * 1e3)
            envs.append(n / (t1 - t0))
        med = int(np.argsort(ms)[len(ms) // 2])
        out = {
            "metric": "env-steps/s (Ape-X Atari 84x84x4, 1 learner GPU + %d actor GPUs, the reference's own layout)" % G,
            "value": envs[med], "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": ms[med], "repeats": R, "ms_per_step_repeats": ms, "env_steps_per_s_repeats": envs,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "f32x3": DTYPE_F32X3}.get(args.precision, DTYPE_NOTE), "data": "synthetic",
            "config": {"workload": "Ape-X DQN in the reference's multi-GPU layout (pyrela/main.py:131-166, BASELINE C3): ONE "
                                   "learner GPU (batch 512) + %d actor-only GPUs x 80 threads x 80 games (6400 envs each), one "
                                   "replay partition of 2^20 / %d per actor GPU, B / G rows sampled per partition; "
                                   "device-resident static frames" % (G, G),
                       "layout": "reference", "actor_gpus": G, "envs_per_actor_gpu": ROWS, "learner_batch": BATCH,
                       "replay_capacity_total": part_cap * G, "parallelism": "1 learner + %d actor shards / replay partitions" % G},
            "grad_steps_per_s": 1e3 / ms[med], "learner": "hip (csrc/learner.hip)",
            "comm": {"backend": dist.get_backend(), "rccl_ranks": world if dist.get_backend() == "nccl" else 0, "ranks": world,
                     "exchange": args.exchange,
                     "collectives_per_step": "1 packed gather (B/G rows per partition), 1 priority scatter, all-reduce SUM "
                                             "(partition size) + MAX (IS-weight maximum) among the actor ranks; 2 broadcasts "
                                             "of 6.8 MB + 1 command word every %d steps" % cycle},
        }
        line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "data", "grad_steps_per_s")}
        line["dtype"] = {"f32": "f32", "f32x3": "f32 (f32 operands as 3 exact bf16 parts on bf16 MFMA; f32-accurate)"}.get(
            args.precision, "f32 results from split-bf16 MFMA (16-bit significands)")
        line["config"] = {"workload": "Ape-X, reference layout: 1 learner GPU (B=512) + %d actor GPUs x 6400 envs, replay "
                                      "partitions of %d rows; device-resident static frames" % (G, part_cap),
                          "layout": "reference", "parallelism": out["config"]["parallelism"]}
        line["comm"] = {"backend": out["comm"]["backend"], "rccl_ranks": out["comm"]["rccl_ranks"]}
        line["env_steps_per_s_repeats"] = [round(e) for e in envs]
        emit(line, out, "apex_reference_layout_n%d" % world)
    dist.barrier(group=ctrl)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # default: 5 repeats x 300 steps of ~1.5 ms = ~2.3 s of timed work per region
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5,
                    help="how many times every K-step timed region is repeated (the line reports the median repeat)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-threaded", action="store_true", help="skip the threaded leg (rela.Context + C++ actor threads)")
    ap.add_argument("--replay-cap", type=int, default=None)
    ap.add_argument("--dedup", default=None, choices=[None, "stack", "plane"],
                    help="frame-stack de-duplication in the replay (SURVEY 8f-3): stack = 28,224 B per env-step, plane = "
                         "7,056 B (the frames of this bench are static, so only the byte traffic is representative); "
                         "default: s and next_s stored in full (56,448 B), as in round 1's headline")
    ap.add_argument("--precision", default="f32x3", choices=["f32", "f32x3", "bf16x2"],
                    help="arithmetic of the headline region (`value`): f32x3 (default) = f32 results with f32 accuracy from "
                         "the bf16 matrix cores (conv2 / conv3 / fc: every f32 operand split exactly in three bf16 parts, six "
                         "products, f32 accumulation; error against f64 no larger than the f32 MFMA kernels' and torch CPU "
                         "f32's, tests/test_ffnet_gpu.py); f32 = exact f32 MFMA for actors and learner; bf16x2 = the fast "
                         "mode, split-bf16 MFMA (two bf16 parts, three products: 16-bit significands, |dQ| < 2e-5 max|Q|). "
                         "Whichever is chosen, the others are timed as further regions of the same run (`summary`)")
    ap.add_argument("--layout", default="replicated", choices=["replicated", "reference"],
                    help="N > 1: replicated (default) = actors + replay partition + learner replica on every rank, gradient "
                         "all-reduce; reference = the reference's own layout, ONE learner rank + N - 1 actor-only ranks "
                         "with replay partitions (pyrela/main.py:131-166, BASELINE C3 / C4)")
    ap.add_argument("--allreduce", default="rccl", choices=["rccl", "ipc"],
                    help="--layout replicated, N > 1: how the learner replicas sum their gradient buckets: rccl = torch.distributed "
                         "(the default); ipc = rela_ipc_allreduce_* (include/rela_amd.h): peer reads of IPC-mapped buckets, summed "
                         "in rank order, no collective library (validated with 2-3 processes on one GPU; same host only)")
    ap.add_argument("--exchange", default="native", choices=["native", "packed"],
                    help="--layout reference: how sampled rows and weights cross processes: native = partitions and flat "
                         "buffers mapped through HIP IPC, the learner's own gather kernel reads the rows over xGMI "
                         "(rela_amd/parallel.py, r4); packed = one packed gather / two broadcasts through the collective (r3)")
    ap.add_argument("--algo", default="apex", help="apex (BASELINE.json's metric, the default) | r2d2 (config C4's "
                                                   "sequence shape: seq 80 / burn-in 40 / n 3, 3200 envs, B = 64)")
    args = ap.parse_args()
    if args.replay_cap is None:
        args.replay_cap = REPLAY_CAP if args.algo == "apex" else R2_REPLAY_CAP

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: this process becomes the launcher (it has made no GPU call) and its children the ranks
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU path")
    # Rehearsal switch for a one-GPU box: all ranks share cuda:0 and talk over gloo (RCCL refuses two
    # ranks on one device).  The driver's real N > 1 runs use one GPU per rank and RCCL.
    rehearsal = os.environ.get("RELA_BENCH_REHEARSAL", "0") == "1"
    if os.environ.get("RELA_BENCH_WATCHDOG"):  # diagnostic: every rank dumps its Python stacks and exits after N seconds
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["RELA_BENCH_WATCHDOG"]), exit=True)
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = "cuda:%d" % dev_index
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    # ONE learner batch of 512 per step for the whole job (pyrela/main.py:221): with G replay partitions every
    # rank draws B/G from its own partition, the gradients are averaged over the ranks and the
    # importance weights are normalised over all partitions.
    assert BATCH % world == 0, "the learner batch (%d) must split evenly over %d ranks" % (BATCH, world)
    B_LOCAL = BATCH // world

    from rela_amd import build as _build

    if rank == 0 or world == 1:
        _build.build_native()
    if world > 1:
        dist.barrier()
    if args.algo == "r2d2":
        return bench_r2d2(args, world, rank, device)
    if args.layout == "reference" and world > 1:
        return bench_reference_layout(args, world, rank, device, rehearsal)
    from rela_amd import _capi as capi
    from rela_amd.engine import ApexActorEngine, FFNetHandle
    from rela_amd.engine import dev_view
    from rela_amd.learner import HipApexLearner, allreduce_grads, global_is_weights, sum_grads
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet
    from rela_amd.replay import FFReplay

    torch.manual_seed(SEED + 2 + rank)
    agent = ApexAgent(lambda: AtariFFNet(NUM_ACTION), MULTI_STEP, GAMMA).to(device)
    if world > 1:  # identical replicas
        for p in agent.parameters():
            dist.broadcast(p.data, 0)
    optim = torch.optim.RMSprop(agent.online_net.parameters(), lr=6.25e-5, eps=1.5e-4)
    # learner step: hand-written HIP (csrc/learner.hip) by default; RELA_BENCH_LEARNER=torch runs the
    # same step through PyTorch autograd (pyrela/main.py:226-239 verbatim) for comparison
    LEARNER = os.environ.get("RELA_BENCH_LEARNER", "hip")
    hip_learner = HipApexLearner.from_agent(agent, B_LOCAL, lr=6.25e-5, eps=1.5e-4, grad_clip=40.0) \
        if LEARNER == "hip" else None
    online, target = FFNetHandle(NUM_ACTION, device), FFNetHandle(NUM_ACTION, device)
    online.load_state_dict(agent.online_net.state_dict())
    target.load_state_dict(agent.target_net.state_dict())
    online.set_precision(args.precision)
    target.set_precision(args.precision)
    if hip_learner is not None:  # the two gradient-free forwards of td_err; the pass feeding the backward stays f32
        hip_learner.set_precision(args.precision)
        if world > 1 and args.allreduce == "ipc":  # gradient buckets mapped into every rank; no collective library
            from rela_amd.parallel import IpcAllReduce

            hip_learner.ipc_allreduce = IpcAllReduce(hip_learner.flat()[1])

    replay = FFReplay(args.replay_cap, SEED + rank, ALPHA, BETA, 0, NUM_ACTION, device, dedup=args.dedup,
                      guard_units=(MULTI_STEP + 8) * ROWS)
    eps_all = generate_eps(0.4, 7, ROWS * world)
    eps = eps_all[rank * ROWS:(rank + 1) * ROWS]
    engine = ApexActorEngine(ROWS, K_GAMES, NUM_ACTION, MULTI_STEP, GAMMA, replay, eps, device, seed=SEED + rank)

    # synthetic inputs resident in HBM: the observation history ring doubles as the frame pool
    g = torch.Generator(device=device)
    g.manual_seed(SEED + 7 + rank)
    for h in range(MULTI_STEP + 1):
        engine.obs_hist[h].copy_(torch.randint(0, 256, engine.obs_hist[h].shape, dtype=torch.uint8, device=device,
                                               generator=g))
    n_pool = 16
    reward_pool = torch.randint(-1, 2, (n_pool, ROWS), device=device, generator=g).float()
    term_pool = (torch.rand((n_pool, ROWS), device=device, generator=g) < 0.005).to(torch.uint8)

    step_idx = [0]
    # Actors and learner run concurrently in the reference (C++ actor threads next to the Python
    # learner loop, pyrela/main.py:185-251).  Here the actor tick goes to its own HIP stream and
    # the learner step stays on torch's default stream; they meet only through the replay, whose
    # private stream serialises insert / sample / update in commit order.
    actor_stream = torch.cuda.Stream(device=device)
    # The learner's launches go to their own stream.  With the eager PyTorch learner of round 1 (~170 short kernels) a
    # HIGH-priority stream kept them from queueing behind the tick's chip-filling kernels; with the HIP learner (a few
    # dozen launches, two lanes) normal priority is ~1 % faster (interleaved on one box: 5.36-5.37 M against 5.31-5.32 M
    # env-steps/s; the legacy default stream 5.36-5.40).  RELA_BENCH_PRIO=1: high priority, 0: torch's default stream.
    prio = os.environ.get("RELA_BENCH_PRIO", "2")
    if prio == "1":
        main_stream = torch.cuda.Stream(device=device, priority=-1)
    elif prio == "2":
        main_stream = torch.cuda.Stream(device=device)
    else:
        main_stream = torch.cuda.current_stream(device)

    if os.environ.get("RELA_BENCH_ONE_STREAM", "0") == "1":  # diagnosis: actor tick and learner step strictly one after the
        actor_stream = main_stream                           # other (the replay keeps its own streams)

    def actor_tick():
        i = step_idx[0] % n_pool
        with torch.cuda.stream(actor_stream):
            engine.act(online)
            engine.post_step(reward_pool[i], term_pool[i], online, target, nonblocking=True)

    # The HIP learner's step is software-pipelined with the replay (RELA_BENCH_PIPELINE=0: strictly one after the
    # other): priorities are final after the forward half (rela_apex_learner_loss), so update_priority and the NEXT
    # sample are queued before the backward half (rela_apex_learner_grad) and the replay's latency-bound sample chain
    # runs next to the gradient kernels.  Same calls, same order on the replay's stream, same results; the batch
    # buffers alternate between two slots because conv1's weight gradient still reads the current batch.
    PIPE = hip_learner is not None and os.environ.get("RELA_BENCH_PIPELINE", "1") == "1"
    sample_slot = [0]
    pending = [None]
    if PIPE:
        replay.set_deferred_wait(True)

    def learner_housekeeping():
        k = step_idx[0]
        if k % 2500 == 0:
            if hip_learner is not None:
                hip_learner.sync_target_with_online()
            else:
                agent.sync_target_with_online()
        if k % 20 == 0:  # ModelLocker.update_model, main.py:213-215 (waits for in-flight actor work)
            main_stream.wait_stream(actor_stream)
            if hip_learner is not None:
                hip_learner.publish(online, target)
            else:
                online.load_state_dict(agent.online_net.state_dict())
                target.load_state_dict(agent.target_net.state_dict())
            actor_stream.wait_stream(main_stream)

    def learner_sample():
        batch, weight = replay.sample(B_LOCAL, slot=sample_slot[0])
        sample_slot[0] ^= 1 if PIPE else 0
        return batch, weight, replay.size()

    def global_weights(weight, size_at_sample):
        if world == 1:
            return weight
        # one replay partition per GPU: normalise the IS weights over all of them (the raw weights and the
        # partition's sum are the last sample's; nothing else samples in between)
        raw_p, sum_p = C.c_void_p(), C.c_void_p()
        capi.check(capi.lib.rela_replay_last_sample_dev(replay.h, C.byref(raw_p), C.byref(sum_p)), "last_sample")
        raw_w = dev_view(raw_p.value, (B_LOCAL,), torch.float32, torch.device(device))
        part_sum = dev_view(sum_p.value, (1,), torch.float32, torch.device(device))
        return global_is_weights(raw_w, part_sum, size_at_sample, BETA)

    def learner_update(batch, weight):
        if hip_learner is not None:
            loss, prio = hip_learner.step(batch, weight, world_size=world)
            replay.update_priority(prio)
            return
        loss, prio = agent.loss(batch, sync_priority=False)
        (loss * weight).mean().backward()
        if world > 1:
            allreduce_grads(agent.online_net.parameters(), world)
        torch.nn.utils.clip_grad_norm_(agent.online_net.parameters(), 40.0)
        optim.step()
        optim.zero_grad(set_to_none=True)
        replay.update_priority(prio)

    # diagnosis only (RELA_BENCH_ONLY=actor|learner): time one side alone; the JSON line says so
    ONLY = os.environ.get("RELA_BENCH_ONLY", "")

    class _HostLaps:
        """RELA_BENCH_HOSTTIME=1: where the HOST's time per step goes (the calls only queue work; none waits for the GPU
        by design, so a long lap is launch overhead or a hidden wait)"""

        def __init__(self, on):
            self.on, self.t, self.acc, self.n = on, 0.0, {}, 0

        def lap(self, name):
            if not self.on:
                return
            now = time.perf_counter()
            if name is None:
                self.n += 1
            else:
                self.acc[name] = self.acc.get(name, 0.0) + now - self.t
            self.t = now

        def report(self):
            if self.on and self.n:
                print("[bench] host us per step: " + ", ".join("%s %.0f" % (k, v / self.n * 1e6) for k, v in self.acc.items()),
                      file=sys.stderr, flush=True)
                self.acc, self.n = {}, 0

    HOST_LAPS = _HostLaps(os.environ.get("RELA_BENCH_HOSTTIME") == "1")
    # RELA_BENCH_JOIN=0 (diagnosis): the learner's stream does not wait for the tick at the end of every step (the publish
    # every 20 steps still orders the two)
    JOIN = os.environ.get("RELA_BENCH_JOIN", "1") == "1"

    def one_step():
        # One step = one learner update + one actor tick, sharing the GPU.  Order of the host calls:
        #  1. sample      -- a handful of launches; sees the replay as the previous tick left it (had
        #                    the tick been queued first, `sample` would sit behind the tick's `add` on
        #                    the replay's in-order stream)
        #  2. actor tick  -- a dozen launches on actor_stream, ~3 ms of GPU work
        #  3. loss / backward / optimiser / update_priority -- ~150 eager PyTorch launches whose
        #                    host-side cost now overlaps the tick's GPU work instead of preceding it
        lap = HOST_LAPS.lap
        with torch.cuda.stream(main_stream):
            lap(None)
            if ONLY != "actor":
                learner_housekeeping()
                if pending[0] is None:
                    pending[0] = learner_sample()
                batch, weight, size_at_sample = pending[0]
                pending[0] = None
                lap("housekeeping")
            if ONLY != "learner":
                actor_tick()
                lap("actor_tick")
            if ONLY != "actor":
                if PIPE:
                    replay.wait()  # the batch sampled during the previous step's backward half
                weight = global_weights(weight, size_at_sample)
                if PIPE:
                    loss, prio = hip_learner.loss(batch, weight)
                    lap("loss")
                    replay.update_priority(prio)
                    lap("update_priority")
                    pending[0] = learner_sample()
                    lap("sample")
                    hip_learner.grad()
                    lap("grad")
                    if world > 1:
                        sum_grads(hip_learner, world)
                    hip_learner.apply()
                    lap("apply")
                else:
                    learner_update(batch, weight)
            if JOIN:
                main_stream.wait_stream(actor_stream)

    # fill the ring to capacity (untimed): real ticks for the history, then bulk inserts
    for _ in range(MULTI_STEP + 1):
        actor_tick()
        step_idx[0] += 1
    prio = torch.empty(ROWS, device=device)
    z_a = torch.zeros(ROWS, dtype=torch.int64, device=device)
    z_f = torch.zeros(ROWS, device=device)
    z_b = torch.zeros(ROWS, dtype=torch.uint8, device=device)
    while args.dedup and replay.size() + ROWS <= args.replay_cap:  # de-duplicated rows enter through the shard
        actor_tick()
        step_idx[0] += 1
    while replay.size() + ROWS <= args.replay_cap:
        prio.uniform_(0.01, 2.0, generator=g)
        obs_a, obs_b = engine.obs_hist[0], engine.obs_hist[1]
        ptrs = [obs_a.data_ptr(), obs_b.data_ptr(), engine.eps.data_ptr(), engine.eps.data_ptr(),
                engine.legal.data_ptr(), engine.legal.data_ptr(), z_a.data_ptr(), z_f.data_ptr(), z_b.data_ptr(),
                z_f.data_ptr()]
        replay.add_rows(ROWS, ptrs, prio)
    torch.cuda.synchronize()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def run_steps(k):
        for _ in range(k):
            one_step()
            step_idx[0] += 1

    def bracketed(k):
        """EXACTLY k steps between barrier + synchronize on both sides; MAX over ranks; -> seconds"""
        sync_all()
        t0 = time.perf_counter()
        run_steps(k)
        t_issue = time.perf_counter() - t0
        sync_all()
        dt = time.perf_counter() - t0
        if os.environ.get("RELA_BENCH_HOSTTIME") == "1":  # diagnosis: is the host (launches) or the GPU the longer pole?
            print("[bench] %d steps: host issued them in %.3f ms per step, done after %.3f ms per step" % (
                k, t_issue / k * 1e3, dt / k * 1e3), file=sys.stderr, flush=True)
            HOST_LAPS.report()
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    run_steps(args.warmup)
    sync_all()
    # Live roofline: HIP events around the four heavy forward kernels only in the exact-f32 regions (the dominant kernel is one
    # of them); timing all ~100 kernels of a step costs 0.37 ms of the step itself, so the full
    # per-kernel table comes from a short untimed pass after the timed region.
    ROOF_FILTER = b"conv12_fused,conv1_bf16x3,conv2_mfma,conv3_mfma,fc_mfma"
    NOPROF = os.environ.get("RELA_BENCH_NOPROF") == "1"
    buf = C.create_string_buffer(1 << 16)

    def prof_summary():
        capi.check(capi.lib.rela_prof_summary_json(buf, len(buf)), "rela_prof_summary_json")
        return json.loads(buf.value.decode())

    cur_prec = [args.precision]

    def set_all_precision(mode):
        if mode == cur_prec[0]:  # (setting a mode bumps the nets' weight version: memoised forwards would be recomputed)
            return
        cur_prec[0] = mode
        online.set_precision(mode)
        target.set_precision(mode)
        if hip_learner is not None:
            hip_learner.set_precision(mode)

    try:
        pr = torch.cuda.get_device_properties(dev_index)
        pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:  # noqa: BLE001
        pci = None

    settle = min(args.warmup, 3) + MULTI_STEP + 1  # (+ n + 1 ticks so every history slot is of the region's mode)

    def timed_region(precision, reuse, settle_steps):
        """One timed region: `settle_steps` untimed steps in the region's mode, then the K-step bracket (barrier +
        synchronize on both sides, MAX over ranks) repeated args.repeats times back to back, with live HIP events
        around the heavy forward kernels and the shader clock sampled next to it."""
        set_all_precision(precision)
        engine.set_reuse(reuse)
        run_steps(settle_steps)
        sync_all()
        # f32x3 / bf16x2: conv1 -> conv2 fused is the dominant kernel by a wide margin (0.26 ms against 0.11 / 0.13 ms), and
        # every timed launch costs the step ~5 us of serialisation around its two events: time that kernel alone (all
        # four cost the f32x3 step 2.4 %, 2.081 -> 2.032 ms with no events at all; one box).  The exact-f32 regions keep all four
        # (conv2, conv3 and fc are within 30 % of each other there).
        capi.lib.rela_prof_set_filter(b"conv12_fused" if precision in ("f32x3", "bf16x2") else ROOF_FILTER)
        capi.lib.rela_prof_enable(0 if NOPROF else 1)
        add0 = replay.num_add()
        with ClockSampler(pci) as clk:
            ms = [bracketed(args.steps) / args.steps * 1e3 for _ in range(args.repeats)]
        capi.lib.rela_prof_enable(0)
        adds = replay.num_add() - add0
        if world > 1:
            a = torch.tensor([adds], device=device, dtype=torch.float64)
            dist.all_reduce(a, op=dist.ReduceOp.SUM)
            adds = float(a.item())
        prof = prof_summary()
        fwd_cnt = prof.get("conv12_fused", prof.get("conv1_bf16x3", {"count": 0}))["count"]
        return {"precision": precision, "reuse": reuse, "ms": ms, "ms_per_step": float(np.median(ms)), "prof": prof,
                "adds_per_s": adds / (sum(ms) * 1e-3 * args.steps), "clock": clk.summary(),
                "env_steps_per_s": ROWS * world / (float(np.median(ms)) * 1e-3),
                "forwards_per_tick": fwd_cnt / (args.steps * args.repeats) if ONLY != "learner" else 0}

    head = timed_region(args.precision, 1, 0)

    # untimed: the same step with every kernel timed, for the kernels_ms_per_step table
    capi.lib.rela_prof_set_filter(None)
    capi.lib.rela_prof_enable(1)
    k_all = max(5, min(100, args.steps * args.repeats // 3))
    run_steps(k_all)
    sync_all()
    capi.lib.rela_prof_enable(0)
    prof_all = prof_summary()

    # More timed regions of the same K steps x repeats, bracketed the same way, so that ONE run carries the four
    # accountings (the headline is f32_mode or fast_mode, whichever --precision names):
    #   f32_mode        the reference's arithmetic (exact f32 MFMA for actors and learner), act()'s forwards memoised
    #   strict          the reference's arithmetic AND the reference's work: f32 + all 4 trunk forwards per env-step
    #   fast_mode       split-bf16 MFMA (stated tolerance), forwards memoised
    #   fast_no_reuse   split-bf16 MFMA, 4 forwards (SURVEY 8d's unit of work)
    #   f32x3_mode      f32 operands as three bf16 parts on the bf16 MFMA (f32 accuracy, csrc/gemm_f32emu.h), memoised
    #   f32x3_strict    the same arithmetic, all 4 trunk forwards
    name_of = {("f32", 1): "f32_mode", ("f32", 0): "strict", ("bf16x2", 1): "fast_mode", ("bf16x2", 0): "fast_no_reuse",
               ("f32x3", 1): "f32x3_mode", ("f32x3", 0): "f32x3_strict"}
    regions = {name_of[(args.precision, 1)]: head}
    if not ONLY:
        order = [(args.precision, 0)] + [(q, r) for q in ("f32x3", "f32", "bf16x2") if q != args.precision for r in (1, 0)]
        for prec, reuse in order:
            regions[name_of[(prec, reuse)]] = timed_region(prec, reuse, settle)
        set_all_precision(args.precision)
        engine.set_reuse(1)
    st = replay.debug_state()
    assert st["dev_error"] == 0, "replay reported device error %d" % st["dev_error"]

    def roofline_of(region):
        """Both rooflines of the region's dominant kernel (largest total time among the timed forward kernels);
        `bound` = the one it sits closer to.  MFMA: algorithmic FLOPs (2 * MACs) per launch over (i) the dense
        peak of the instruction class the kernel issues -- v_mfma_f32_16x16x4_f32 157.3 TFLOP/s, bf16 2,500 TFLOP/s --
        = `frac_algorithmic`, and for the split-bf16 kernels also over (ii) that peak divided by the bf16 products
        ISSUED per algorithmic product = `frac_issued` (MFMA issue utilisation, what PMC SQ_VALU_MFMA_BUSY sees).
        HBM: algorithmic bytes (input + output tensors, 4 B per activation) over 8 TB/s."""
        prof = region["prof"] or {"(profiling off)": {"total_ms": 0.0, "count": 1}}
        name, rec = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        avg_ms = rec["total_ms"] / max(rec["count"], 1)
        roof = {"kernel": name, "bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None,
                "traffic": None, "avg_launch_ms": avg_ms, "launches": rec["count"]}
        if name in FLOP and avg_ms > 0:
            fast = region["precision"] == "bf16x2"
            flops = FLOP[name] * ROWS
            # bf16 MFMA products issued per algorithmic product: conv1 multiplies exact u8 inputs by weights split
            # in 3 bf16 pieces (f32 mode: exact) or 2 (bf16x2 mode); the other layers 3 in bf16x2 mode, f32 MFMA else
            products = (2 if fast else 3) if name == "conv1_bf16x3" else (3 if fast else 0)
            x3 = region["precision"] == "f32x3" and name in ("conv12_fused", "conv3_mfma", "fc_mfma")
            if x3:
                products = 6  # both operands in three bf16 parts, the six products with i + j <= 2
            conv1_i8 = True  # (the half-frame bf16 conv1 -> conv2 kernel was removed in r4)
            if name == "conv12_fused":
                # conv2: 3 (bf16x2) or 6 (f32x3) bf16 products per product.  conv1: 3 int8 digit products (conv12_i8 /
                # conv12_s3) on v_mfma_i32_16x16x64_i8, which runs at twice the bf16 rate = 1.5 bf16-MFMA-equivalents
                f1, f2 = FLOP["conv1_bf16x3"], FLOP["conv2_mfma"]
                products = ((1.5 if conv1_i8 else 2) * f1 + (6 if x3 else 3) * f2) / (f1 + f2)
            peak_class = PEAK_BF16_MFMA_TFLOPS if products else PEAK_F32_MFMA_TFLOPS
            if x3:
                # an f32 product of this mode IS six bf16 MFMA products (conv1 -> conv2 fused: 4.85 on average): the dense
                # peak for f32 results is the bf16 peak / products; `frac` is therefore the MFMA ISSUE utilisation (what PMC
                # SQ_VALU_MFMA_BUSY sees); the algorithmic fraction of the bf16 peak and the fraction of the f32 MFMA's
                # own peak (157.3) ride next to it
                peak_class = PEAK_BF16_MFMA_TFLOPS / products
                issued_products, products = products, 1
                f32x3_note = True
            else:
                f32x3_note = False
            ach = flops / (avg_ms * 1e-3) / 1e12
            mfma = {"achieved": ach, "peak": peak_class, "unit": "TFLOP/s", "frac_algorithmic": ach / peak_class,
                    "frac_issued": ach * products / peak_class if products else ach / peak_class,
                    "products_per_product": products or 1, "algorithmic_flop_per_launch": flops,
                    "instruction": ("v_mfma_f32_16x16x32_bf16 x%.3g (split operands%s)" % (
                        products, "; conv1: three int8 digit products on v_mfma_i32_16x16x64_i8, counted as 1.5 bf16 products"
                        if name == "conv12_fused" and conv1_i8 else "")) if products else "v_mfma_f32_16x16x4_f32"}
            if f32x3_note:
                mfma["instruction"] = ("v_mfma_f32_16x16x32_bf16, %.3g bf16-MFMA products per f32 product (both operands as three "
                                       "exact bf16 parts%s): peak = 2,500 / %.3g TFLOP/s" % (
                                           issued_products, "; conv1: three int8 digit products counted as 1.5"
                                           if name == "conv12_fused" else "", issued_products))
                mfma["frac_of_f32_mfma_peak"] = ach / PEAK_F32_MFMA_TFLOPS
                mfma["frac_algorithmic_bf16"] = ach / PEAK_BF16_MFMA_TFLOPS
                mfma["frac_issued"] = ach / peak_class
                mfma["products_per_product"] = issued_products
            nbytes = (BYTES_PER_SAMPLE_F32X3 if x3 else BYTES_PER_SAMPLE)[name] * ROWS
            hbm = {"achieved": nbytes / (avg_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                   "algorithmic_bytes_per_launch": nbytes}
            hbm["frac"] = hbm["achieved"] / PEAK_HBM_GBS
            if mfma["frac_algorithmic"] >= hbm["frac"]:
                roof.update(bound="mfma", achieved=ach, peak=peak_class, unit="TFLOP/s", frac=mfma["frac_algorithmic"])
            else:
                roof.update(bound="hbm", achieved=hbm["achieved"], frac=hbm["frac"])
            roof["mfma"], roof["hbm"] = mfma, hbm
        roof["traffic"], roof["traffic_source"] = traffic_from_profiles(name, region["precision"])
        roof.update(region["clock"])
        return roof

    if rank == 0:
        ms_med = head["ms_per_step"]
        roof = roofline_of(head)
        prof = prof_all  # (the untimed pass right after the headline region, same mode, every kernel timed)
        fwd_ms = sum(prof[k]["total_ms"] for k in FLOP if k in prof)
        fwd_cnt = prof.get("conv12_fused", prof.get("conv1_bf16x3", {"count": 1}))["count"]
        scan_ms = sum(v["total_ms"] for k, v in prof_all.items() if k.startswith("seq_") or k in (
            "replay_targets", "replay_search", "replay_pop", "replay_is_weights"))
        notes = {"f32_mode": "actor nets and learner in the exact f32 MFMA mode (the reference's arithmetic); act()'s forwards "
                             "memoised (bit-identical to recomputing them, tests/test_agent_ops_gpu.py)",
                 "strict": "the reference's arithmetic AND the reference's work: exact f32 MFMA mode for actors and learner, "
                           "all 4 trunk forwards per env-step (SURVEY 8d: 74.8 MFLOP)",
                 "fast_mode": "split-bf16 MFMA (|dQ| < 2e-5 max|Q|, tests/test_ffnet_gpu.py), forwards memoised",
                 "fast_no_reuse": "split-bf16 MFMA, all 4 trunk forwards per env-step",
                 "f32x3_mode": "f32 accuracy on the bf16 matrix cores: conv2 / conv3 / fc with every f32 operand split exactly "
                               "in three bf16 parts and six products each (error against f64 no larger than the exact-f32-MFMA "
                               "mode's and torch-CPU-f32's, tests/test_ffnet_gpu.py); learner in exact f32 MFMA; forwards memoised",
                 "f32x3_strict": "the same arithmetic, all 4 trunk forwards per env-step"}
        detail_regions = {k: {"precision": r["precision"], "forwards_per_tick": r["forwards_per_tick"], "steps": args.steps,
                              "repeats": len(r["ms"]), "ms_per_step": r["ms_per_step"], "ms_per_step_repeats": r["ms"],
                              "env_steps_per_s": r["env_steps_per_s"], "grad_steps_per_s": 1e3 / r["ms_per_step"],
                              "clock": r["clock"], "roofline": roofline_of(r), "note": notes[k]}
                          for k, r in regions.items()}
        workload = ("Ape-X DQN, 80 threads x 80 games (6400 envs) per GPU, actor+learner on one MI355X, replay 2^20 per GPU "
                    "device-resident, A=18, n=3, ONE learner batch of 512 per step for the whole job (B/G sampled per replay "
                    "partition); device-resident static frames: no env stepping and no H2D inside the timed region")
        dtype = {"f32": "f32", "f32x3": DTYPE_F32X3}.get(args.precision, DTYPE_NOTE)
        comm = None if world == 1 else {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size()
                                        if dist.get_backend() == "nccl" else 0, "ranks": dist.get_world_size(),
                                        "gradient_allreduce": args.allreduce,
                                        "collectives_per_step": ("all-reduce SUM of the flat 6.8 MB gradient buffer"
                                                                 if args.allreduce == "rccl" else
                                                                 "gradient buckets summed by peer reads of IPC-mapped buffers "
                                                                 "(rela_ipc_allreduce_run, no collective)")
                                        + "; IS weights: SUM of the partition size + MAX of the weight maximum"}
        detail = {
            "metric": "env-steps/s (Ape-X Atari 84x84x4, actor tick + learner grad-step)",
            "value": head["env_steps_per_s"], "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_med, "repeats": args.repeats, "ms_per_step_repeats": head["ms"],
            "ms_per_step_note": "the K-step region (barrier + synchronize on both sides, MAX over ranks) is timed "
                                "`repeats` times back to back; ms_per_step and value are the MEDIAN repeat",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload, "envs_per_gpu": ROWS, "replay_capacity": args.replay_cap, "learner_batch": BATCH,
                       "learner_batch_per_gpu": B_LOCAL, "replay_dedup": args.dedup, "actor_precision": args.precision,
                       "learner_precision": LEARNER_PRECISION_NOTE[args.precision] if hip_learner is not None
                       else "f32 (PyTorch autograd)",
                       "replay_frame_bytes_per_transition": {None: 56448, "stack": 28224, "plane": 7056}[args.dedup],
                       "parallelism": ("actor-shards%d+replay-partitions+grad-allreduce" % world) if world > 1
                       else "single"},
            "comm": comm,
            "grad_steps_per_s": 1e3 / ms_med, "train_samples_per_s": BATCH * 1e3 / ms_med,
            "buffer_add_per_s": head["adds_per_s"],
            "learner": "hip (csrc/learner.hip)" if hip_learner is not None else "torch autograd",
            **({"diagnostic_only": ONLY} if ONLY else {}),
            # act + compute_priority's target(next_obs); online(next_obs) and online(obs) are act's own forwards
            # of this tick and of n ticks ago (same weights, same batch), reused bit-identically -> 2, else 3-4
            "forwards_per_tick": head["forwards_per_tick"],
            "regions": detail_regions,
            "forward_ms_per_6400": fwd_ms / max(fwd_cnt, 1),
            "forward_tflops": sum(v for k, v in FLOP.items() if k != "conv12_fused") * ROWS
            / (max(fwd_ms, 1e-9) / max(fwd_cnt, 1) * 1e-3) / 1e12,
            "replay_sample_scan_ms": scan_ms / k_all,
            "kernels_ms_per_step": {k: v["total_ms"] / k_all for k, v in sorted(prof_all.items())},
            "kernels_ms_per_step_note": "untimed pass of %d steps with every kernel timed; the timed regions time "
                                        "only conv1/conv2/conv3/fc (roofline)" % k_all,
            "roofline": roof,
            # the HBM-bound side of the path.  (i) the sample call, SURVEY 8d: 4*N + B*56,448 algorithmic bytes
            # (the reference's linear scan over the N live weights + the gather of the batch rows) over the
            # summed launch time of ALL kernels of one rela_replay_sample; (ii) the insert's row copies.
            "roofline_hbm": (lambda ms, nbytes: None if ms <= 0 else {
                "kernel": "rela_replay_sample (seq_* + replay_targets/search/pop/is_weights/gather_*)", "bound": "hbm",
                "unit": "GB/s", "peak": PEAK_HBM_GBS, "achieved": nbytes / (ms * 1e-3) / 1e9,
                "frac": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "algorithmic_bytes_per_call": nbytes,
                "ms_per_call": ms, "scan_ms": scan_ms / k_all})(
                    (scan_ms + sum(v["total_ms"] for k, v in prof_all.items() if k.startswith("replay_gather"))) / k_all,
                    4 * st["safe_size"] + B_LOCAL * 56448),
            "roofline_hbm_insert": (lambda ms: None if ms <= 0 else {
                "kernel": "replay_scatter_rows", "bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS,
                "achieved": ROWS * 2 * 28224 * 2 / (ms * 1e-3) / 1e9,
                "frac": ROWS * 2 * 28224 * 2 / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "algorithmic_bytes_per_step": ROWS * 2 * 28224 * 2, "ms_per_step": ms})(
                    prof_all.get("replay_scatter_rows", {"total_ms": 0.0})["total_ms"] / k_all),
        }
        if world == 1 and not ONLY and not args.no_threaded:
            # the drop-in's own metric; the big device buffers of this process go first (the child owns a replay too)
            replay.close()
            torch.cuda.empty_cache()
            detail["threaded"] = threaded_leg(precision=args.precision)
        if world == 1 and not args.no_cpu_baseline:
            port = cpu_baseline()
            ref = cpu_baseline_reference()
            detail["cpu_baseline"] = ref if ref is not None else port
            if ref is not None:
                detail["cpu_baseline"]["port"] = {k: port[k] for k in ("value", "cores", "sample")}

        # ---- the compact line: everything the driver must see, inside its 2,000-character tail ----
        def short_roof(r):
            m = r.get("mfma")
            out = {"kernel": r["kernel"], "bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
                   "frac": r["frac"], "traffic": r["traffic"], "avg_launch_ms": r["avg_launch_ms"]}
            if m is not None and m["products_per_product"] != 1:
                out["frac_issued"] = m["frac_issued"]  # MFMA issue utilisation (bf16 products issued per product counted)
            if m is not None and "frac_of_f32_mfma_peak" in m:  # f32x3: the three readings of the same launch, side by side
                out["frac_algorithmic_bf16"] = m["frac_algorithmic_bf16"]
                out["frac_of_f32_mfma_peak"] = m["frac_of_f32_mfma_peak"]
            if "sclk_mhz" in r:
                out["sclk_mhz"] = r["sclk_mhz"]
            return out
        summary = {k: round(r["env_steps_per_s"]) for k, r in regions.items()}
        if "fast_mode" in regions and args.precision == "f32":
            fr = roofline_of(regions["fast_mode"])
            summary["fast_roofline"] = {"kernel": fr["kernel"], "frac": fr["frac"],
                                        "frac_issued": (fr.get("mfma") or {}).get("frac_issued"),
                                        "avg_launch_ms": fr["avg_launch_ms"]}
        th = detail.get("threaded")
        if th is not None:  # [without sampler, with sampler] per env flavour
            for env in ("fresh", "sliding"):
                e = th.get(env, {})
                summary["threaded_" + env] = e.get("error", [round(e.get("without_sampler", 0)), round(e.get("with_sampler", 0))])
        line = {"metric": detail["metric"], "value": detail["value"], "unit": "env-steps/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_med, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None,
                "dtype": {"f32": "f32", "f32x3": "f32 (f32 operands as 3 exact bf16 parts on bf16 MFMA; f32-accurate)"}.get(
                    args.precision, "f32 results from split-bf16 MFMA (16-bit significands)"),
                "data": "synthetic",
                "config": {"workload": "Ape-X DQN 80x80 envs/GPU, actor+learner on one MI355X, replay 2^20, B=512, A=18, n=3; "
                                       "device-resident static frames", "forwards_per_tick": round(head["forwards_per_tick"], 2),
                           "parallelism": detail["config"]["parallelism"]},
                "grad_steps_per_s": 1e3 / ms_med, "roofline": short_roof(roof)}
        if comm is not None:
            line["comm"] = {"backend": comm["backend"], "rccl_ranks": comm["rccl_ranks"]}
        cb = detail.get("cpu_baseline")
        if cb is not None:
            line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                    "sample": cb["sample"][:150]}
        if ONLY:
            line["diagnostic_only"] = ONLY
        line["summary"] = summary
        emit(line, detail, "apex_n%d" % world)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
