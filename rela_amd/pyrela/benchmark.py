"""Actor-rate benchmark of the threaded drop-in (counterpart of pyrela/benchmark.py:19-126).

T C++ actor threads x K synthetic envs run BasicThreadLoop -> DQNActor (HIP) -> FFPrioritizedReplay
(HBM).  The act rate (sum of num_act() deltas / s, pyrela/utils.py:40-44) is measured over
`num_epoch` windows without sampling and `num_epoch` windows with a concurrent B = 512
sample / update_priority loop; the mean of the last half of each is reported, as upstream.
Unlike bench.py this path includes host env stepping and the per-step host->HBM observation upload.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from rela_amd.pyrela import create_env, utils  # noqa: E402
from rela_amd.pyrela.apex import ApexAgent  # noqa: E402
from rela_amd.pyrela.net import AtariFFNet, AtariLSTMNet  # noqa: E402
from rela_amd.pyrela.r2d2 import R2D2Agent  # noqa: E402

rela = create_env.rela


def benchmark_fps(num_thread, num_game_per_thread, args):
    num_action = create_env.get_num_action("synthetic")
    eps = utils.generate_eps(0.4, 7, num_thread * num_game_per_thread)
    if args.algo == "r2d2":  # not in the reference's benchmark.py; same protocol on the R2D2 classes
        agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, num_action), "cpu", 3, 0.997, 0.9, args.seq_len, args.seq_burn_in, 0)
        locker = rela.ModelLocker([agent], args.device)
        replay_buffer = rela.RNNPrioritizedReplay(args.replay_buffer_size, args.seed, 0.9, 0.6, 0)
        make_actor = lambda i: rela.R2D2Actor(locker, 3, num_game_per_thread, 0.997, args.seq_len, args.seq_burn_in,
                                              replay_buffer)
    else:
        agent = ApexAgent(lambda: AtariFFNet(num_action), 3, 0.99).to(args.device)
        locker = rela.ModelLocker([agent], args.device)
        replay_buffer = rela.FFPrioritizedReplay(args.replay_buffer_size, args.seed, 0.6, 0.4, 0)
        make_actor = lambda i: rela.DQNActor(locker, 1, num_game_per_thread, 0.99, replay_buffer)
    batch = 64 if args.algo == "r2d2" else 512
    context, games, actors = create_env.create_train_env(args.seed, eps, args.episode_len, num_thread,
                                                         num_game_per_thread, make_actor)
    context.start()
    while replay_buffer.size() < args.burn_in_frames:
        time.sleep(0.2)
    seen = utils.total_acts(actors)
    rates = {"without": [], "with": []}
    ring = int(1.25 * args.replay_buffer_size)
    rows = num_thread * num_game_per_thread
    for mode in ("without", "with"):
        for epoch in range(args.num_epoch):
            t0 = time.time()
            n_sample = 0
            if mode == "without":
                # Nobody evicts in this mode (sampling does, prioritized_replay.h:311-315), so the actors would park on
                # back-pressure once the ring is full: at ~2 M env-steps/s a 2^22 replay fills within three
                # seconds, where the reference's CPU actors never got there in 6 x 30 s.  One sample / update_priority
                # pair is therefore issued whenever the ring is about to fill (it evicts down to capacity: about two
                # per second, against ~1,000 per second in the other mode) and reported as `sample rate`.
                while time.time() - t0 < args.epoch_sec:
                    if replay_buffer.size() + 8 * rows >= ring:
                        _, weight = replay_buffer.sample(batch, args.device)
                        replay_buffer.update_priority(weight)
                        n_sample += 1
                    time.sleep(0.005)
            else:
                while time.time() - t0 <= args.epoch_sec:
                    _, weight = replay_buffer.sample(batch, args.device)
                    replay_buffer.update_priority(weight)
                    n_sample += 1
                torch.cuda.synchronize()  # the loop only enqueues: count what the GPU finished
            dt = time.time() - t0
            now = utils.total_acts(actors)
            rates[mode].append((now - seen) / dt)
            seen = now
            print("%s sample: epoch %d, act rate: %d, buffer size: %d, sample rate: %d/s" % (
                mode, epoch, rates[mode][-1], replay_buffer.size(), n_sample / dt), flush=True)
            if os.environ.get("RELA_THREADED_STATS") == "1":
                st = rela.threaded_stats(True)
                print("   thread-us per env-step: " + ", ".join("%s %.2f" % (k[:-3], v) for k, v in st.items() if k.endswith("_us")),
                      flush=True)
    context.terminate()
    context.resume()
    t_drain = time.time()
    while not context.terminated():
        if time.time() - t_drain > 120:
            raise RuntimeError("actor threads did not terminate within 120 s (replay size %d)" % replay_buffer.size())
        if replay_buffer.size() >= batch:  # unpark actors blocked on a full ring
            _, weight = replay_buffer.sample(batch, args.device)
            replay_buffer.update_priority(weight)
        time.sleep(0.01)
    half = args.num_epoch // 2
    tail = lambda xs: float(np.mean(xs[-max(1, min(half, len(xs))):])) if xs else float("nan")
    return tail(rates["without"]), tail(rates["with"])


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--device", default="cuda:0")
    p.add_argument("--algo", default="apex", help="apex (the reference's benchmark) | r2d2")
    p.add_argument("--seq_len", type=int, default=80)
    p.add_argument("--seq_burn_in", type=int, default=40)
    p.add_argument("--seed", type=int, default=10001)
    p.add_argument("--replay_buffer_size", type=int, default=2 ** 21)
    p.add_argument("--burn_in_frames", type=int, default=1000)
    p.add_argument("--episode_len", type=int, default=200)
    p.add_argument("--epoch_sec", type=float, default=30)
    p.add_argument("--num_epoch", type=int, default=6)
    p.add_argument("--grid", default="80x20,80x40,80x80,80x160", help="threads x games per thread")
    p.add_argument("--env", default=None, choices=[None, "fresh", "sliding", "null"],
                   help="synthetic env flavour: fresh = four new LCG planes per step (SURVEY 8d, the default), sliding = "
                        "ONE new plane per step, Atari's frame stacking (atari/game_state.h:53-82), null = constant frames "
                        "and zero host cost (the runtime's own ceiling); default: what RELA_SYNTH_SLIDING / RELA_SYNTH_ENV say")
    args = p.parse_args(argv)
    if args.env is not None:
        os.environ["RELA_SYNTH_SLIDING"] = "1" if args.env == "sliding" else "0"
        os.environ["RELA_SYNTH_ENV"] = "null" if args.env == "null" else ""
    rows = []
    for cell in args.grid.split(","):
        t, k = (int(v) for v in cell.split("x"))
        without, with_ = benchmark_fps(t, k, args)
        rows.append((t, k, without, with_))
        print("act rate: without sample: %.2f, with sample: %.2f" % (without, with_), flush=True)
    print("%8s %14s %24s %24s" % ("#thread", "#game/thread", "act rate (w/o sample)", "act rate (with sample)"))
    for r in rows:
        print("%8d %14d %24.1f %24.1f" % r)
    return rows


if __name__ == "__main__":
    main()
