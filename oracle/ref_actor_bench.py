#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (cpu_baseline leg of bench.py) -- never imported by the product path.

Times the REAL reference's CPU-thread actor path: the `rela` extension compiled from /root/reference
by oracle/Makefile (oracle/_ref/rela*.so, prebuilt; the reference sources never travel) runs its own
Context / BasicThreadLoop / DQNActor / FFPrioritizedReplay / ModelLocker C++ code with T actor threads
x K synthetic envs (oracle/_ref/synth_atari*.so = this repo's env compiled against the reference's
rela/env.h) on the host cores.  The TorchScript agent it calls by method name (`act`,
`compute_priority`, rela/dqn_actor.h:161,199) is written here from the published Ape-X arithmetic
(pyrela/apex.py:30-78, pyrela/net.py:8-55 for the shapes) -- no reference Python is needed at run time.
Protocol of pyrela/benchmark.py:55-109: warm up, then a fixed window of d(sum num_act) / dt, with a
light sampler (B = 32 every 50 ms; sampling evicts the overflow, prioritized_replay.h:311-315) so that
the ring never fills and its memory stays bounded.  OMP_NUM_THREADS=1 as the reference recommends
(README.md:125-130).

  python oracle/ref_actor_bench.py --threads 16 --games 20 --seconds 12   -> one JSON line
"""
import argparse
import glob
import json
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
HERE = os.path.dirname(os.path.abspath(__file__))

import torch  # noqa: E402
from torch import nn  # noqa: E402
from typing import Dict  # noqa: E402


class FFNet(nn.Module):
    def __init__(self, num_action: int):
        super().__init__()
        self.net = nn.Sequential(nn.Conv2d(4, 32, 8, stride=4), nn.ReLU(), nn.Conv2d(32, 64, 4, stride=2), nn.ReLU(),
                                 nn.Conv2d(64, 64, 3, stride=1), nn.ReLU())
        self.linear = nn.Sequential(nn.Linear(3136, 512), nn.ReLU())
        self.fc_v = nn.Linear(512, 1)
        self.fc_a = nn.Linear(512, num_action)

    def forward(self, obs: Dict[str, torch.Tensor]) -> torch.Tensor:
        x = self.net(obs["s"].float() / 255.0)
        h = self.linear(x.flatten(1))
        a = self.fc_a(h) * obs["legal_move"]
        return self.fc_v(h) + a - a.mean(1, keepdim=True)


class Agent(nn.Module):
    def __init__(self, num_action: int, multi_step: int, gamma: float):
        super().__init__()
        self.online_net = FFNet(num_action)
        self.target_net = FFNet(num_action)
        self.multi_step = multi_step
        self.gamma = gamma

    def greedy(self, obs: Dict[str, torch.Tensor]) -> torch.Tensor:
        q = self.online_net(obs).detach()
        return ((1 + q - q.min()) * obs["legal_move"]).argmax(1)

    @torch.jit.export
    def act(self, obs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        greedy = self.greedy(obs)
        eps = obs["eps"].squeeze(1)
        explore = obs["legal_move"].multinomial(1).squeeze(1)
        coin = (torch.rand(greedy.size(0), device=greedy.device) < eps).long()
        return {"a": (greedy * (1 - coin) + explore * coin).long().detach().cpu()}

    @torch.jit.export
    def compute_priority(self, obs: Dict[str, torch.Tensor], action: Dict[str, torch.Tensor], reward: torch.Tensor,
                         terminal: torch.Tensor, bootstrap: torch.Tensor,
                         next_obs: Dict[str, torch.Tensor]) -> torch.Tensor:
        qa = self.online_net(obs).gather(1, action["a"].unsqueeze(1)).squeeze(1)
        next_a = self.greedy(next_obs)
        boot = self.target_net(next_obs).gather(1, next_a.unsqueeze(1)).squeeze(1)
        target = reward + bootstrap * (self.gamma ** self.multi_step) * boot
        return (target.detach() - qa).detach().abs().cpu()

    def forward(self, obs: Dict[str, torch.Tensor]) -> torch.Tensor:
        return self.online_net(obs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--games", type=int, default=20)
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--warmup", type=float, default=4.0)
    ap.add_argument("--num_action", type=int, default=18)
    ap.add_argument("--capacity", type=int, default=1 << 15)  # x 1.25 x 56 KB of frames = 2.3 GB at most
    args = ap.parse_args()
    ref = os.path.join(HERE, "_ref")
    if not glob.glob(os.path.join(ref, "rela*.so")):
        raise SystemExit("oracle/_ref/rela*.so is missing (built by `make -C oracle ref` where /root/reference exists)")
    sys.path.insert(0, ref)
    torch.set_num_threads(1)
    import rela  # the reference's module
    import synth_atari

    assert os.path.dirname(os.path.abspath(rela.__file__)) == ref
    torch.manual_seed(1)
    agent = torch.jit.script(Agent(args.num_action, 3, 0.997))
    locker = rela.ModelLocker([agent], "cpu")
    replay = rela.FFPrioritizedReplay(args.capacity, 10002, 0.6, 0.4, 0)
    n = args.threads * args.games
    eps = [0.4 ** (1 + i / max(1, n - 1) * 7) for i in range(n)]  # generate_eps(0.4, 7, n), pyrela/utils.py
    ctx = rela.Context()
    actors, games = [], []
    for t in range(args.threads):
        vec = rela.VectorEnv()
        for g in range(args.games):
            i = t * args.games + g
            game = synth_atari.SyntheticAtariEnv(10002 + i, eps[i], args.num_action, 200)
            games.append(game)
            vec.append(game)
        actor = rela.DQNActor(locker, 3, args.games, 0.997, replay)
        actors.append(actor)
        ctx.push_env_thread(rela.BasicThreadLoop(actor, vec, False))
    ctx.start()
    total = lambda: sum(a.num_act() for a in actors)
    n_sample = 0

    def wait(seconds):
        nonlocal n_sample
        t_end = time.time() + seconds
        while time.time() < t_end:
            if replay.size() > args.capacity:  # evict the overflow so that no actor parks on a full ring
                batch, weight = replay.sample(32, "cpu")
                replay.update_priority(weight)
                n_sample += 1
            time.sleep(0.05)

    wait(args.warmup)
    a0, t0 = total(), time.time()
    wait(args.seconds)
    a1, t1 = total(), time.time()
    print(json.dumps({"env_steps_per_s": (a1 - a0) / (t1 - t0), "threads": args.threads, "games": args.games,
                      "seconds": t1 - t0, "buffer_size": replay.size(), "num_action": args.num_action,
                      "samples": n_sample}), flush=True)
    # leave without joining the actor threads: they may be mid-forward; the process ends here
    os._exit(0)


if __name__ == "__main__":
    main()
