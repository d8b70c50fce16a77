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

`--algo r2d2` does the same for the R2D2 classes (R2D2Actor / RNNPrioritizedReplay, rela/r2d2_actor.h:189-353) of
oracle/_ref/h6/rela*.so -- the reference compiled from a scratch copy with SURVEY H6's one-line fix, without which its
R2D2 path cannot sample under torch >= 1.5 -- with an AtariLSTMNet-shaped TorchScript agent written here from the
published arithmetic (pyrela/r2d2.py:58-120, pyrela/net.py:58-163).

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
from typing import Dict, Tuple  # noqa: E402


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


class LSTMNet(nn.Module):
    def __init__(self, num_action: int):
        super().__init__()
        self.net = nn.Sequential(nn.Conv2d(4, 32, 8, stride=4), nn.ReLU(), nn.Conv2d(32, 64, 4, stride=2), nn.ReLU(),
                                 nn.Conv2d(64, 64, 3, stride=1), nn.ReLU())
        self.lstm = nn.LSTM(3136, 512, num_layers=1)
        self.fc_v = nn.Linear(512, 1)
        self.fc_a = nn.Linear(512, num_action)

    @torch.jit.export
    def step(self, obs: Dict[str, torch.Tensor], hid: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor,
                                                                                         Dict[str, torch.Tensor]]:
        """one LSTM step on a [K, ...] batch -> (v [K,1], legal-masked advantages [K,A], new hidden)"""
        x = self.net(obs["s"].float() / 255.0).flatten(1).unsqueeze(0)
        o, (h, c) = self.lstm(x, (hid["h0"], hid["c0"]))
        o = o.squeeze(0)
        return self.fc_v(o), self.fc_a(o), {"h0": h, "c0": c}


class R2D2Agent(nn.Module):
    def __init__(self, num_action: int, multi_step: int, gamma: float, eta: float, burn_in: int):
        super().__init__()
        self.online_net = LSTMNet(num_action)
        self.target_net = LSTMNet(num_action)
        self.multi_step = multi_step
        self.gamma = gamma
        self.eta = eta
        self.burn_in = burn_in

    @torch.jit.export
    def get_h0(self, batchsize: int) -> Dict[str, torch.Tensor]:
        return {"h0": torch.zeros(1, batchsize, 512), "c0": torch.zeros(1, batchsize, 512)}

    def greedy(self, obs: Dict[str, torch.Tensor], hid: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        v, a, new_hid = self.online_net.step(obs, hid)
        return ((1 + a - a.min()) * obs["legal_move"]).argmax(1), new_hid

    @torch.jit.export
    def act(self, obs: Dict[str, torch.Tensor], hid: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor],
                                                                                        Dict[str, torch.Tensor]]:
        greedy, new_hid = self.greedy(obs, hid)
        eps = obs["eps"].squeeze(1)
        explore = obs["legal_move"].multinomial(1).squeeze(1)
        coin = (torch.rand(greedy.size(0), device=greedy.device) < eps).long()
        return {"a": (greedy * (1 - coin) + explore * coin).long().detach().cpu()}, \
               {"h0": new_hid["h0"].detach(), "c0": new_hid["c0"].detach()}

    @torch.jit.export
    def compute_priority(self, obs: Dict[str, torch.Tensor], action: Dict[str, torch.Tensor], reward: torch.Tensor,
                         terminal: torch.Tensor, bootstrap: torch.Tensor, next_obs: Dict[str, torch.Tensor],
                         hid: Dict[str, torch.Tensor], next_hid: Dict[str, torch.Tensor]) -> torch.Tensor:
        v, a, _ = self.online_net.step(obs, hid)
        la = a * obs["legal_move"]
        qa = (v + la - la.mean(1, keepdim=True)).gather(1, action["a"].unsqueeze(1)).squeeze(1)
        next_a, _ = self.greedy(next_obs, next_hid)
        tv, ta, _ = self.target_net.step(next_obs, next_hid)
        tla = ta * next_obs["legal_move"]
        boot = (tv + tla - tla.mean(1, keepdim=True)).gather(1, next_a.unsqueeze(1)).squeeze(1)
        target = reward + bootstrap * (self.gamma ** self.multi_step) * boot
        return (target.detach() - qa).detach().abs().cpu()

    @torch.jit.export
    def aggregate_priority(self, priority: torch.Tensor, seq_len: torch.Tensor) -> torch.Tensor:
        t = torch.arange(priority.size(1), device=seq_len.device)
        masked = priority * (t.unsqueeze(0) < seq_len.unsqueeze(1)).float()
        mean = masked.sum(1) / (seq_len - self.burn_in)
        return (self.eta * masked.max(1)[0] + (1.0 - self.eta) * mean).detach().cpu()

    def forward(self, obs: Dict[str, torch.Tensor], hid: Dict[str, torch.Tensor]) -> torch.Tensor:
        return self.online_net.step(obs, hid)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--games", type=int, default=20)
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--warmup", type=float, default=4.0)
    ap.add_argument("--num_action", type=int, default=18)
    ap.add_argument("--capacity", type=int, default=None)
    ap.add_argument("--algo", default="apex", help="apex | r2d2 (oracle/_ref/h6: the reference with SURVEY H6's fix)")
    ap.add_argument("--seq_len", type=int, default=80)
    ap.add_argument("--burn_in", type=int, default=40)
    args = ap.parse_args()
    r2d2 = args.algo == "r2d2"
    if args.capacity is None:  # apex: x 1.25 x 56 KB of frames = 2.3 GB at most; r2d2: x 1.25 x 3.47 MB = 1.1 GB
        args.capacity = 256 if r2d2 else 1 << 15
    ref = os.path.join(HERE, "_ref", "h6") if r2d2 else os.path.join(HERE, "_ref")
    if not glob.glob(os.path.join(ref, "rela*.so")):
        raise SystemExit("oracle/_ref/rela*.so is missing (built by `make -C oracle ref` where /root/reference exists)")
    sys.path.insert(0, os.path.join(HERE, "_ref"))  # synth_atari (registers against whichever `rela` is imported first)
    sys.path.insert(0, ref)
    torch.set_num_threads(1)
    import rela  # the reference's module
    import synth_atari

    assert os.path.dirname(os.path.abspath(rela.__file__)) == ref
    torch.manual_seed(1)
    if r2d2:
        agent = torch.jit.script(R2D2Agent(args.num_action, 3, 0.997, 0.9, args.burn_in))
        locker = rela.ModelLocker([agent], "cpu")
        replay = rela.RNNPrioritizedReplay(args.capacity, 10002, 0.9, 0.6, 0)
    else:
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
        actor = rela.R2D2Actor(locker, 3, args.games, 0.997, args.seq_len, args.burn_in, replay) if r2d2 \
            else rela.DQNActor(locker, 3, args.games, 0.997, replay)
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
                batch, weight = replay.sample(8 if r2d2 else 32, "cpu")
                replay.update_priority(weight)
                n_sample += 1
            time.sleep(0.05)

    wait(args.warmup)
    a0, t0 = total(), time.time()
    wait(args.seconds)
    a1, t1 = total(), time.time()
    print(json.dumps({"env_steps_per_s": (a1 - a0) / (t1 - t0), "threads": args.threads, "games": args.games,
                      "seconds": t1 - t0, "buffer_size": replay.size(), "num_action": args.num_action,
                      "samples": n_sample, "algo": args.algo}), flush=True)
    # leave without joining the actor threads: they may be mid-forward; the process ends here
    os._exit(0)


if __name__ == "__main__":
    main()
