#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs oracle/_ref/*, built by `make -C oracle ref` from
/root/reference).  The outputs (small JSON files: inputs + the reference's outputs) are
committed; this script and the harness sources are committed too, the reference is not.

  replay_*.json : rela::PrioritizedReplay<FFTransition> driven by oracle/ref_harness/replay_kat.cc
  nstep_*.json  : rela::MultiStepTransitionBuffer driven by oracle/ref_harness/nstep_kat.cc
  ffnet_*.json  : pyrela/net.py AtariFFNet + pyrela/apex.py ApexAgent imported from
                  /root/reference/pyrela (Q-values, greedy action, priority, loss)
"""
import json
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from synth import f32_to_hex, synth_obs, synth_params, synth_r2d2_batch  # noqa: E402  (tests/synth.py)
ROOT = os.path.dirname(os.path.dirname(HERE))
REFBIN = os.path.join(ROOT, "oracle", "_ref")


def f2h(x):
    return struct.pack(">f", float(np.float32(x))).hex()


def run(binary, script):
    out = subprocess.run(
        [os.path.join(REFBIN, binary)], input="\n".join(script) + "\n", capture_output=True, text=True, check=True
    )
    return [json.loads(l) for l in out.stdout.splitlines() if l.strip()]


def save(name, script, expect, **meta):
    path = os.path.join(HERE, name + ".json")
    with open(path, "w") as f:
        json.dump({"name": name, "script": script, "expect": expect, **meta}, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


# ------------------------------------------------------------------ replay cases
def replay_kat(name, cap, seed, alpha, beta, batch, rounds):
    """capacity 32, priorities arange(32), update ones*(round+2)."""
    s = ["new %d %d %s %s 0" % (cap, seed, f2h(alpha), f2h(beta))]
    s.append("add %d 0 %s" % (cap, " ".join(f2h(i) for i in range(cap))))
    for r in range(rounds):
        s.append("sample %d" % batch)
        s.append("update %d %s" % (batch, " ".join(f2h(r + 2) for _ in range(batch))))
    save(name, s, run("replay_kat", s))


def replay_random(name, cap, seed, alpha, beta, batch, block, n_blocks, sample_every, prio_fn, rng_seed):
    rng = np.random.default_rng(rng_seed)
    ring = int(1.25 * cap)
    s = ["new %d %d %s %s 0" % (cap, seed, f2h(alpha), f2h(beta))]
    tag = 0
    size = 0
    for b in range(n_blocks):
        if size + block > ring:
            # the reference would block here; sample to evict
            s.append("sample %d" % batch)
            s.append("update %d %s" % (batch, " ".join(f2h(p) for p in prio_fn(rng, batch))))
            size = min(size, cap)
        s.append("add %d %d %s" % (block, tag, " ".join(f2h(p) for p in prio_fn(rng, block))))
        tag += block
        size += block
        if (b + 1) % sample_every == 0 and size >= batch:
            s.append("sample %d" % batch)
            s.append("update %d %s" % (batch, " ".join(f2h(p) for p in prio_fn(rng, batch))))
            size = min(size, cap)
    save(name, s, run("replay_kat", s))


def p_uniform(rng, n):
    return rng.uniform(0.01, 2.0, n).astype(np.float32)


def p_loguniform(rng, n):
    return np.exp(rng.uniform(np.log(1e-6), np.log(1e3), n)).astype(np.float32)


def p_sparse(rng, n):
    p = rng.uniform(0.0, 1.0, n).astype(np.float32)
    p[rng.uniform(size=n) < 0.6] = 0.0
    return p


def p_spiky(rng, n):
    p = rng.uniform(0.001, 0.01, n).astype(np.float32)
    p[rng.uniform(size=n) < 0.05] = 1000.0
    return p


def replay_cases():
    replay_kat("replay_kat_a1_b1_seed42", 32, 42, 1.0, 1.0, 8, 3)
    replay_kat("replay_kat_a06_b04_seed7", 32, 7, 0.6, 0.4, 8, 3)
    # wrap-around + eviction + updates of evicted ids
    replay_random("replay_wrap_evict_a1", 64, 10002, 1.0, 0.4, 8, 16, 40, 2, p_uniform, 1)
    replay_random("replay_wrap_evict_a06", 64, 10002, 0.6, 0.4, 8, 16, 40, 2, p_uniform, 2)
    # zero priorities (leading zeros, acc > 0 guard)
    replay_random("replay_zeros_a1", 128, 5, 1.0, 1.0, 16, 32, 12, 3, p_sparse, 3)
    # duplicates: a few huge weights take several strata each
    replay_random("replay_spiky_a1", 256, 77, 1.0, 0.4, 32, 64, 12, 2, p_spiky, 4)
    # wide dynamic range: exercises f64 rounding in the sequential accumulator
    replay_random("replay_lograng_a1", 2048, 123, 1.0, 0.4, 64, 256, 14, 4, p_loguniform, 5)
    replay_random("replay_lograng_a06", 2048, 321, 0.6, 0.4, 64, 256, 14, 4, p_loguniform, 6)
    # tiny sum: sum - 0.2 < 0 clamps every target below zero (first positive weight wins)
    s = ["new 16 9 %s %s 0" % (f2h(1.0), f2h(1.0))]
    s.append("add 8 0 " + " ".join(f2h(v) for v in [0, 0, 0.01, 0.02, 0, 0.03, 0.01, 0.02]))
    s.append("sample 4")
    s.append("update 4 " + " ".join(f2h(v) for v in [0.5, 0.25, 0.125, 0.0625]))
    s.append("sample 4")
    s.append("update 4 " + " ".join(f2h(v) for v in [1, 1, 1, 1]))
    s.append("sample 4")
    save("replay_tiny_sum_clamp", s, run("replay_kat", s))


def replay_cases_r2():
    """alpha = 0.9 (R2D2's priority exponent, ref_run_r2d2.sh) and block sizes that mix ATen's vectorised and
    scalar-tail pow inside one call (40 = 32 + 8; batches of 24 are all tail)."""
    replay_kat("replay_kat_a09_b06_seed5", 32, 5, 0.9, 0.6, 8, 3)
    replay_random("replay_wrap_evict_a09", 64, 777, 0.9, 0.6, 8, 16, 40, 2, p_uniform, 7)
    replay_random("replay_lograng_a09", 2048, 99, 0.9, 0.6, 64, 256, 14, 4, p_loguniform, 8)
    replay_random("replay_mixed_tail_a09", 512, 31, 0.9, 0.4, 24, 40, 30, 3, p_uniform, 9)
    replay_random("replay_mixed_tail_a06", 512, 32, 0.6, 0.4, 40, 72, 20, 2, p_loguniform, 10)


# ------------------------------------------------------------------ n-step cases
def nstep_cases():
    for n, gamma, seed in [(1, 0.99, 11), (3, 0.997, 12), (5, 0.997, 13), (3, 0.5, 14)]:
        rng = np.random.default_rng(seed)
        K = 6
        s = ["new %d %d %s" % (n, K, f2h(gamma))]
        for _ in range(48):
            if seed % 2:
                r = rng.integers(-1, 2, K).astype(np.float32)
            else:
                r = rng.normal(0, 1, K).astype(np.float32)
            t = (rng.uniform(size=K) < 0.15).astype(int)
            s.append("step %s %s" % (" ".join(f2h(v) for v in r), " ".join(str(v) for v in t)))
        save("nstep_n%d_seed%d" % (n, seed), s, run("nstep_kat", s), multi_step=n, K=K, gamma=f2h(gamma))


# ------------------------------------------------------------------ network cases
def ffnet_cases():
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference/pyrela")
    import torch
    from apex import ApexAgent  # noqa: the reference's own modules
    from net import AtariFFNet

    torch.set_num_threads(4)
    for name, A, N, legal_mode in [("ffnet_A18_N5", 18, 5, "ones"), ("ffnet_A6_N3_masked", 6, 3, "mask")]:
        agent = ApexAgent(lambda: AtariFFNet(A), 3, 0.997)
        on, tg = synth_params(A, 1001), synth_params(A, 2002)
        agent.online_net.load_state_dict({k: torch.from_numpy(v) for k, v in on.items()})
        agent.target_net.load_state_dict({k: torch.from_numpy(v) for k, v in tg.items()})
        s, ns = synth_obs(N, 31), synth_obs(N, 32)
        rng = np.random.default_rng(33)
        legal = np.ones((N, A), np.float32)
        if legal_mode == "mask":
            legal[:, 1::2] = 0.0
        action = rng.integers(0, A, N)
        if legal_mode == "mask":
            action = (action // 2) * 2
        reward = rng.integers(-1, 2, N).astype(np.float32)
        bootstrap = (rng.uniform(size=N) < 0.8).astype(np.float32)
        obs = {"s": torch.from_numpy(s), "legal_move": torch.from_numpy(legal), "eps": torch.zeros(N, 1)}
        nobs = {"s": torch.from_numpy(ns), "legal_move": torch.from_numpy(legal), "eps": torch.zeros(N, 1)}
        with torch.no_grad():
            q = agent.online_net(obs)
            qn = agent.online_net(nobs)
            qt = agent.target_net(nobs)
            greedy = agent.greedy_act(obs)
            act = agent.act(obs)["a"]  # eps == 0 -> greedy branch
            prio = agent.compute_priority(
                obs, {"a": torch.from_numpy(action)}, torch.from_numpy(reward), torch.zeros(N, dtype=torch.bool),
                torch.from_numpy(bootstrap), nobs)
            err = agent.td_err(obs, {"a": torch.from_numpy(action)}, torch.from_numpy(reward),
                               torch.from_numpy(bootstrap), nobs)
        save(name, [], [], num_action=A, N=N, legal_mode=legal_mode, online_seed=1001, target_seed=2002,
             obs_seed=31, next_obs_seed=32, misc_seed=33, multi_step=3, gamma=0.997,
             action=action.tolist(), reward=reward.tolist(), bootstrap=bootstrap.tolist(),
             q=q.numpy().astype(np.float64).tolist(), q_next_online=qn.numpy().astype(np.float64).tolist(),
             q_next_target=qt.numpy().astype(np.float64).tolist(), greedy=greedy.tolist(), act_eps0=act.tolist(),
             priority=prio.numpy().astype(np.float64).tolist(), td_err=err.numpy().astype(np.float64).tolist())


def learner_cases():
    """One learner step of the REAL reference on CPU (pyrela/main.py:226-239 with pyrela/apex.py):
    loss -> (loss * weight).mean().backward() -> clip_grad_norm_ -> RMSprop.step.  Gradients and
    updated parameters are recorded as per-tensor norms / sums plus 48 sampled entries."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference/pyrela")
    import types

    import torch
    from apex import ApexAgent  # noqa: the reference's own modules
    from net import AtariFFNet

    torch.set_num_threads(4)
    for name, A, B, clip in [("learner_apex_A18_B32", 18, 32, 0.5), ("learner_apex_A6_B8_noclip", 6, 8, 40.0)]:
        multi_step, gamma, lr, eps = 3, 0.997, 1e-3, 1.5e-4
        agent = ApexAgent(lambda: AtariFFNet(A), multi_step, gamma)
        on, tg = synth_params(A, 3003), synth_params(A, 4004)
        agent.online_net.load_state_dict({k: torch.from_numpy(v) for k, v in on.items()})
        agent.target_net.load_state_dict({k: torch.from_numpy(v) for k, v in tg.items()})
        s, ns = synth_obs(B, 41), synth_obs(B, 42)
        rng = np.random.default_rng(43)
        legal = (rng.uniform(size=(B, A)) < 0.8).astype(np.float32)
        nlegal = (rng.uniform(size=(B, A)) < 0.8).astype(np.float32)
        legal[:, 0] = 1.0
        nlegal[:, 1] = 1.0
        action = np.array([rng.choice(np.flatnonzero(legal[i])) for i in range(B)], np.int64)
        reward = rng.normal(0.0, 1.5, B).astype(np.float32)  # some |td| > 1: both Huber branches
        bootstrap = (rng.uniform(size=B) < 0.8).astype(np.float32)
        weight = rng.uniform(0.1, 1.0, B).astype(np.float32)
        batch = types.SimpleNamespace(
            obs={"s": torch.from_numpy(s), "legal_move": torch.from_numpy(legal), "eps": torch.zeros(B, 1)},
            next_obs={"s": torch.from_numpy(ns), "legal_move": torch.from_numpy(nlegal), "eps": torch.zeros(B, 1)},
            action={"a": torch.from_numpy(action)}, reward=torch.from_numpy(reward),
            terminal=torch.zeros(B, dtype=torch.bool), bootstrap=torch.from_numpy(bootstrap))
        params = list(agent.online_net.parameters())
        optim = torch.optim.RMSprop(params, lr=lr, eps=eps)  # main.py:120
        loss, priority = agent.loss(batch)                    # main.py:226
        loss = (loss * torch.from_numpy(weight)).mean()       # :228
        loss.backward()                                       # :229
        named = dict(agent.online_net.named_parameters())
        pick = {k: np.random.default_rng(7).integers(0, v.numel(), 48) for k, v in named.items()}

        def digest(get):
            out = {}
            for k, v in named.items():
                t = get(v).detach().double().reshape(-1)
                out[k] = {"l2": float(t.norm()), "sum": float(t.sum()), "absmax": float(t.abs().max()),
                          "idx": pick[k].tolist(), "val": t[torch.from_numpy(pick[k])].tolist()}
            return out

        grads = digest(lambda v: v.grad)
        g_norm = torch.nn.utils.clip_grad_norm_(params, clip)  # :233
        optim.step()                                           # :238
        after = digest(lambda v: v)
        save(name, [], [], num_action=A, B=B, multi_step=multi_step, gamma=gamma, lr=lr, eps=eps, grad_clip=clip,
             online_seed=3003, target_seed=4004, obs_seed=41, next_obs_seed=42,
             legal=legal.tolist(), next_legal=nlegal.tolist(), action=action.tolist(), reward=reward.tolist(),
             bootstrap=bootstrap.tolist(), weight=weight.tolist(), loss=float(loss),
             priority=priority.numpy().astype(np.float64).tolist(), grad_norm=float(g_norm), grads=grads,
             params_after=after)



def ffnet_big_cases():
    """Batches large enough to reach the split-bf16 kernels (ffnet.hip: trunk from 128 rows, fc from 1,024), recorded
    from the REAL reference's net.py / apex.py on the CPU; Q tables stored bit-exactly as hex strings of their
    float32 bytes.  `*_q50`: every weight tensor times 4.6 (|Q| of 30-60, a trained agent's scale)."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference/pyrela")
    import torch
    from apex import ApexAgent  # noqa: the reference's own modules
    from net import AtariFFNet

    torch.set_num_threads(8)
    for name, A, N, gain in [("ffnet_A18_N1024", 18, 1024, 1.0), ("ffnet_A18_N256_q50", 18, 256, 4.6)]:
        agent = ApexAgent(lambda: AtariFFNet(A), 3, 0.997)
        on, tg = synth_params(A, 1001, gain), synth_params(A, 2002, gain)
        agent.online_net.load_state_dict({k: torch.from_numpy(v) for k, v in on.items()})
        agent.target_net.load_state_dict({k: torch.from_numpy(v) for k, v in tg.items()})
        s, ns = synth_obs(N, 131), synth_obs(N, 132)
        rng = np.random.default_rng(133)
        legal = (rng.uniform(size=(N, A)) < 0.8).astype(np.float32)
        legal[:, 0] = 1.0
        action = (rng.uniform(size=(N, A)) * legal).argmax(1)
        reward = rng.integers(-1, 2, N).astype(np.float32)
        bootstrap = (rng.uniform(size=N) < 0.8).astype(np.float32)
        obs = {"s": torch.from_numpy(s), "legal_move": torch.from_numpy(legal), "eps": torch.zeros(N, 1)}
        nobs = {"s": torch.from_numpy(ns), "legal_move": torch.from_numpy(legal), "eps": torch.zeros(N, 1)}
        with torch.no_grad():
            q = agent.online_net(obs)
            qn = agent.online_net(nobs)
            qt = agent.target_net(nobs)
            greedy = agent.greedy_act(obs)
            prio = agent.compute_priority(
                obs, {"a": torch.from_numpy(action)}, torch.from_numpy(reward), torch.zeros(N, dtype=torch.bool),
                torch.from_numpy(bootstrap), nobs)
            err = agent.td_err(obs, {"a": torch.from_numpy(action)}, torch.from_numpy(reward),
                               torch.from_numpy(bootstrap), nobs)
        save(name, [], [], num_action=A, N=N, gain=gain, legal_mode="random(133) < 0.8, column 0 legal", online_seed=1001,
             target_seed=2002, obs_seed=131, next_obs_seed=132, misc_seed=133, multi_step=3, gamma=0.997,
             q_absmax=float(q.abs().max()), q_hex=f32_to_hex(q.numpy()), q_next_online_hex=f32_to_hex(qn.numpy()),
             q_next_target_hex=f32_to_hex(qt.numpy()), greedy=greedy.tolist(), priority_hex=f32_to_hex(prio.numpy()),
             td_err_hex=f32_to_hex(err.numpy()))


def learner_big_cases(only=None):
    """learner_cases at B = 128 (the learner's split-bf16 kernels start at 128 rows) and, r5, at B = 512 (the learner
    batch of pyrela/main.py:28 and the batch size from which the f32x3 mode's three-part kernels run), inputs
    re-derived from seeds by the test (legal / action / reward / bootstrap / weight: rng `misc` in this order)."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference/pyrela")
    import types

    import torch
    from apex import ApexAgent  # noqa: the reference's own modules
    from net import AtariFFNet

    torch.set_num_threads(8)
    for name, A, B, clip, seeds in [("learner_apex_A18_B128", 18, 128, 0.5, (3003, 4004, 141, 142, 143)),
                                    ("learner_apex_A18_B512", 18, 512, 0.5, (3013, 4014, 151, 152, 153))]:
        if only and name != only:
            continue
        s_on, s_tg, s_obs, s_nobs, s_misc = seeds
        multi_step, gamma, lr, eps = 3, 0.997, 1e-3, 1.5e-4
        agent = ApexAgent(lambda: AtariFFNet(A), multi_step, gamma)
        on, tg = synth_params(A, s_on), synth_params(A, s_tg)
        agent.online_net.load_state_dict({k: torch.from_numpy(v) for k, v in on.items()})
        agent.target_net.load_state_dict({k: torch.from_numpy(v) for k, v in tg.items()})
        s, ns = synth_obs(B, s_obs), synth_obs(B, s_nobs)
        rng = np.random.default_rng(s_misc)
        legal = (rng.uniform(size=(B, A)) < 0.8).astype(np.float32)
        nlegal = (rng.uniform(size=(B, A)) < 0.8).astype(np.float32)
        legal[:, 0] = 1.0
        nlegal[:, 1] = 1.0
        action = np.array([rng.choice(np.flatnonzero(legal[i])) for i in range(B)], np.int64)
        reward = rng.normal(0.0, 1.5, B).astype(np.float32)
        bootstrap = (rng.uniform(size=B) < 0.8).astype(np.float32)
        weight = rng.uniform(0.1, 1.0, B).astype(np.float32)
        batch = types.SimpleNamespace(
            obs={"s": torch.from_numpy(s), "legal_move": torch.from_numpy(legal), "eps": torch.zeros(B, 1)},
            next_obs={"s": torch.from_numpy(ns), "legal_move": torch.from_numpy(nlegal), "eps": torch.zeros(B, 1)},
            action={"a": torch.from_numpy(action)}, reward=torch.from_numpy(reward),
            terminal=torch.zeros(B, dtype=torch.bool), bootstrap=torch.from_numpy(bootstrap))
        params = list(agent.online_net.parameters())
        optim = torch.optim.RMSprop(params, lr=lr, eps=eps)
        loss, priority = agent.loss(batch)
        loss = (loss * torch.from_numpy(weight)).mean()
        loss.backward()
        named = dict(agent.online_net.named_parameters())
        pick = {k: np.random.default_rng(7).integers(0, v.numel(), 48) for k, v in named.items()}

        def digest(get):
            out = {}
            for k, v in named.items():
                t = get(v).detach().double().reshape(-1)
                out[k] = {"l2": float(t.norm()), "sum": float(t.sum()), "absmax": float(t.abs().max()),
                          "idx": pick[k].tolist(), "val": t[torch.from_numpy(pick[k])].tolist()}
            return out

        grads = digest(lambda v: v.grad)
        g_norm = torch.nn.utils.clip_grad_norm_(params, clip)
        optim.step()
        after = digest(lambda v: v)
        save(name, [], [], num_action=A, B=B, multi_step=multi_step, gamma=gamma, lr=lr, eps=eps, grad_clip=clip,
             online_seed=s_on, target_seed=s_tg, obs_seed=s_obs, next_obs_seed=s_nobs,
             legal=legal.tolist(), next_legal=nlegal.tolist(), action=action.tolist(), reward=reward.tolist(),
             bootstrap=bootstrap.tolist(), weight=weight.tolist(), loss=float(loss),
             priority=priority.numpy().astype(np.float64).tolist(), grad_norm=float(g_norm), grads=grads,
             params_after=after)


def r2d2loss_big_cases():
    """R2D2Agent.loss + backward of the REAL reference at BASELINE config C4's sequence shape (seq 80 / burn-in 40 /
    n 3: T = 123) with B = 16 and A = 18: 1,968 frames, enough rows for every split-bf16 kernel of the HIP learner
    (trunk >= 128 frames, rec64 GEMMs, conv gradient kernels >= 128).  Inputs come from tests/synth.py
    (synth_r2d2_batch, seed 61); only the reference's outputs are stored."""
    import types

    sys.dont_write_bytecode = True
    sys.modules.setdefault("tensorboardX", types.ModuleType("tensorboardX"))
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.path.insert(0, "/root/reference/pyrela")
    import torch
    from net import AtariLSTMNet
    from r2d2 import R2D2Agent
    from synth import synth_lstm_params

    torch.set_num_threads(8)
    A, B, seq, burn, n, gamma, eta = 18, 16, 80, 40, 3, 0.997, 0.9
    agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, A), "cpu", n, gamma, eta, seq, burn, 0)
    sd = {}
    for prefix, seed in (("online_net.", 7007), ("target_net.", 8008)):
        for k, v in synth_lstm_params(A, seed).items():
            sd[prefix + k] = torch.from_numpy(v)
    agent.load_state_dict(sd)
    d = synth_r2d2_batch(61, A, B, seq, burn, n)
    T = burn + seq + n
    tt = torch.from_numpy
    batch = types.SimpleNamespace(
        obs={"s": tt(d["s"]), "legal_move": tt(d["legal"]), "eps": torch.zeros(T, B, 1)},
        h0={"h0": tt(d["h0"]), "c0": tt(d["c0"])}, action={"a": tt(d["action"])}, reward=tt(d["reward"]),
        terminal=tt(d["terminal"]).bool(), bootstrap=tt(d["bootstrap"]), seq_len=tt(d["seq_len"]))
    loss, priority = agent.loss(batch)
    (loss * tt(d["weight"])).mean().backward()
    named = dict(agent.online_net.named_parameters())
    grads = {}
    for k, v in named.items():
        t = v.grad.detach().double().reshape(-1)
        idx = np.random.default_rng(7).integers(0, t.numel(), 32)
        grads[k] = {"l2": float(t.norm()), "absmax": float(t.abs().max()), "idx": idx.tolist(),
                    "val": t[torch.from_numpy(idx)].tolist()}
    save("r2d2_loss_A18_B16_T123", [], [], num_action=A, B=B, seq_len=seq, burn_in=burn, multi_step=n, gamma=gamma,
         eta=eta, online_seed=7007, target_seed=8008, batch_seed=61, seq_lens=d["seq_len"].tolist(),
         loss=loss.detach().double().tolist(), priority=priority.double().tolist(), grads=grads)


def r2d2buf_cases():
    """R2D2TransitionBuffer traces: episodes shorter than the window, full windows with carry-over,
    terminals inside the carried region, overlapping carry (burn+n > seq), burn_in = 0."""
    for name, K, n, seq, burn, pterm, seed in [("r2d2buf_k3_n3_s8_b4", 3, 3, 8, 4, 0.07, 1),
                                                ("r2d2buf_k2_n1_s4_b4", 2, 1, 4, 4, 0.10, 2),
                                                ("r2d2buf_k4_n3_s8_b0", 4, 3, 8, 0, 0.06, 3),
                                                ("r2d2buf_k2_n5_s6_b2", 2, 5, 6, 2, 0.05, 4),
                                                ("r2d2buf_k1_n3_s80_b40", 1, 3, 80, 40, 0.01, 5)]:
        rng = np.random.default_rng(seed)
        s = ["new %d %d %d %d" % (K, n, seq, burn)]
        for _ in range(400 if seq == 80 else 150):
            t = (rng.uniform(size=K) < pterm).astype(int)
            p = rng.uniform(0, 2, K).astype(np.float32)
            s.append("push %s %s" % (" ".join(str(v) for v in t), " ".join(f2h(v) for v in p)))
        save(name, s, run("r2d2_kat", s), K=K, multi_step=n, seq_len=seq, burn_in=burn)


def r2d2agg_cases():
    """R2D2Agent.aggregate_priority from the reference's own pyrela/r2d2.py."""
    import types

    sys.dont_write_bytecode = True
    sys.modules.setdefault("tensorboardX", types.ModuleType("tensorboardX"))
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.path.insert(0, "/root/reference/pyrela")
    import torch
    from net import AtariLSTMNet
    from r2d2 import R2D2Agent

    out = []
    for seq, burn, eta, seed in [(8, 4, 0.9, 1), (80, 40, 0.9, 2), (6, 0, 0.5, 3)]:
        agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, 6), "cpu", 3, 0.997, eta, seq, burn, 0)
        rng = np.random.default_rng(seed)
        nseq = 5
        prio = rng.uniform(0, 3, (nseq, seq)).astype(np.float32)
        lens = rng.integers(burn + 1, burn + seq + 1, nseq).astype(np.float32)
        for q in range(nseq):  # what the buffer hands over: zeros past the episode end
            prio[q, max(0, int(lens[q]) - burn):] = 0
        agg = agent.aggregate_priority(torch.from_numpy(prio), torch.from_numpy(lens))
        out.append(dict(seq_len=seq, burn_in=burn, eta=eta, priority=[[f2h(v) for v in row] for row in prio],
                        lens=lens.tolist(), agg=[f2h(v) for v in agg.numpy()]))
    save("r2d2_aggregate", [], out)


def _r2d2_batch(seed, A, B, seq, burn, n):
    """Deterministic RNNTransition-shaped batch ([T,B,...], types.cc:140-182) with consistent padding:
    sequence b has seq_len L_b; its train-part terminals are 1 from step L_b - burn - 1 on
    (r2d2.py:169-173 asserts exactly this); sequence 0 starts an episode (dummy burn-in)."""
    import types

    import torch

    T = burn + seq + n
    rng = np.random.default_rng(seed)
    s = synth_obs(T * B, seed + 1).reshape(T, B, 4, 84, 84)
    legal = (rng.uniform(size=(T, B, A)) < 0.85).astype(np.float32)
    legal[:, :, 0] = 1.0
    lens = np.array([burn + seq, burn + 3, burn + seq - 1][:B], np.float32)
    term = np.zeros((T, B), np.float32)
    for b in range(B):
        first = burn + int(lens[b]) - burn - 1
        if lens[b] < burn + seq:
            term[first:, b] = 1.0
    term[:burn, 0] = 1.0  # padLike'd burn-in of an episode start (r2d2_actor.h:55-66)
    boot = (1.0 - np.maximum.reduce([np.roll(term, -k, 0) for k in range(n)])).astype(np.float32)
    boot[T - n:] = 0.0
    action = np.zeros((T, B), np.int64)
    for t in range(T):
        for b in range(B):
            action[t, b] = rng.choice(np.flatnonzero(legal[t, b]))
    reward = rng.normal(0, 1.2, (T, B)).astype(np.float32)
    h0 = rng.normal(0, 0.3, (1, B, 512)).astype(np.float32)
    c0 = rng.normal(0, 0.3, (1, B, 512)).astype(np.float32)
    weight = rng.uniform(0.2, 1.0, B).astype(np.float32)
    tt = torch.from_numpy
    batch = types.SimpleNamespace(
        obs={"s": tt(s), "legal_move": tt(legal), "eps": torch.zeros(T, B, 1)}, h0={"h0": tt(h0), "c0": tt(c0)},
        action={"a": tt(action)}, reward=tt(reward), terminal=tt(term).bool(), bootstrap=tt(boot), seq_len=tt(lens))
    meta = dict(legal=legal.tolist(), action=action.tolist(), reward=reward.tolist(), terminal=term.tolist(),
                bootstrap=boot.tolist(), seq_len=lens.tolist(), weight=weight.tolist(), obs_seed=seed + 1,
                h0=[f2h(v) for v in h0.reshape(-1)], c0=[f2h(v) for v in c0.reshape(-1)])
    return batch, tt(weight), meta


def r2d2loss_cases():
    """R2D2Agent.loss of the REAL reference (pyrela/r2d2.py:122-206) + the backward of
    (loss * weight).mean() (pyrela/main.py:226-229) on one small batch, CPU."""
    import types

    sys.dont_write_bytecode = True
    sys.modules.setdefault("tensorboardX", types.ModuleType("tensorboardX"))
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.path.insert(0, "/root/reference/pyrela")
    import torch
    from net import AtariLSTMNet
    from r2d2 import R2D2Agent
    from synth import synth_lstm_params

    torch.set_num_threads(4)
    A, B, seq, burn, n, gamma, eta = 6, 3, 6, 2, 2, 0.997, 0.9
    agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, A), "cpu", n, gamma, eta, seq, burn, 0)
    sd = {}
    for prefix, seed in (("online_net.", 5005), ("target_net.", 6006)):
        for k, v in synth_lstm_params(A, seed).items():
            sd[prefix + k] = torch.from_numpy(v)
    agent.load_state_dict(sd)
    batch, weight, meta = _r2d2_batch(51, A, B, seq, burn, n)
    loss, priority = agent.loss(batch)
    (loss * weight).mean().backward()
    named = dict(agent.online_net.named_parameters())
    grads = {}
    for k, v in named.items():
        t = v.grad.detach().double().reshape(-1)
        idx = np.random.default_rng(7).integers(0, t.numel(), 32)
        grads[k] = {"l2": float(t.norm()), "absmax": float(t.abs().max()), "idx": idx.tolist(),
                    "val": t[torch.from_numpy(idx)].tolist()}
    save("r2d2_loss_A6_B3", [], [], num_action=A, B=B, seq_len=seq, burn_in=burn, multi_step=n, gamma=gamma, eta=eta,
         online_seed=5005, target_seed=6006, batch=meta, loss=loss.detach().double().tolist(),
         priority=priority.double().tolist(), grads=grads)


def powf_cases():
    """torch.pow(tensor, exponent) of the ATen build in this container (the op behind the reference's
    `torch::pow(priority, alpha_)`, rela/prioritized_replay.h:188,239,321): tensors of several lengths so that
    both the 32-wide SLEEF part and the scalar n % 32 tail are recorded; values as hex floats."""
    import torch

    torch.set_num_threads(1)
    rng = np.random.default_rng(77)
    out = []
    for ex in (0.6, 0.9, -0.4, -0.6, 0.4, -1.0):  # 0.5 is ATen's own vectorised sqrt, not pow: not pinned
        for n in (80, 512, 31, 33, 200, 8):
            kinds = [np.abs(rng.normal(0, 1, n)) + 1e-6, np.exp(rng.uniform(np.log(1e-12), np.log(1e6), n)),
                     rng.uniform(0, 4, n)]
            for x in kinds:
                x = x.astype(np.float32)
                if n >= 80:  # the values around the [0.75, 1.5) mantissa split of logk, and exact powers of two
                    x[:6] = np.array([1.5, np.nextafter(np.float32(1.5), np.float32(0)), np.nextafter(np.float32(1.5), np.float32(9)), 0.75, 1.0, 2.0], np.float32)
                    x[-3:] = np.array([0.0, 1.0, 3.0], np.float32)
                y = torch.pow(torch.from_numpy(x), float(np.float32(ex))).numpy()
                out.append(dict(exponent=f2h(ex), n=n, x=[f2h(v) for v in x], y=[f2h(v) for v in y]))
    save("aten_powf_vectors", [], out)


def e2e_cases(cfg_name="CFG", out_name="e2e_lockstep_apex"):
    """The REAL reference end to end: its pybind module (oracle/_ref/rela*.so), its TorchScript
    ApexAgent on the CPU, our synthetic env compiled against its rela/env.h."""
    code = r"""
import json, sys
sys.dont_write_bytecode = True
sys.path[:0] = [%r, %r, "/root/reference/pyrela"]
import torch
torch.set_num_threads(1)
import rela, synth_atari
assert "_ref" in rela.__file__
from apex import ApexAgent
from net import AtariFFNet
from e2e_lockstep import %s as CFG, run_lockstep, load_agent_params
agent = load_agent_params(ApexAgent(lambda: AtariFFNet(CFG["num_action"]), CFG["multi_step"], CFG["gamma"]), CFG)
rounds = run_lockstep(rela, synth_atari, agent, "cpu", "cpu", CFG)
print("RESULT" + json.dumps(rounds))
""" % (REFBIN, os.path.dirname(HERE), cfg_name)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1]
    import e2e_lockstep
    save(out_name, [], json.loads(line[len("RESULT"):]), cfg=getattr(e2e_lockstep, cfg_name))


def e2e_big_cases(cfg_name="CFG_BIG", out_name="e2e_lockstep_apex_k128"):
    """r4: the lock-step run at K = 128 rows (one reference actor thread, one TorchScript call per 128 envs), recorded
    from the REAL reference on the CPU, plus what justifies exact comparison of the greedy actions under the split-bf16
    mode's tolerance: over every frame the run can see (the first 24 observations of each env; the run acts on 15) the reference's own
    online net gives max|Q| and the smallest gap between the two best legal Q-values."""
    code = r"""
import json, sys
sys.dont_write_bytecode = True
sys.path[:0] = [%r, %r, "/root/reference/pyrela"]
import numpy as np
import torch
torch.set_num_threads(8)
import rela, synth_atari
assert "_ref" in rela.__file__
from apex import ApexAgent
from net import AtariFFNet
from e2e_lockstep import %s as CFG, run_lockstep, load_agent_params, frames_of_run
agent = load_agent_params(ApexAgent(lambda: AtariFFNet(CFG["num_action"]), CFG["multi_step"], CFG["gamma"]), CFG)
rounds = run_lockstep(rela, synth_atari, agent, "cpu", "cpu", CFG)
frames = frames_of_run(synth_atari, CFG, 24)
gap, qmax = float("inf"), 0.0
with torch.no_grad():
    for t in range(frames.shape[0]):
        obs = {"s": torch.from_numpy(frames[t]), "legal_move": torch.ones(CFG["K"], CFG["num_action"])}
        q = agent.online_net(obs)
        top = q.topk(2, dim=1)[0]
        gap = min(gap, float((top[:, 0] - top[:, 1]).min()))
        qmax = max(qmax, float(q.abs().max()))
print("RESULT" + json.dumps({"rounds": rounds, "min_top2_gap": gap, "max_abs_q": qmax, "frames_checked": int(frames.shape[0])}))
""" % (REFBIN, os.path.dirname(HERE), cfg_name)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if out.returncode != 0:
        print(out.stdout[-2000:], out.stderr[-4000:])
        raise SystemExit(1)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1]
    import e2e_lockstep
    rec = json.loads(line[len("RESULT"):])
    save(out_name, [], rec["rounds"], cfg=getattr(e2e_lockstep, cfg_name), min_top2_gap=rec["min_top2_gap"],
         max_abs_q=rec["max_abs_q"], frames_checked=rec["frames_checked"])


def e2e_k512_scan(cfg_name="CFG_K512"):
    """Prints, for the frames the K = 512 lock-step run can see, the advantage-bias offset of action 14 that maximises
    the smallest gap between the two best Q-values (the value then written into tests/e2e_lockstep.py CFG_K512)."""
    code = r"""
import json, sys
sys.dont_write_bytecode = True
sys.path[:0] = [%r, %r, "/root/reference/pyrela"]
import numpy as np
import torch
torch.set_num_threads(8)
import rela, synth_atari
from apex import ApexAgent
from net import AtariFFNet
from e2e_lockstep import %s as CFG, load_agent_params, frames_of_run
cfg = dict(CFG, online_fc_a_bias_add=None)
agent = load_agent_params(ApexAgent(lambda: AtariFFNet(cfg["num_action"]), cfg["multi_step"], cfg["gamma"]), cfg)
frames = frames_of_run(synth_atari, cfg, 24)
qs = []
with torch.no_grad():
    for t in range(frames.shape[0]):
        obs = {"s": torch.from_numpy(frames[t]), "legal_move": torch.ones(cfg["K"], cfg["num_action"])}
        qs.append(agent.online_net(obs).double().numpy())
q = np.concatenate(qs)
best = None
for delta in np.linspace(0.0, 0.03, 3001):
    qq = q.copy(); qq[:, 14] += delta
    top = np.sort(qq, axis=1)[:, -2:]
    gap = float((top[:, 1] - top[:, 0]).min())
    other = int((qq.argmax(1) != 14).sum())
    if other >= 20 and (best is None or gap > best[1]):
        best = (float(delta), gap, other)
print("RESULT" + json.dumps({"delta": best[0], "min_gap": best[1], "decisions_not_14": best[2], "rows": int(q.shape[0])}))
""" % (REFBIN, os.path.dirname(HERE), cfg_name)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print(out.stdout[-2000:], out.stderr[-3000:])


def e2e_r2d2_cases(cfg_name="CFG_R2D2", out_name="e2e_lockstep_r2d2", quiet=1.0):
    """The REAL reference's R2D2 path end to end (H6-shimmed module, see oracle/Makefile)."""
    code = r"""
import json, sys, types
sys.dont_write_bytecode = True
sys.modules.setdefault("tensorboardX", types.ModuleType("tensorboardX"))
sys.modules["tensorboardX"].SummaryWriter = object
sys.path[:0] = [%r, %r, %r, "/root/reference/pyrela"]
import torch
torch.set_num_threads(1)
import rela, synth_atari
assert "_ref/h6" in rela.__file__
from r2d2 import R2D2Agent
from net import AtariLSTMNet
from e2e_lockstep import %s as C, run_lockstep_r2d2, load_lstm_agent_params
agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, C["num_action"]), "cpu", C["multi_step"], C["gamma"], C["eta"],
                  C["seq_len"], C["burn_in"], 0)
load_lstm_agent_params(agent, C)
rounds = run_lockstep_r2d2(rela, synth_atari, agent, "cpu", "cpu", C, quiet=%r)
print("RESULT" + json.dumps(rounds))
""" % (os.path.join(REFBIN, "h6"), REFBIN, os.path.dirname(HERE), cfg_name, quiet)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    if out.returncode != 0:
        print(out.stdout[-2000:], out.stderr[-4000:])
        raise SystemExit(1)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1]
    import e2e_lockstep
    save(out_name, [], json.loads(line[len("RESULT"):]), cfg=getattr(e2e_lockstep, cfg_name))


if __name__ == "__main__":
    which = sys.argv[1:] or ["replay", "nstep", "ffnet", "e2e", "r2d2buf", "r2d2agg", "e2e_r2d2"]
    if "e2e_r2d2" in which:
        e2e_r2d2_cases()
    if "e2e_r2d2_c4" in which:  # BASELINE config C4's window shape (seq 80 / burn-in 40 / n 3); ~5 minutes
        e2e_r2d2_cases("CFG_R2D2_C4", "e2e_lockstep_r2d2_c4", quiet=12.0)
    if "r2d2buf" in which:
        r2d2buf_cases()
    if "r2d2agg" in which:
        r2d2agg_cases()
    if "e2e" in which:
        e2e_cases()
    if "e2e_big" in which:  # r4: K = 128 rows, reaches the split-bf16 kernels; ~1 minute
        e2e_big_cases()
    if "e2e_k512_scan" in which:
        e2e_k512_scan()
    if "e2e_k512" in which:  # r5: K = 512 rows, reaches every three-part kernel of the f32x3 mode; ~2 minutes
        e2e_big_cases("CFG_K512", "e2e_lockstep_apex_k512")
    if "e2e_sliding" in which:  # Atari-like sliding frame stacks (for the de-duplicating replay)
        e2e_cases("CFG_SLIDING", "e2e_lockstep_apex_sliding")
    if "replay" in which:
        replay_cases()
    if "replay_r2" in which:
        replay_cases_r2()
    if "nstep" in which:
        nstep_cases()
    if "ffnet" in which:
        ffnet_cases()
    if "learner" in which:
        learner_cases()
    if "r2d2loss" in which:
        r2d2loss_cases()
    if "powf" in which:
        powf_cases()
    if "ffnet_big" in which:      # r3: batches that reach the split-bf16 kernels
        ffnet_big_cases()
    if "learner_big" in which:
        learner_big_cases()
    if "learner_b512" in which:   # r5: only the B = 512 case
        learner_big_cases("learner_apex_A18_B512")
    if "r2d2loss_big" in which:   # ~2 minutes on 8 cores
        r2d2loss_big_cases()
