"""Pins oracle/r2d2_oracle.c to traces of the REAL R2D2TransitionBuffer (rela/r2d2_actor.h:10-187,
driven by oracle/ref_harness/r2d2_kat.cc) and to the reference's aggregate_priority
(pyrela/r2d2.py:103-120).  CPU only."""
import ctypes as C
import glob
import json
import os

import numpy as np
import pytest

from oracle_lib import f2h, h2f, load

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bind(lib):
    P = C.POINTER
    lib.oracle_r2d2buf_new.restype = C.c_void_p
    lib.oracle_r2d2buf_new.argtypes = [C.c_int] * 4
    lib.oracle_r2d2buf_free.argtypes = [C.c_void_p]
    lib.oracle_r2d2buf_push.argtypes = [C.c_void_p] + [C.c_void_p] * 7
    lib.oracle_r2d2buf_pop.argtypes = [C.c_void_p] + [C.c_void_p] * 9
    lib.oracle_r2d2_aggregate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


class OracleR2D2Buf:
    def __init__(self, K, n, seq, burn):
        self.lib = bind(load())
        self.K, self.n, self.seq, self.burn, self.T = K, n, seq, burn, burn + seq + n
        self.h = self.lib.oracle_r2d2buf_new(K, n, seq, burn)
        self.step = 0
        self.fresh = [True] * K

    def push(self, term, prio):
        """Same tagging as oracle/ref_harness/r2d2_kat.cc."""
        K, s = self.K, self.step
        tag = (np.arange(K) + s * 100).astype(np.int64)
        act = np.full(K, s, np.int64)
        rew = np.full(K, s + 0.5, np.float32)
        boot = np.full(K, float(s % 2), np.float32)
        hid = np.array([0.0 if f else s + 1.0 for f in self.fresh], np.float32)
        t = np.asarray(term, np.uint8)
        p = np.asarray(prio, np.float32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        can = self.lib.oracle_r2d2buf_push(self.h, vp(tag), vp(act), vp(rew), vp(boot), vp(t), vp(p), vp(hid))
        self.fresh = [bool(x) for x in t]
        self.step += 1
        return bool(can)

    def pop(self):
        K, T, seq = self.K, self.T, self.seq
        Q = 2 * K
        ln, h0, env = np.zeros(Q, np.float32), np.zeros(Q, np.float32), np.zeros(Q, np.int32)
        tag, act = np.zeros((Q, T), np.int64), np.zeros((Q, T), np.int64)
        rew, boot = np.zeros((Q, T), np.float32), np.zeros((Q, T), np.float32)
        term = np.zeros((Q, T), np.uint8)
        prio = np.zeros((Q, seq), np.float32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        q = self.lib.oracle_r2d2buf_pop(self.h, vp(ln), vp(h0), vp(env), vp(tag), vp(act), vp(rew), vp(term), vp(boot),
                                        vp(prio))
        return [dict(len=float(ln[i]), h0=float(h0[i]), env=int(env[i]), tag=tag[i].tolist(), a=act[i].tolist(),
                     reward=rew[i].tolist(), terminal=term[i].tolist(), bootstrap=boot[i].tolist(),
                     prio=[f2h(v) for v in prio[i]]) for i in range(q)]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "r2d2buf_*.json"))), ids=os.path.basename)
def test_r2d2_buffer_matches_reference(path):
    case = json.load(open(path))
    buf = OracleR2D2Buf(case["K"], case["multi_step"], case["seq_len"], case["burn_in"])
    K = case["K"]
    n_seq = 0
    kinds = set()
    for line, exp in zip(case["script"][1:], case["expect"][1:]):
        tok = line.split()[1:]
        can = buf.push([int(t) for t in tok[:K]], [h2f(t) for t in tok[K:]])
        assert can == exp["pop"]
        if not can:
            continue
        got = buf.pop()
        assert len(got) == len(exp["seqs"])
        for g, e in zip(got, exp["seqs"]):
            n_seq += 1
            for k in ("len", "h0", "tag", "a", "reward", "terminal", "bootstrap", "prio"):
                assert g[k] == e[k], k
            kinds.add("short" if e["len"] < case["burn_in"] + case["seq_len"] else "full")
    assert n_seq >= 5 and kinds == {"short", "full"}  # the traces exercise both pop branches


def test_aggregate_priority_matches_reference():
    lib = bind(load())
    for case in json.load(open(os.path.join(GOLD, "r2d2_aggregate.json")))["expect"]:
        prio = np.array([[h2f(v) for v in row] for row in case["priority"]], np.float32)
        lens = np.array(case["lens"], np.float32)
        out = np.zeros(len(lens), np.float32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        lib.oracle_r2d2_aggregate(len(lens), case["seq_len"], case["burn_in"], case["eta"], vp(prio), vp(lens), vp(out))
        ref = np.array([h2f(v) for v in case["agg"]], np.float32)
        np.testing.assert_allclose(out, ref, rtol=2e-7, atol=0)  # torch sums pairwise; the oracle left to right


@pytest.mark.parametrize("name", ["r2d2_loss_A6_B3", "r2d2_loss_A18_B16_T123"])
def test_r2d2_learner_loss_matches_reference_golden(name):
    """rela_amd/pyrela/r2d2.py (R2D2Agent.td_err / loss / aggregate_priority, the learner side of SURVEY
    rows N2/G2) against vectors recorded from the REAL reference's R2D2Agent on CPU
    (tests/golden/make_golden.py r2d2loss: loss per sequence, aggregated priority, and the gradients of
    (loss * weight).mean() w.r.t. every online parameter), including a padded short sequence and a
    sequence that starts an episode (zeroed state after the dummy burn-in)."""
    import sys
    from types import SimpleNamespace

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from rela_amd.pyrela.net import AtariLSTMNet
    from rela_amd.pyrela.r2d2 import R2D2Agent
    from synth import synth_lstm_params, synth_obs

    g = json.load(open(os.path.join(GOLD, name + ".json")))
    A, B, seq, burn, n = g["num_action"], g["B"], g["seq_len"], g["burn_in"], g["multi_step"]
    T = burn + seq + n
    torch.set_num_threads(min(8, os.cpu_count() or 4))
    agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, A), "cpu", n, g["gamma"], g["eta"], seq, burn, 0)
    sd = {}
    for prefix, seed in (("online_net.", g["online_seed"]), ("target_net.", g["target_seed"])):
        for k, v in synth_lstm_params(A, seed).items():
            sd[prefix + k] = torch.from_numpy(v)
    agent.load_state_dict(sd)
    f32 = lambda x: torch.tensor(x, dtype=torch.float32)
    if "batch" in g:
        m = g["batch"]
        hid = lambda key: torch.tensor([h2f(v) for v in m[key]], dtype=torch.float32).reshape(1, B, 512)
        batch = SimpleNamespace(
            obs={"s": torch.from_numpy(synth_obs(T * B, m["obs_seed"]).reshape(T, B, 4, 84, 84)),
                 "legal_move": f32(m["legal"]), "eps": torch.zeros(T, B, 1)},
            h0={"h0": hid("h0"), "c0": hid("c0")}, action={"a": torch.tensor(m["action"], dtype=torch.int64)},
            reward=f32(m["reward"]), terminal=f32(m["terminal"]).bool(), bootstrap=f32(m["bootstrap"]),
            seq_len=f32(m["seq_len"]))
    else:  # C4's sequence shape: the inputs are re-derived from tests/synth.py
        from synth import synth_r2d2_batch

        d = synth_r2d2_batch(g["batch_seed"], A, B, seq, burn, n)
        tt = torch.from_numpy
        m = {"weight": d["weight"].tolist()}
        batch = SimpleNamespace(
            obs={"s": tt(d["s"]), "legal_move": tt(d["legal"]), "eps": torch.zeros(T, B, 1)},
            h0={"h0": tt(d["h0"]), "c0": tt(d["c0"])}, action={"a": tt(d["action"])}, reward=tt(d["reward"]),
            terminal=tt(d["terminal"]).bool(), bootstrap=tt(d["bootstrap"]), seq_len=tt(d["seq_len"]))
    loss, prio = agent.loss(batch)
    np.testing.assert_allclose(loss.detach().numpy(), np.array(g["loss"]), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(prio.numpy(), np.array(g["priority"]), rtol=1e-4, atol=1e-5)
    (loss * f32(m["weight"])).mean().backward()
    named = dict(agent.online_net.named_parameters())
    assert set(named) == set(g["grads"])
    for key, rec in g["grads"].items():
        t = named[key].grad.detach().double().reshape(-1)
        scale = rec["absmax"] + 1e-12
        np.testing.assert_allclose(float(t.norm()), rec["l2"], rtol=1e-3, atol=1e-4 * scale, err_msg=key)
        np.testing.assert_allclose(t[torch.tensor(rec["idx"])].numpy(), np.array(rec["val"]), rtol=1e-3,
                                   atol=1e-3 * scale, err_msg=key)
