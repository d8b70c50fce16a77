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
