"""Host-logic tests of the R2D2 sequence bookkeeping (rela_amd/csrc/r2d2_seq_core.h): the plan it
emits, executed on host arrays by tests/cpu_shims/r2d2_seq_host.cpp, must reproduce the traces of
the REAL R2D2TransitionBuffer (tests/golden/r2d2buf_*.json) and agree with the oracle on random
scenarios.  CPU only."""
import ctypes as C
import glob
import json
import os
import subprocess

import numpy as np
import pytest

from oracle_lib import f2h, h2f
from test_oracle_r2d2 import OracleR2D2Buf

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def shim():
    src = os.path.join(HERE, "cpu_shims", "r2d2_seq_host.cpp")
    hdr = os.path.join(HERE, "..", "rela_amd", "csrc", "r2d2_seq_core.h")
    so = os.path.join(HERE, "cpu_shims", "libr2d2_seq_host.so")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src], check=True)
    lib = C.CDLL(so)
    lib.shim_r2d2_new.restype = C.c_void_p
    lib.shim_r2d2_new.argtypes = [C.c_int] * 4
    lib.shim_r2d2_free.argtypes = [C.c_void_p]
    lib.shim_r2d2_step.argtypes = [C.c_void_p] * 16
    return lib


class ShimBuf:
    def __init__(self, lib, K, n, seq, burn):
        self.lib, self.K, self.n, self.seq, self.burn, self.T = lib, K, n, seq, burn, burn + seq + n
        self.h = lib.shim_r2d2_new(K, n, seq, burn)
        self.step_no = 0
        self.fresh = [True] * K

    def step(self, term, prio):
        K, s, T, seq = self.K, self.step_no, self.T, self.seq
        tag = (np.arange(K) + s * 100).astype(np.int64)
        act = np.full(K, s, np.int64)
        rew = np.full(K, s + 0.5, np.float32)
        boot = np.full(K, float(s % 2), np.float32)
        hid = np.array([0.0 if f else s + 1.0 for f in self.fresh], np.float32)
        t = np.asarray(term, np.uint8)
        p = np.asarray(prio, np.float32)
        Q = 2 * K
        ln, h0 = np.zeros(Q, np.float32), np.zeros(Q, np.float32)
        otag, oact = np.zeros((Q, T), np.int64), np.zeros((Q, T), np.int64)
        orew, oboot = np.zeros((Q, T), np.float32), np.zeros((Q, T), np.float32)
        oterm = np.zeros((Q, T), np.uint8)
        oprio = np.zeros((Q, seq), np.float32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        q = self.lib.shim_r2d2_step(self.h, vp(tag), vp(act), vp(rew), vp(boot), vp(t), vp(p), vp(hid), vp(ln), vp(h0),
                                    vp(otag), vp(oact), vp(orew), vp(oterm), vp(oboot), vp(oprio))
        self.fresh = [bool(x) for x in t]
        self.step_no += 1
        return [dict(len=float(ln[i]), h0=float(h0[i]), tag=otag[i].tolist(), a=oact[i].tolist(), reward=orew[i].tolist(),
                     terminal=oterm[i].tolist(), bootstrap=oboot[i].tolist(), prio=[f2h(v) for v in oprio[i]])
                for i in range(q)]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "r2d2buf_*.json"))), ids=os.path.basename)
def test_plan_reproduces_reference_traces(shim, path):
    case = json.load(open(path))
    K = case["K"]
    buf = ShimBuf(shim, K, case["multi_step"], case["seq_len"], case["burn_in"])
    for line, exp in zip(case["script"][1:], case["expect"][1:]):
        tok = line.split()[1:]
        got = buf.step([int(t) for t in tok[:K]], [h2f(t) for t in tok[K:]])
        assert bool(got) == exp["pop"]
        if not got:
            continue
        assert len(got) == len(exp["seqs"])
        for g, e in zip(got, exp["seqs"]):
            for k in ("len", "h0", "tag", "a", "reward", "terminal", "bootstrap", "prio"):
                assert g[k] == e[k], k


@pytest.mark.parametrize("K,n,seq,burn,pterm", [(5, 3, 8, 4, 0.1), (3, 2, 5, 5, 0.15), (4, 5, 5, 0, 0.05),
                                                 (2, 3, 80, 40, 0.02), (6, 1, 3, 3, 0.3)])
def test_plan_vs_oracle_random(shim, K, n, seq, burn, pterm):
    rng = np.random.default_rng(K * 1000 + seq)
    a, b = ShimBuf(shim, K, n, seq, burn), OracleR2D2Buf(K, n, seq, burn)
    emitted = 0
    for _ in range(600):
        t = (rng.uniform(size=K) < pterm).astype(np.uint8)
        p = rng.uniform(0, 2, K).astype(np.float32)
        got = a.step(t, p)
        can = b.push(t, p)
        assert bool(got) == can
        if can:
            exp = b.pop()
            assert len(got) == len(exp)
            for g, e in zip(got, exp):
                emitted += 1
                for k in ("len", "h0", "tag", "a", "reward", "terminal", "bootstrap", "prio"):
                    assert g[k] == e[k], k
    assert emitted > 10
