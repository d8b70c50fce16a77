"""Host-logic tests for the exact sequential-f64-sum emulation (rela_amd/csrc/seqsum_core.h).

The transfer-table arithmetic is __host__ __device__; tests/cpu_shims/seqsum_host.cpp replays the
kernel structure on the CPU and is compared, bit for bit, with the oracle's sequential scan
(oracle_scan_search, restating rela/prioritized_replay.h:266-308).  CPU-only.
"""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

from oracle_lib import load as load_oracle

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim():
    src = os.path.join(HERE, "cpu_shims", "seqsum_host.cpp")
    so = os.path.join(HERE, "cpu_shims", "libseqsum_host.so")
    hdr = os.path.join(HERE, "..", "rela_amd", "csrc", "seqsum_core.h")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src], check=True)
    lib = C.CDLL(so)
    P = C.POINTER
    lib.shim_seq_search.argtypes = [P(C.c_float), C.c_int64, C.c_int64, C.c_int64, P(C.c_double), C.c_int, C.c_int,
                                    C.c_uint32, P(C.c_int64), P(C.c_double), P(C.c_float), P(C.c_double), C.c_int64,
                                    P(C.c_double)]
    return lib


def oracle_scan(w_logical, targets):
    lib = load_oracle()
    w = np.ascontiguousarray(w_logical, np.float32)
    t = np.ascontiguousarray(targets, np.float32)
    idx = np.zeros(len(t), np.int32)
    acc = np.zeros(len(t), np.float64)
    lib.oracle_scan_search(w.ctypes.data_as(C.POINTER(C.c_float)), len(w), t.ctypes.data_as(C.POINTER(C.c_float)),
                           len(t), idx.ctypes.data_as(C.POINTER(C.c_int32)), acc.ctypes.data_as(C.POINTER(C.c_double)))
    return idx, acc


def seq_total(w):
    acc = 0.0
    for x in np.asarray(w, np.float32):
        acc += float(x)
    return acc


def run_shim(shim, ring_w, head, size, targets, mode, seed=1, prefix_at=0):
    ring_w = np.ascontiguousarray(ring_w, np.float32)
    t = np.ascontiguousarray(targets, np.float64)
    k = np.zeros(len(t), np.int64)
    A = np.zeros(len(t), np.float64)
    w = np.zeros(len(t), np.float32)
    total = C.c_double()
    prefix = C.c_double()
    fp = lambda a, ty: a.ctypes.data_as(C.POINTER(ty))
    shim.shim_seq_search(fp(ring_w, C.c_float), size, head, len(ring_w), fp(t, C.c_double), len(t), mode, seed,
                         fp(k, C.c_int64), fp(A, C.c_double), fp(w, C.c_float), C.byref(total), prefix_at,
                         C.byref(prefix))
    return k, A, w, total.value, prefix.value


def gen(kind, n, rng):
    if kind == "uniform":
        return rng.uniform(0.01, 2, n).astype(np.float32)
    if kind == "lognormal":
        return np.exp(rng.normal(0, 4, n)).astype(np.float32)
    if kind == "logwide":
        return np.exp(rng.uniform(np.log(1e-12), np.log(1e8), n)).astype(np.float32)
    if kind == "sparse":
        w = rng.uniform(0, 1, n).astype(np.float32)
        w[rng.uniform(size=n) < 0.7] = 0
        return w
    if kind == "ties":
        # exact half-ulp patterns: small powers of two against a large running sum
        w = np.full(n, 2.0 ** -30, np.float32)
        w[0] = 2.0 ** 22
        w[rng.integers(0, n, n // 8)] = np.float32(3 * 2.0 ** -31)
        return w
    if kind == "pow06":
        return (np.abs(rng.normal(0, 1, n)) ** 0.6).astype(np.float32)
    if kind == "denorm":
        w = (rng.integers(1, 1 << 20, n).astype(np.float64) * 2.0 ** -149).astype(np.float32)
        w[rng.uniform(size=n) < 0.01] = 1.0
        return w
    if kind == "giant_first":
        w = rng.uniform(0, 1e-3, n).astype(np.float32)
        w[0] = 3e7
        return w
    if kind == "leading_zeros":
        w = rng.uniform(0, 1, n).astype(np.float32)
        w[: n // 3] = 0
        return w
    raise ValueError(kind)


KINDS = ["uniform", "lognormal", "logwide", "sparse", "ties", "pow06", "denorm", "giant_first", "leading_zeros"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 1024, 5000, 40000])
def test_search_is_bit_exact(shim, kind, n):
    rng = np.random.default_rng(zlib.crc32(("%s-%d" % (kind, n)).encode()))
    ring = n + int(rng.integers(0, 100))
    head = int(rng.integers(0, ring))
    ringw = rng.uniform(5, 6, ring).astype(np.float32)  # junk outside the live range
    logical = gen(kind, n, rng)
    ringw[(head + np.arange(n)) % ring] = logical
    total = seq_total(logical)
    nt = 48
    targets = np.sort(rng.uniform(0, total, nt)).astype(np.float32)
    targets[0] = 0.0  # the acc > 0 guard
    exp_idx, exp_acc = oracle_scan(logical, targets)
    for mode in (0, 1, 2, 3):
        eff = np.maximum(targets.astype(np.float64), 5e-324)
        k, A, w, tot, prefix = run_shim(shim, ringw, head, n, eff, mode, seed=n, prefix_at=n // 2)
        assert tot == total, (mode, tot, total)
        assert prefix == seq_total(logical[: n // 2])
        for i in range(nt):
            if exp_idx[i] < 0:
                assert k[i] == -1
            else:
                assert k[i] == exp_idx[i], (mode, i)
                assert A[i] == exp_acc[i]
                assert w[i] == logical[exp_idx[i]]


def test_large_ring_realistic(shim):
    """2^20-class ring with |td|^0.6-like weights: the sum is NOT exactly representable step by
    step (rounding happens), and the emulation still matches the sequential scan bit for bit."""
    rng = np.random.default_rng(7)
    n = 300_000
    logical = gen("pow06", n, rng)
    logical[rng.integers(0, n, 500)] *= np.float32(1e-5)
    total = seq_total(logical)
    tree = float(np.sum(logical.astype(np.float64)))
    assert tree != total  # sequential rounding really differs from an exact/tree sum here
    targets = np.sort(rng.uniform(0, total, 512)).astype(np.float32)
    exp_idx, exp_acc = oracle_scan(logical, targets)
    k, A, w, tot, _ = run_shim(shim, logical, 0, n, targets.astype(np.float64), 0)
    assert tot == total
    assert (k == exp_idx).all() and (A == exp_acc).all()
