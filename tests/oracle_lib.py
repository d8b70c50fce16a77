"""ctypes view of oracle/liboracle.so -- test infrastructure only (see oracle/oracle.h)."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def h2f(h):
    return struct.unpack(">f", bytes.fromhex(h))[0]


def f2h(x):
    return struct.pack(">f", float(np.float32(x))).hex()


def h2d(h):
    return struct.unpack(">d", bytes.fromhex(h))[0]


def load():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["make", "-C", ORACLE_DIR, "liboracle.so"], check=True, capture_output=True)
    lib = C.CDLL(so)
    vp, i32, i64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
    P = C.POINTER
    lib.oracle_replay_new.restype = vp
    lib.oracle_replay_new.argtypes = [i32, i32, f32, f32]
    lib.oracle_replay_free.argtypes = [vp]
    for name in ("oracle_replay_add", "oracle_replay_add_w"):
        getattr(lib, name).argtypes = [vp, i32, P(i64), P(f32)]
    lib.oracle_replay_sample.argtypes = [vp, i32, P(C.c_int32), P(i64), P(f32)]
    for name in ("oracle_replay_update", "oracle_replay_update_w"):
        getattr(lib, name).argtypes = [vp, i32, P(f32)]
    for name in ("oracle_replay_size", "oracle_replay_full_size", "oracle_replay_head", "oracle_replay_tail",
                 "oracle_replay_ring"):
        getattr(lib, name).argtypes = [vp]
    lib.oracle_replay_num_add.argtypes = [vp]
    lib.oracle_replay_num_add.restype = i64
    lib.oracle_replay_sum.argtypes = [vp]
    lib.oracle_replay_sum.restype = f64
    for name in ("oracle_replay_last_targets", "oracle_replay_last_raw_w", "oracle_replay_weights"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = P(f32)
    lib.oracle_replay_evicted.argtypes = [vp]
    lib.oracle_replay_evicted.restype = P(C.c_uint8)
    lib.oracle_cqueue_pop.argtypes = [vp, i32]
    lib.oracle_scan_search.argtypes = [P(f32), i32, P(f32), i32, P(C.c_int32), P(f64)]
    lib.oracle_nstep_pop.argtypes = [i32, i32, f32, P(f32), P(C.c_uint8), P(f32), P(f32), P(C.c_uint8)]
    lib.oracle_canonical_from_u32.argtypes = [C.c_uint32]
    lib.oracle_canonical_from_u32.restype = f32
    lib.oracle_set_threads.argtypes = [i32]
    return lib


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleReplay:
    """Thin object wrapper over the oracle_replay_* functions."""

    def __init__(self, capacity, seed, alpha, beta, lib=None):
        self.lib = lib or load()
        self.h = self.lib.oracle_replay_new(capacity, seed, alpha, beta)
        self.ring = self.lib.oracle_replay_ring(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_replay_free(self.h)
            self.h = None

    def add(self, tags, prio, weights=False):
        tags = np.ascontiguousarray(tags, np.int64)
        prio = np.ascontiguousarray(prio, np.float32)
        fn = self.lib.oracle_replay_add_w if weights else self.lib.oracle_replay_add
        return fn(self.h, len(prio), tags.ctypes.data_as(C.POINTER(C.c_int64)), fptr(prio))

    def sample(self, batch):
        ids = np.zeros(batch, np.int32)
        tags = np.zeros(batch, np.int64)
        w = np.zeros(batch, np.float32)
        rc = self.lib.oracle_replay_sample(self.h, batch, ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                           tags.ctypes.data_as(C.POINTER(C.c_int64)), fptr(w))
        return rc, ids, tags, w

    def update(self, prio, weights=False):
        prio = np.ascontiguousarray(prio, np.float32)
        fn = self.lib.oracle_replay_update_w if weights else self.lib.oracle_replay_update
        return fn(self.h, len(prio), fptr(prio))

    def last_targets(self, n):
        return np.ctypeslib.as_array(self.lib.oracle_replay_last_targets(self.h), (n,)).copy()

    def last_raw_w(self, n):
        return np.ctypeslib.as_array(self.lib.oracle_replay_last_raw_w(self.h), (n,)).copy()

    def weights(self):
        return np.ctypeslib.as_array(self.lib.oracle_replay_weights(self.h), (self.ring,)).copy()

    def evicted(self):
        return np.ctypeslib.as_array(self.lib.oracle_replay_evicted(self.h), (self.ring,)).copy()

    def state(self):
        L, h = self.lib, self.h
        return dict(head=L.oracle_replay_head(h), tail=L.oracle_replay_tail(h), size=L.oracle_replay_full_size(h),
                    safe_size=L.oracle_replay_size(h), sum=L.oracle_replay_sum(h), num_add=L.oracle_replay_num_add(h))
