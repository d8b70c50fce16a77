"""Helpers for the GPU parity tests: device buffers via torch, calls through the C ABI (ctypes)."""
import ctypes as C

import numpy as np
import torch

from rela_amd import _capi as capi


def dev(arr):
    """numpy -> contiguous cuda tensor (same dtype)."""
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def cur_stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class GpuReplay:
    """Test-side wrapper over rela_replay_* with a single int64 'tag' field (+ optional payload)."""

    def __init__(self, capacity, seed, alpha, beta, row_bytes=(8,)):
        h = C.c_void_p()
        capi.check(capi.lib.rela_replay_create(C.byref(h), capacity, seed, alpha, beta, 0, 0), "rela_replay_create")
        self.h = h
        self.row_bytes = list(row_bytes)
        rb = (C.c_int64 * len(row_bytes))(*row_bytes)
        capi.check(capi.lib.rela_replay_set_schema(h, len(row_bytes), rb), "rela_replay_set_schema")
        self.keep = []

    def close(self):
        if self.h:
            capi.lib.rela_replay_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def add(self, fields, prio, nonblocking=1):
        """fields: list of cuda tensors [n, ...] matching the schema; prio: numpy/cuda f32[n]."""
        p = prio if isinstance(prio, torch.Tensor) else dev(np.asarray(prio, np.float32))
        n = p.numel()
        rows = (C.c_void_p * len(fields))(*[f.data_ptr() for f in fields])
        rc = capi.lib.rela_replay_add(self.h, n, rows, ptr(p), nonblocking, cur_stream())
        self.keep = [fields, p]
        return rc

    def add_tags(self, tags, prio):
        t = dev(np.asarray(tags, np.int64))
        return self.add([t], prio)

    def sample(self, batch, gather=True):
        outs = [torch.empty((batch, rb), dtype=torch.uint8, device="cuda") for rb in self.row_bytes]
        w = torch.empty(batch, dtype=torch.float32, device="cuda")
        rows = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs]) if gather else None
        rc = capi.lib.rela_replay_sample(self.h, batch, rows, ptr(w), cur_stream())
        return rc, outs, w

    def update(self, prio, on_device=False):
        if on_device:
            p = prio if isinstance(prio, torch.Tensor) else dev(np.asarray(prio, np.float32))
            self.keep.append(p)
            return capi.lib.rela_replay_update_priority(self.h, p.numel(), ptr(p), 1, cur_stream())
        p = np.ascontiguousarray(prio, np.float32)
        return capi.lib.rela_replay_update_priority(self.h, len(p), p.ctypes.data_as(C.c_void_p), 0, cur_stream())

    def state(self, n=0):
        st = capi.ReplayState()
        ids = np.zeros(max(n, 1), np.int32)
        raw = np.zeros(max(n, 1), np.float32)
        tg = np.zeros(max(n, 1), np.float32)
        capi.check(capi.lib.rela_replay_debug_state(self.h, C.byref(st), ids.ctypes.data_as(C.c_void_p),
                                                    raw.ctypes.data_as(C.c_void_p), tg.ctypes.data_as(C.c_void_p)),
                   "rela_replay_debug_state")
        d = {k: getattr(st, k) for k, _ in capi.ReplayState._fields_}
        d.update(ids=ids[:n], raw_w=raw[:n], targets=tg[:n])
        return d

    def weights(self):
        ring = self.state()["ring"]
        w = np.zeros(ring, np.float32)
        ev = np.zeros(ring, np.uint8)
        capi.check(capi.lib.rela_replay_debug_weights(self.h, w.ctypes.data_as(C.c_void_p),
                                                      ev.ctypes.data_as(C.c_void_p)), "debug_weights")
        return w, ev


def seqscan(ring_w_dev, head, size, targets):
    t = np.ascontiguousarray(targets, np.float64)
    nt = len(t)
    k = np.zeros(nt, np.int64)
    A = np.zeros(nt, np.float64)
    w = np.zeros(nt, np.float32)
    total = C.c_double()
    capi.check(capi.lib.rela_seqscan_search(ptr(ring_w_dev), ring_w_dev.numel(), head, size,
                                            t.ctypes.data_as(C.c_void_p), nt, k.ctypes.data_as(C.c_void_p),
                                            A.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p),
                                            C.byref(total), cur_stream()), "rela_seqscan_search")
    return k, A, w, total.value
