"""Child process of tests/test_ipc_gpu.py: OWNS a replay partition on cuda:0 -- rows of an 8-byte tag field and a
4,096-byte payload whose bytes are a function of the tag, plus a sequence field of 3 sub-rows -- exports it through HIP
IPC (rela_replay_export_ipc) and, on request, samples WITHOUT gathering (ids stay in its memory), reporting what a
local gather of those ids holds so that the parent's remote gather can be checked.  Talks over stdin / stdout lines."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gpu_util import cur_stream, dev, ptr
from rela_amd import _capi as capi

CAP, B, N = 4096, 64, 4096
SEQ = 3


def payload_of(tags, width):
    """row bytes = (tag * 31 + column) mod 251"""
    col = np.arange(width, dtype=np.int64)[None, :]
    return ((np.asarray(tags, np.int64)[:, None] * 31 + col) % 251).astype(np.uint8)


h = C.c_void_p()
capi.check(capi.lib.rela_replay_create(C.byref(h), CAP, 5, 1.0, 0.4, 0, 0), "create")
rb = (C.c_int64 * 3)(8, 4096, SEQ * 1024)
steps = (C.c_int32 * 3)(1, 1, SEQ)
capi.check(capi.lib.rela_replay_set_schema_seq(h, 3, rb, steps), "schema")
rng = np.random.default_rng(2)
tags = np.arange(N, dtype=np.int64) * 7 + 3
f0, f1, f2 = dev(tags), dev(payload_of(tags, 4096)), dev(payload_of(tags + 1000003, SEQ * 1024))
prio = dev(rng.uniform(0.1, 2.0, N).astype(np.float32))
rows = (C.c_void_p * 3)(f0.data_ptr(), f1.data_ptr(), f2.data_ptr())
capi.check(capi.lib.rela_replay_add(h, N, rows, ptr(prio), 0, cur_stream()), "add")
torch.cuda.synchronize()
desc = (C.c_ubyte * 4096)()
capi.check(capi.lib.rela_replay_export_ipc(h, desc), "export")
print("DESC " + bytes(desc).hex(), flush=True)
for line in sys.stdin:
    cmd = line.strip()
    if cmd == "sample":
        w = torch.empty(B, device="cuda")
        capi.check(capi.lib.rela_replay_sample(h, B, None, ptr(w), cur_stream()), "sample")  # ids only: no gather
        torch.cuda.synchronize()  # ... and complete before the learner is told
        st = capi.ReplayState()
        ids = np.zeros(B, np.int32)
        raw = np.zeros(B, np.float32)
        capi.check(capi.lib.rela_replay_debug_state(h, C.byref(st), ids.ctypes.data_as(C.c_void_p),
                                                    raw.ctypes.data_as(C.c_void_p), None), "state")
        slot_tags = np.zeros(B, np.int64)
        for i, s in enumerate(ids):  # the tag stored in each sampled slot
            one = np.zeros(1, np.int64)
            capi.check(capi.lib.rela_replay_debug_read_rows(h, 0, int(s), 1, one.ctypes.data_as(C.c_void_p)), "read")
            slot_tags[i] = one[0]
        print("SAMPLED " + json.dumps({"tags": slot_tags.tolist(), "raw_w": [float(x) for x in raw],
                                       "sum": st.sum}), flush=True)
    elif cmd == "update":
        capi.check(capi.lib.rela_replay_update_priority(h, B, ptr(torch.ones(B, device="cuda")), 1, cur_stream()), "update")
        torch.cuda.synchronize()
        print("UPDATED", flush=True)
    elif cmd == "quit":
        break
capi.lib.rela_replay_destroy(h)
