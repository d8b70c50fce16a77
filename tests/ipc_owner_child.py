"""Child process of tests/test_ipc_gpu.py: OWNS a replay partition on cuda:0 -- rows of an 8-byte tag field and a
4,096-byte payload whose bytes are a function of the tag, plus a sequence field of 3 sub-rows -- exports it through HIP
IPC (rela_replay_export_ipc) and, on request, samples WITHOUT gathering (ids stay in its memory), reporting what a
local gather of those ids holds so that the parent's remote gather can be checked.  Talks over stdin / stdout lines."""
import ctypes as C
import faulthandler
import json
import os
import sys
import time

T0 = time.perf_counter()
faulthandler.dump_traceback_later(150, exit=True)  # a blocked owner ends itself: the parent then fails instead of waiting
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gpu_util import cur_stream, dev, ptr
from rela_amd import _capi as capi
from rela_amd.engine import dev_view

CAP, B = int(os.environ.get("CAP", "4096")), 64
N = int(os.environ.get("N", str(CAP)))  # up to 1.25 * CAP (the ring) without a sample in between
WIDTH = int(os.environ.get("WIDTH", "4096"))  # bytes of a payload row
SEQ = 3
CHUNK_BYTES = int(os.environ.get("CHUNK_BYTES", "0"))  # > 0: the two large fields are chunks behind one virtual range


def payload_of(tags, width):
    """row bytes = (tag * 31 + column) mod 251"""
    col = np.arange(width, dtype=np.int64)[None, :]
    return ((np.asarray(tags, np.int64)[:, None] * 31 + col) % 251).astype(np.uint8)


h = C.c_void_p()
capi.check(capi.lib.rela_replay_create(C.byref(h), CAP, 5, 1.0, 0.4, 0, 0), "create")
rb = (C.c_int64 * 3)(8, WIDTH, SEQ * 1024)
steps = (C.c_int32 * 3)(1, 1, SEQ)
if CHUNK_BYTES:
    capi.check(capi.lib.rela_replay_set_chunk_bytes(h, CHUNK_BYTES), "chunk bytes")
capi.check(capi.lib.rela_replay_set_schema_seq(h, 3, rb, steps), "schema")
rng = np.random.default_rng(2)


def payload_dev(tags_dev, width):  # payload_of on the device (the 2^20-row case fills 37 GB)
    col = torch.arange(width, dtype=torch.int64, device="cuda")[None, :]
    return ((tags_dev[:, None] * 31 + col) % 251).to(torch.uint8)


def note(msg):  # progress on stderr: the 37 GB case takes a while, and a silent GPU box is taken to be hung
    print("[owner %.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


note("partition of %d slots created (chunk_bytes %d)" % (int(1.25 * CAP), CHUNK_BYTES))
BLOCK = 4096
for at in range(0, N, BLOCK):
    n = min(BLOCK, N - at)
    tags = np.arange(at, at + n, dtype=np.int64) * 7 + 3
    f0 = dev(tags)
    f1, f2 = payload_dev(f0, WIDTH), payload_dev(f0 + 1000003, SEQ * 1024)
    prio = dev(rng.uniform(0.1, 2.0, n).astype(np.float32))
    rows = (C.c_void_p * 3)(f0.data_ptr(), f1.data_ptr(), f2.data_ptr())
    capi.check(capi.lib.rela_replay_add(h, n, rows, ptr(prio), 0, cur_stream()), "add")
    torch.cuda.synchronize()
    if (at // BLOCK) % 64 == 63:
        note("%d rows inserted" % (at + n))
note("filled")
if CHUNK_BYTES:
    from rela_amd.parallel import _export_desc, _FdServer

    plain = (C.c_ubyte * 4096)()
    rc = capi.lib.rela_replay_export_ipc(h, plain)  # one handle per field cannot describe a chunked field: refused
    assert rc == capi.EINVAL and b"rela_replay_export_chunks" in capi.lib.rela_last_error(), rc
    raw, fds = _export_desc(h)
    server = _FdServer(fds)
    note("exported: %d descriptors" % len(fds))
    print("DESC2 %s %s" % (server.name, raw.hex()), flush=True)
else:
    desc = (C.c_ubyte * 4096)()
    capi.check(capi.lib.rela_replay_export_ipc(h, desc), "export")
    print("DESC " + bytes(desc).hex(), flush=True)
for line in sys.stdin:
    cmd = line.strip()
    if cmd == "sample":
        w = torch.empty(B, device="cuda")
        capi.check(capi.lib.rela_replay_sample(h, B, None, ptr(w), cur_stream()), "sample")  # ids only: no gather
        torch.cuda.synchronize()  # ... and complete before the learner is told
        note("sampled")
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
        raw_p, sum_p = C.c_void_p(), C.c_void_p()  # the float sum the draws were made against (BEFORE this sample's eviction)
        capi.check(capi.lib.rela_replay_last_sample_dev(h, C.byref(raw_p), C.byref(sum_p)), "last_sample")
        drawn_against = float(dev_view(sum_p.value, (1,), torch.float32, torch.device("cuda:0")).item())
        print("SAMPLED " + json.dumps({"tags": slot_tags.tolist(), "raw_w": [float(x) for x in raw], "sum": drawn_against,
                                       "slots": [int(x) for x in ids]}), flush=True)
    elif cmd == "update":
        capi.check(capi.lib.rela_replay_update_priority(h, B, ptr(torch.ones(B, device="cuda")), 1, cur_stream()), "update")
        torch.cuda.synchronize()
        print("UPDATED", flush=True)
    elif cmd == "quit":
        break
capi.lib.rela_replay_destroy(h)
