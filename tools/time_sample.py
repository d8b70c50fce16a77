#!/usr/bin/env python3
"""Isolated timing of the replay sample / update_priority path at BASELINE config C2's ring
(capacity 2^20 -> ring 1,310,720 live weights, B = 512): per-kernel HIP-event averages, the sum per call
and the host-perceived latency of sample() + synchronize.

  ITERS=50 python tools/time_sample.py      -> one JSON line
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from rela_amd import _capi as capi
from rela_amd.replay import FFReplay

A, ITERS, B = 18, int(os.environ.get("ITERS", "50")), int(os.environ.get("BATCH", "512"))
cap = int(os.environ.get("CAP", str(1 << 20)))
rep = FFReplay(cap, 7, 0.6, 0.4, 0, A, "cuda:0")
rows = 65536
obs = torch.randint(0, 256, (rows, 4, 84, 84), dtype=torch.uint8, device="cuda")
z = torch.zeros(rows, device="cuda")
zi = torch.zeros(rows, dtype=torch.int64, device="cuda")
zb = torch.zeros(rows, dtype=torch.uint8, device="cuda")
eps = torch.zeros(rows, 1, device="cuda")
lg = torch.ones(rows, A, device="cuda")
ptrs = [obs.data_ptr(), obs.data_ptr(), eps.data_ptr(), eps.data_ptr(), lg.data_ptr(), lg.data_ptr(), zi.data_ptr(),
        z.data_ptr(), zb.data_ptr(), z.data_ptr()]
while rep.size() + rows <= int(1.25 * cap):
    rep.add_rows(rows, ptrs, torch.rand(rows, device="cuda") * 2 + 0.01)
small = torch.rand(6400, device="cuda") * 2 + 0.01
for _ in range(5):
    batch, w = rep.sample(B)
    rep.update_priority(torch.rand(B, device="cuda") + 0.1)
    rep.add_rows(6400, ptrs, small, nonblocking=True)
torch.cuda.synchronize()
capi.lib.rela_prof_enable(1)
for _ in range(ITERS):
    batch, w = rep.sample(B)
    rep.update_priority(torch.rand(B, device="cuda") + 0.1)
    rep.add_rows(6400, ptrs, small, nonblocking=True)  # keeps the ring at capacity: every sample evicts 6,400
torch.cuda.synchronize()
capi.lib.rela_prof_enable(0)
buf = C.create_string_buffer(1 << 16)
capi.check(capi.lib.rela_prof_summary_json(buf, len(buf)), "prof")
prof = json.loads(buf.value.decode())
us = {k: round(v["total_ms"] / ITERS * 1e3, 2) for k, v in sorted(prof.items())}
lat = []
for _ in range(ITERS):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    batch, w = rep.sample(B)
    torch.cuda.synchronize()
    lat.append(time.perf_counter() - t0)
    rep.update_priority(torch.rand(B, device="cuda") + 0.1)
    rep.add_rows(6400, ptrs, small, nonblocking=True)
torch.cuda.synchronize()
lat.sort()
sample_keys = [k for k in us if k.startswith("seq_") or k.startswith("replay_gather") or k in (
    "replay_targets", "replay_search", "replay_pop", "replay_is_weights", "replay_sample_finish")]
desc, fds = capi.ReplayChunkDesc(), (C.c_int * capi.IPC_MAX_FDS)()  # RELA_REPLAY_CHUNK_GB=8: is the ring really in chunks?
capi.check(capi.lib.rela_replay_export_chunks(rep.h, C.byref(desc), fds, capi.IPC_MAX_FDS), "export_chunks")
for i in range(desc.nfds):
    os.close(fds[i])
print(json.dumps({"field_chunks": list(desc.field_chunks[:desc.ipc.nfields]), "ring": int(1.25 * cap), "batch": B, "size": rep.size(), "us_per_call": us,
                  "sample_kernels_us": round(sum(us[k] for k in sample_keys), 1),
                  "update_us": us.get("replay_update"),
                  "host_latency_us_median": round(lat[len(lat) // 2] * 1e6, 1),
                  "host_latency_us_min": round(lat[0] * 1e6, 1), "dev_error": rep.debug_state()["dev_error"]}))
