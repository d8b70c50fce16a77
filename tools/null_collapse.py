#!/usr/bin/env python3
"""Diagnosis: the threaded benchmark with the constant-frame env collapses (sample rate 1,200/s -> 40/s) once the
sampler has fed its own importance weights back as priorities for a while.  Reproduce it on the replay alone: a ring
of CONSTANT priorities, then sample / update_priority(weight) / add(constant) rounds, timing blocks of rounds and
printing the per-kernel averages of the slowest block."""
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

A, B = 18, 512
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
const = torch.full((rows,), float(os.environ.get("PRIO", "0.0043")), device="cuda")
while rep.size() + rows <= int(1.25 * cap):
    rep.add_rows(rows, ptrs, const)
torch.cuda.synchronize()
buf = C.create_string_buffer(1 << 16)
blocks = int(os.environ.get("BLOCKS", "12"))
per = int(os.environ.get("PER", "300"))
for b in range(blocks):
    capi.lib.rela_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(per):
        batch, w = rep.sample(B)
        rep.update_priority(w)
        rep.add_rows(6400, ptrs, const[:6400], nonblocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    capi.lib.rela_prof_enable(0)
    capi.check(capi.lib.rela_prof_summary_json(buf, len(buf)), "prof")
    prof = json.loads(buf.value.decode())
    us = {k: round(v["total_ms"] / per * 1e3, 1) for k, v in sorted(prof.items())}
    st = rep.debug_state()
    print(json.dumps({"block": b, "rounds_per_s": round(per / dt), "w_min_max": [float(w.min()), float(w.max())],
                      "sum": st["sum"], "kernels_us": us}), flush=True)
