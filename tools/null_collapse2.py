#!/usr/bin/env python3
"""Finer diagnosis of tools/null_collapse.py: print the first rounds one by one."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from rela_amd import _capi as capi
from rela_amd.replay import FFReplay

A, B = 18, 512
cap = int(os.environ.get("CAP", str(1 << 16)))
prio = float(os.environ.get("PRIO", "0.0043"))
feed = os.environ.get("FEED", "weight")  # what update_priority gets: the IS weights, or ones
rep = FFReplay(cap, 7, 0.6, 0.4, 0, A, "cuda:0")
rows = int(os.environ.get("ROWS", "4096"))
add_rows = int(os.environ.get("ADD", str(rows)))
obs = torch.randint(0, 256, (rows, 4, 84, 84), dtype=torch.uint8, device="cuda")
z = torch.zeros(rows, device="cuda")
zi = torch.zeros(rows, dtype=torch.int64, device="cuda")
zb = torch.zeros(rows, dtype=torch.uint8, device="cuda")
eps = torch.zeros(rows, 1, device="cuda")
lg = torch.ones(rows, A, device="cuda")
ptrs = [obs.data_ptr(), obs.data_ptr(), eps.data_ptr(), eps.data_ptr(), lg.data_ptr(), lg.data_ptr(), zi.data_ptr(),
        z.data_ptr(), zb.data_ptr(), z.data_ptr()]
const = torch.full((rows,), prio, device="cuda")
while rep.size() + rows <= int(1.25 * cap):
    rep.add_rows(rows, ptrs, const)
torch.cuda.synchronize()
print("filled", rep.debug_state())
for r in range(int(os.environ.get("ROUNDS", "12"))):
    batch, w = rep.sample(B)
    torch.cuda.synchronize()
    st = capi.ReplayState()
    ids = np.zeros(B, np.int32)
    raw = np.zeros(B, np.float32)
    capi.check(capi.lib.rela_replay_debug_state(rep.h, C.byref(st), ids.ctypes.data_as(C.c_void_p),
                                                raw.ctypes.data_as(C.c_void_p), None), "debug_state")
    wc = w.cpu().numpy()
    print(json.dumps({"round": r, "w": [float(np.nanmin(wc)), float(np.nanmax(wc)), int(np.isnan(wc).sum())],
                      "raw": [float(raw.min()), float(raw.max()), int((raw == 0).sum())],
                      "ids": [int(ids.min()), int(ids.max()), int(len(set(ids.tolist())))], "sum": st.sum,
                      "size": st.size, "dev_error": st.dev_error}), flush=True)
    rep.update_priority(w if feed == "weight" else torch.ones(B, device="cuda"))
    rep.add_rows(add_rows, ptrs, const[:add_rows], nonblocking=True)
    torch.cuda.synchronize()
