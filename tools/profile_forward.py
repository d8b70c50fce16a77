#!/usr/bin/env python3
"""Isolated AtariFFNet forward (N = 6400) and replay sample loop for rocprofv3 passes.

  rocprofv3 --kernel-trace --stats ...  -- python3 tools/profile_forward.py
  rocprofv3 --pmc FETCH_SIZE ...        -- python3 tools/profile_forward.py      (one pass per counter set)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from rela_amd import _capi as capi
from rela_amd.engine import FFNetHandle
from synth import synth_params

N, A, ITERS = 6400, 18, int(os.environ.get("ITERS", "10"))
net = FFNetHandle(A, "cuda:0")
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_params(A, 1).items()})
net.set_precision(os.environ.get("PRECISION", "bf16x2"))  # bench.py's default; PRECISION=f32 for the parity mode
s = torch.randint(0, 256, (N, 4, 84, 84), dtype=torch.uint8, device="cuda")
legal = torch.ones((N, A), device="cuda")
q = torch.empty((N, A), device="cuda")
nb = capi.lib.rela_ffnet_workspace_bytes(net.h, N)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(ITERS):
    capi.check(capi.lib.rela_ffnet_forward(net.h, N, C.c_void_p(s.data_ptr()), C.c_void_p(legal.data_ptr()),
                                           C.c_void_p(q.data_ptr()), C.c_void_p(ws.data_ptr()), nb, stream), "fwd")
torch.cuda.synchronize()

if os.environ.get("WITH_REPLAY", "1") == "1":
    from rela_amd.replay import FFReplay

    cap = 1 << 20
    rep = FFReplay(cap, 7, 0.6, 0.4, 0, A, "cuda:0")
    rows = 65536
    obs = torch.randint(0, 256, (rows, 4, 84, 84), dtype=torch.uint8, device="cuda")
    z = torch.zeros(rows, device="cuda")
    zi = torch.zeros(rows, dtype=torch.int64, device="cuda")
    zb = torch.zeros(rows, dtype=torch.uint8, device="cuda")
    eps = torch.zeros(rows, 1, device="cuda")
    lg = torch.ones(rows, A, device="cuda")
    while rep.size() + rows <= int(1.25 * cap):
        pr = torch.rand(rows, device="cuda") * 2 + 0.01
        rep.add_rows(rows, [obs.data_ptr(), obs.data_ptr(), eps.data_ptr(), eps.data_ptr(), lg.data_ptr(), lg.data_ptr(),
                            zi.data_ptr(), z.data_ptr(), zb.data_ptr(), z.data_ptr()], pr)
    for _ in range(ITERS):
        batch, w = rep.sample(512)
        rep.update_priority(torch.rand(512, device="cuda") + 0.1)
    torch.cuda.synchronize()
    print("replay size", rep.size(), rep.debug_state()["dev_error"])
print("done")
