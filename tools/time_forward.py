#!/usr/bin/env python3
"""Per-kernel HIP-event timings of an isolated AtariFFNet forward (N envs): tuning aid.

  N=6400 ITERS=20 python tools/time_forward.py      -> one line: kernel -> average us
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from rela_amd import _capi as capi
from rela_amd.engine import FFNetHandle
from synth import synth_params

N, A, ITERS = int(os.environ.get("N", "6400")), 18, int(os.environ.get("ITERS", "20"))
net = FFNetHandle(A, "cuda:0")
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_params(A, 1).items()})
net.set_precision(os.environ.get("PRECISION", "f32"))  # f32 | bf16x2
s = torch.randint(0, 256, (N, 4, 84, 84), dtype=torch.uint8, device="cuda")
legal = torch.ones((N, A), device="cuda")
q = torch.empty((N, A), device="cuda")
nb = capi.lib.rela_ffnet_workspace_bytes(net.h, N)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def fwd():
    capi.check(capi.lib.rela_ffnet_forward(net.h, N, C.c_void_p(s.data_ptr()), C.c_void_p(legal.data_ptr()),
                                           C.c_void_p(q.data_ptr()), C.c_void_p(ws.data_ptr()), nb, stream), "fwd")


for _ in range(3):
    fwd()
torch.cuda.synchronize()
capi.lib.rela_prof_enable(1)
for _ in range(ITERS):
    fwd()
torch.cuda.synchronize()
capi.lib.rela_prof_enable(0)
buf = C.create_string_buffer(1 << 16)
capi.check(capi.lib.rela_prof_summary_json(buf, len(buf)), "prof")
prof = json.loads(buf.value.decode())
print(os.environ.get("TAG", ""), {k: round(v["total_ms"] / v["count"] * 1e3, 1) for k, v in prof.items()}, flush=True)
