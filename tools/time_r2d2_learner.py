#!/usr/bin/env python3
"""Per-kernel HIP-event timings of the isolated R2D2 learner step at C4's shape (B = 64, seq 80 / burn 40 / n 3).

  ITERS=10 [PRECISION=bf16x2] python tools/time_r2d2_learner.py      -> one line: kernel -> ms per step
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from rela_amd import _capi as capi
from rela_amd.learner import HipR2D2Learner
from test_r2d2_learner_gpu import _agent, _random_batch

A, B, SEQ, BURN, N = 18, int(os.environ.get("B", "64")), 80, 40, 3
ITERS = int(os.environ.get("ITERS", "10"))
rng = np.random.default_rng(5)
agent = _agent(A, N, 0.997, 0.9, SEQ, BURN, 71, 72, "cuda:0")
batch, weight = _random_batch(rng, A, B, SEQ, BURN, N, "cuda:0")
learner = HipR2D2Learner.from_agent(agent, B)
learner.set_precision(os.environ.get("PRECISION", "f32"))
for _ in range(2):
    learner.step(batch, weight)
learner.check()
torch.cuda.synchronize()
capi.lib.rela_prof_enable(1)
t0 = torch.cuda.Event(enable_timing=True)
t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(ITERS):
    learner.step(batch, weight)
t1.record()
learner.check()
capi.lib.rela_prof_enable(0)
buf = C.create_string_buffer(1 << 16)
capi.check(capi.lib.rela_prof_summary_json(buf, len(buf)), "prof")
prof = json.loads(buf.value.decode())
tab = {k: round(v["total_ms"] / ITERS, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
print(os.environ.get("TAG", ""), "ms/step", round(t0.elapsed_time(t1) / ITERS, 3), "kernels", round(sum(tab.values()), 3), tab, flush=True)
