#!/usr/bin/env python3
"""Isolated Ape-X learner step at B = 512: GPU time of backward (3 forwards + loss + backward pass) and of apply
(clip + optimiser + repack of the kernel-layout weight copies), HIP events on the current stream.

  python tools/time_learner_phases.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from rela_amd.learner import HipApexLearner
from test_learner_gpu import make_agent, make_batch
agent = make_agent(18, 3)
learner = HipApexLearner.from_agent(agent, 512)
batch, w = make_batch(512, 18, 11)
for _ in range(3): learner.step(batch, w)
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
N = 50
tb = ta = 0.0
for _ in range(N):
    e0.record(); learner.backward(batch, w); e1.record(); learner.apply(); e2.record()
    torch.cuda.synchronize()
    tb += e0.elapsed_time(e1); ta += e1.elapsed_time(e2)
print("backward %.3f ms, apply (clip + optimiser + repack) %.3f ms" % (tb / N, ta / N))
