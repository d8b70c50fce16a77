#!/usr/bin/env python3
"""Where a frame of the fused conv1 -> conv2 kernel spends its cycles: shader-clock stamps at the phase boundaries
(rela_ffnet_debug_conv12_stamps; block 0, waves 0 and 7, frames 2..7 averaged)."""
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

N, A = int(os.environ.get("N", "6400")), 18
net = FFNetHandle(A, "cuda:0")
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_params(A, 1).items()})
net.set_precision("bf16x2")
s = torch.randint(0, 256, (N, 4, 84, 84), dtype=torch.uint8, device="cuda")
out = np.zeros((2, 8, 12), np.uint64)
for _ in range(3):
    capi.check(capi.lib.rela_ffnet_debug_conv12_stamps(net.h, N, C.c_void_p(s.data_ptr()), out.ctypes.data_as(C.c_void_p), None), "stamps")
I8 = True  # (the bf16 half-frame kernel was removed in r4)
names8 = ["conv1 (3 passes: MFMAs, epilogues, previous tile's copy-out)", "barrier", "conv2 MFMA loop (+ loads, staging stores)", "conv2 epilogue", "barrier"]
names = ["conv1 half 0 (MFMA + epilogue)", "barrier", "convert half 1 + issue loads", "barrier", "conv1 half 1", "barrier",
         "conv2 MFMA loop (+ a1 copy-out in job form)", "convert next half 0 + issue loads", "conv2 epilogue", "barrier", "copy-out issue"]
for w, wave in enumerate((0, 7)):
    st = out[w].astype(np.int64)
    if I8:
        st = st[:, :6]
    d = np.diff(st[2:8], axis=1).mean(0)
    frame = (st[3:8, 0] - st[2:7, 0]).mean()
    print("wave %d: %.0f cycles per frame" % (wave, frame))
    for n_, v in zip(names8 if I8 else names, d):
        print("   %-46s %7.0f  (%.1f %%)" % (n_, v, 100 * v / frame))
    print("   %-46s %7.0f" % ("(copy-out end -> next frame start)", frame - d.sum()))
