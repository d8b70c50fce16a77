#!/usr/bin/env python3
"""Where a group (two frames) of conv3 of the split-bf16 mode spends its cycles: shader-clock stamps at the phase
boundaries (rela_ffnet_debug_conv3_stamps; block 0, waves 0 and 7, groups 2..7)."""
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
rec = torch.randint(0, 120, (N, 81 * 256), dtype=torch.uint8, device="cuda")  # (small bf16 values: no inf / nan)
out = np.zeros((2, 8, 12), np.uint64)
for _ in range(3):
    capi.check(capi.lib.rela_ffnet_debug_conv3_stamps(net.h, N, C.c_void_p(rec.data_ptr()), out.ctypes.data_as(C.c_void_p), None), "stamps")
names = ["MFMA loop (216 MFMAs, fragment reads, next group's staging)", "epilogue (bias, ReLU, split, LDS stores)", "barrier", "copy-out issue"]
for w, wave in enumerate((0, 7)):
    st = out[w].astype(np.int64)[:, :5]
    d = np.diff(st[2:8], axis=1).mean(0)
    per = (st[3:8, 0] - st[2:7, 0]).mean()
    print("wave %d: %.0f cycles per group of two frames" % (wave, per))
    for n_, v in zip(names, d):
        print("   %-62s %7.0f  (%.1f %%)" % (n_, v, 100 * v / per))
    print("   %-62s %7.0f" % ("(copy-out -> next group start)", per - d.sum()))
