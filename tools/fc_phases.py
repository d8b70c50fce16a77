#!/usr/bin/env python3
"""Where a position (64 of the 3136 contraction elements) of fc_bf16s spends its cycles: shader-clock stamps at the
phase boundaries (rela_ffnet_debug_fc_stamps; block 0, waves 0 and 7, positions 8..15)."""
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
rec = torch.randint(0, 120, (N, 49 * 256), dtype=torch.uint8, device="cuda")  # (small bf16 values: no inf / nan)
out = np.zeros((2, 8, 12), np.uint64)
for _ in range(3):
    capi.check(capi.lib.rela_ffnet_debug_fc_stamps(net.h, N, C.c_void_p(rec.data_ptr()), out.ctypes.data_as(C.c_void_p), None), "stamps")
names = ["MFMA loop (42 MFMAs, fragment reads, weight loads issued)", "store the tile of position + 2", "issue loads of position + 4", "barrier"]
for w, wave in enumerate((0, 7)):
    st = out[w].astype(np.int64)
    d = np.diff(st[:, :5], axis=1).mean(0)
    per = (st[1:, 0] - st[:-1, 0]).mean()
    print("wave %d: %.0f cycles per position (x 49 = %.0f)" % (wave, per, per * 49))
    for n_, v in zip(names, d):
        print("   %-58s %7.0f  (%.1f %%)" % (n_, v, 100 * v / per))
    print("   %-58s %7.0f" % ("(barrier -> next position's first MFMA: weight wait)", per - d.sum()))
