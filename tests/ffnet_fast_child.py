"""Child process of test_ffnet_gpu.py::test_ffnet_fast_mode_fusion_variants: the conv1 -> conv2 variant is chosen
once per process (RELA_FUSE12), so each variant is checked in a process of its own.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from synth import synth_obs, synth_params
from test_ffnet_gpu import GpuNet

N, A = int(sys.argv[1]), 18
net = GpuNet(synth_params(A, 77), A)
s = synth_obs(N, 3000 + N)
legal = np.ones((N, A), np.float32)
q_ref = net.forward(s, legal).cpu().numpy()
net.capi.check(net.capi.lib.rela_ffnet_set_precision(net.h, 1), "set_precision")
q_fast = net.forward(s, legal).cpu().numpy()
tmo = C.c_uint(7)
net.capi.check(net.capi.lib.rela_ffnet_debug_pipe_timeout(net.h, C.byref(tmo)), "pipe_timeout")
print(json.dumps({"max_err": float(np.abs(q_fast - q_ref).max()), "timeout": int(tmo.value),
                  "agree": float((q_fast.argmax(1) == q_ref.argmax(1)).mean())}))
net.close()
