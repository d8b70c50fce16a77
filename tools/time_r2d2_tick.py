#!/usr/bin/env python3
"""R2D2 actor tick (rela_r2d2_actor_act + rela_r2d2_actor_post_step through the C ABI) at the scale of
BASELINE config C4: `ROWS` envs in groups of K, seq_len 80 / burn_in 40 / n 3 (window T = 123 slots per
env in HBM), AtariLSTMNet with A = 18, sequences emitted into a device-resident RNN replay.

  ROWS=3200 TICKS=260 python tools/time_r2d2_tick.py    -> env-steps/s, ms per tick, sequences inserted
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from rela_amd import _capi as capi
from synth import synth_lstm_params

KEYS = ["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias",
        "lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0",
        "fc_v.weight", "fc_v.bias", "fc_a.weight", "fc_a.bias"]
ROWS, K, A = int(os.environ.get("ROWS", "3200")), 80, 18
N_STEP, GAMMA, SEQ, BURN, ETA = 3, 0.997, 80, 40, 0.9
TICKS = int(os.environ.get("TICKS", "260"))
EPISODE = int(os.environ.get("EPISODE", "400"))
CAP = int(os.environ.get("CAP", "8192"))  # ring 10,240 sequences x 3.47 MB; one pop of all envs must fit
T = BURN + SEQ + N_STEP


def make_net(seed):
    h = C.c_void_p()
    capi.check(capi.lib.rela_lstmnet_create(C.byref(h), A, 0), "rela_lstmnet_create")
    p, keep = capi.LSTMNetParams(), []
    params = synth_lstm_params(A, seed)
    for (field, _), k in zip(capi.LSTMNetParams._fields_, KEYS):
        a = np.ascontiguousarray(params[k], np.float32)
        keep.append(a)
        setattr(p, field, a.ctypes.data_as(C.c_void_p))
    capi.check(capi.lib.rela_lstmnet_load(h, C.byref(p), 0, None), "rela_lstmnet_load")
    return h


online, target = make_net(1), make_net(2)
PRECISION = os.environ.get("PRECISION", "f32")  # f32 | bf16x2 | f32x3 (rela_lstmnet_set_precision)
for h_ in (online, target):
    capi.check(capi.lib.rela_lstmnet_set_precision(h_, {"f32": 0, "bf16x2": 1, "f32x3": 2}[PRECISION]), "set_precision")
replay = C.c_void_p()
capi.check(capi.lib.rela_replay_create(C.byref(replay), CAP, 7, 0.9, 0.6, 0, 0), "rela_replay_create")
rb = (C.c_int64 * 10)(T * 28224, T * 4, T * 4 * A, T * 8, T * 4, T, T * 4, 2048, 2048, 4)
st = (C.c_int32 * 10)(T, T, T, T, T, T, T, 1, 1, 1)
capi.check(capi.lib.rela_replay_set_schema_seq(replay, 10, rb, st), "rela_replay_set_schema_seq")
actor = C.c_void_p()
capi.check(capi.lib.rela_r2d2_actor_create(C.byref(actor), ROWS, K, A, N_STEP, GAMMA, SEQ, BURN, ETA, replay, 11, 0),
           "rela_r2d2_actor_create")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(0)
eps = np.full(ROWS, 0.05, np.float32)
legal = np.ones((ROWS, A), np.float32)
reward = rng.integers(-1, 2, ROWS).astype(np.float32)
# episodes end at staggered times so that sequence emission is spread over the ticks
phase = rng.integers(0, EPISODE, ROWS)
nseq_total, first = 0, True
times = []
for tick in range(TICKS + 20):
    if tick == 20:
        torch.cuda.synchronize()
        capi.lib.rela_prof_enable(1)
        t0 = time.perf_counter()
        nseq_total = 0
    # frames stay resident in HBM: the slot keeps whatever the previous pass over the ring left there
    # (timing only; the first passes see zeros)
    capi.check(capi.lib.rela_r2d2_actor_act(actor, online, None, eps.ctypes.data_as(C.c_void_p) if first else None,
                                            legal.ctypes.data_as(C.c_void_p) if first else None, None, None, stream),
               "rela_r2d2_actor_act")
    first = False
    term = ((tick + phase) % EPISODE == EPISODE - 1).astype(np.uint8)
    nseq = C.c_int(0)
    rc = capi.lib.rela_r2d2_actor_post_step(actor, reward.ctypes.data_as(C.c_void_p), term.ctypes.data_as(C.c_void_p),
                                            online, target, 1, C.byref(nseq), stream)
    if rc != capi.EWOULDBLOCK:
        capi.check(rc, "rela_r2d2_actor_post_step")
    nseq_total += nseq.value
torch.cuda.synchronize()
dt = time.perf_counter() - t0
capi.lib.rela_prof_enable(0)
buf = C.create_string_buffer(1 << 16)
capi.check(capi.lib.rela_prof_summary_json(buf, len(buf)), "prof")
prof = json.loads(buf.value.decode())
print(json.dumps({"workload": "R2D2 actor tick, %d envs (K=%d), seq %d / burn %d / n %d, A=%d" % (ROWS, K, SEQ, BURN, N_STEP, A),
                  "env_steps_per_s": ROWS * TICKS / dt, "ms_per_tick": dt / TICKS * 1e3, "ticks": TICKS,
                  "sequences_inserted": nseq_total, "replay_size": capi.lib.rela_replay_size(replay),
                  "kernels_ms_per_tick": {k: round(v["total_ms"] / TICKS, 4) for k, v in sorted(prof.items())}}))
capi.lib.rela_r2d2_actor_destroy(actor)
capi.lib.rela_replay_destroy(replay)
capi.lib.rela_lstmnet_destroy(online)
capi.lib.rela_lstmnet_destroy(target)
