"""R2D2 actor shard through the C ABI: the act-step reuse of post_step must be bit-identical to
recomputation (rela/r2d2_actor.h:221-302 with compute_priority of pyrela/r2d2.py:76-100)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias",
        "lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0", "fc_v.weight", "fc_v.bias",
        "fc_a.weight", "fc_a.bias"]


def _net(capi, A, seed):
    from synth import synth_lstm_params

    h = C.c_void_p()
    capi.check(capi.lib.rela_lstmnet_create(C.byref(h), A, 0), "rela_lstmnet_create")
    p, keep = capi.LSTMNetParams(), []
    params = synth_lstm_params(A, seed)
    for (field, _), k in zip(capi.LSTMNetParams._fields_, KEYS):
        a = np.ascontiguousarray(params[k], np.float32)
        keep.append(a)
        setattr(p, field, a.ctypes.data_as(C.c_void_p))
    capi.check(capi.lib.rela_lstmnet_load(h, C.byref(p), 0, None), "rela_lstmnet_load")
    return h, (p, keep)


def _run(mode):
    """mode: "reuse" | "switch" (set_reuse(0)) | "reload" (same weights re-loaded before every post_step)."""
    import torch

    from rela_amd import _capi as capi
    from rela_amd.engine import dev_view
    from synth import synth_obs

    R, K, A, n, seq, burn = 8, 4, 6, 2, 6, 2
    T = burn + seq + n
    online, keep_on = _net(capi, A, 1)
    target, _keep_tg = _net(capi, A, 2)
    replay = C.c_void_p()
    capi.check(capi.lib.rela_replay_create(C.byref(replay), 64, 7, 0.9, 0.6, 0, 0), "rela_replay_create")
    rb = (C.c_int64 * 10)(T * 28224, T * 4, T * 4 * A, T * 8, T * 4, T, T * 4, 2048, 2048, 4)
    st = (C.c_int32 * 10)(T, T, T, T, T, T, T, 1, 1, 1)
    capi.check(capi.lib.rela_replay_set_schema_seq(replay, 10, rb, st), "schema")
    actor = C.c_void_p()
    capi.check(capi.lib.rela_r2d2_actor_create(C.byref(actor), R, K, A, n, 0.99, seq, burn, 0.9, replay, 3, 0), "create")
    if mode == "switch":
        capi.check(capi.lib.rela_r2d2_actor_set_reuse(actor, 0), "set_reuse")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(5)
    prios, acts, nseq_total = [], [], 0
    for t in range(26):
        obs = synth_obs(R, 500 + t)
        eps = np.full(R, 0.0, np.float32)
        legal = (rng.uniform(size=(R, A)) < 0.8).astype(np.float32)
        legal[:, t % A] = 1.0
        act = np.zeros(R, np.int64)
        capi.check(capi.lib.rela_r2d2_actor_act(actor, online, obs.ctypes.data_as(C.c_void_p),
                                                eps.ctypes.data_as(C.c_void_p), legal.ctypes.data_as(C.c_void_p),
                                                act.ctypes.data_as(C.c_void_p), None, stream), "act")
        acts.append(act.copy())
        if mode == "reload":
            v0 = capi.lib.rela_lstmnet_version(online)
            capi.check(capi.lib.rela_lstmnet_load(online, C.byref(keep_on[0]), 0, None), "reload")
            assert capi.lib.rela_lstmnet_version(online) == v0 + 1
        reward = rng.integers(-1, 2, R).astype(np.float32)
        term = (rng.uniform(size=R) < 0.12).astype(np.uint8)
        ns = C.c_int(0)
        capi.check(capi.lib.rela_r2d2_actor_post_step(actor, reward.ctypes.data_as(C.c_void_p),
                                                      term.ctypes.data_as(C.c_void_p), online, target, 0, C.byref(ns),
                                                      stream), "post_step")
        nseq_total += ns.value
        torch.cuda.synchronize()
        p = dev_view(capi.lib.rela_r2d2_actor_last_priority_dev(actor), (R,), torch.float32, torch.device("cuda:0"))
        prios.append(p.cpu().numpy().copy())
    st_ = capi.ReplayState()
    capi.check(capi.lib.rela_replay_debug_state(replay, C.byref(st_), None, None, None), "state")
    w, ev = np.zeros(st_.ring, np.float32), np.zeros(st_.ring, np.uint8)
    capi.check(capi.lib.rela_replay_debug_weights(replay, w.ctypes.data_as(C.c_void_p), ev.ctypes.data_as(C.c_void_p)),
               "weights")
    size = capi.lib.rela_replay_size(replay)
    capi.lib.rela_r2d2_actor_destroy(actor)
    capi.lib.rela_replay_destroy(replay)
    capi.lib.rela_lstmnet_destroy(online)
    capi.lib.rela_lstmnet_destroy(target)
    return np.array(acts), np.array(prios), w, size, nseq_total


def test_post_step_reuses_the_act_step_bit_identically():
    base = _run("reuse")
    assert base[3] > 0 and base[4] == base[3]
    for mode in ("switch", "reload"):
        other = _run(mode)
        assert other[3] == base[3] and other[4] == base[4]
        assert np.array_equal(base[0], other[0])
        assert np.array_equal(base[1].view(np.uint32), other[1].view(np.uint32)), mode
        assert np.array_equal(base[2].view(np.uint32), other[2].view(np.uint32)), mode
