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


def _run(mode, R=8, K=4, precision="f32"):
    """mode: "reuse" | "switch" (set_reuse(0)) | "reload" (same weights re-loaded before every post_step)."""
    import torch

    from rela_amd import _capi as capi
    from rela_amd.engine import dev_view
    from synth import synth_obs

    A, n, seq, burn = 6, 2, 6, 2
    T = burn + seq + n
    online, keep_on = _net(capi, A, 1)
    target, _keep_tg = _net(capi, A, 2)
    for net in (online, target):
        capi.check(capi.lib.rela_lstmnet_set_precision(net, {"f32": 0, "bf16x2": 1, "f32x3": 2}[precision]), "set_precision")
    replay = C.c_void_p()
    capi.check(capi.lib.rela_replay_create(C.byref(replay), 8 * R, 7, 0.9, 0.6, 0, 0), "rela_replay_create")
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


@pytest.mark.parametrize("R,K,precision", [(8, 4, "f32"), (128, 64, "f32"), (128, 64, "bf16x2")])
def test_post_step_reuses_the_act_step_bit_identically(R, K, precision):
    """(r4: also at 128 rows in BOTH precision modes: from 128 rows up the bf16x2 LSTM net runs its conv trunk on the
    split-bf16 kernels -- asserted through the launch census -- and the memoised step must still be bit-identical)"""
    from kernel_names import CONV12
    from rela_amd import _capi as capi

    with capi.launch_census() as census:
        base = _run("reuse", R, K, precision)
    fast = {CONV12, "conv_bf16s<Conv3F>"}
    ran = set(census.counts)
    assert (fast <= ran) if (precision == "bf16x2" and R >= 128) else not (fast & ran), (precision, sorted(ran))
    assert base[3] > 0 and base[4] == base[3]
    for mode in ("switch", "reload"):
        other = _run(mode, R, K, precision)
        assert other[3] == base[3] and other[4] == base[4]
        assert np.array_equal(base[0], other[0])
        assert np.array_equal(base[1].view(np.uint32), other[1].view(np.uint32)), mode
        assert np.array_equal(base[2].view(np.uint32), other[2].view(np.uint32)), mode


def test_device_windows_at_c4_shape_match_the_oracle_buffer():
    """BASELINE config C4's window shape (seq 80 / burn-in 40 / n 3: 123 slots, 3.47 MB of frames per
    sequence) on the device: every sequence the R2D2 actor shard appends to the replay -- slot contents,
    front / tail padding, carry-over, second short sequence after a terminal in the carried region, length,
    aggregated priority -- against the oracle's R2D2TransitionBuffer (oracle/r2d2_oracle.c, pinned to traces
    of the real rela::R2D2TransitionBuffer incl. r2d2buf_k1_n3_s80_b40.json) fed the same per-step
    terminals and the device's own per-step priorities.  Frames carry their (env, step) tag in the first
    four bytes, so a misplaced row of the 3.47 MB windows cannot go unnoticed."""
    import torch

    from oracle_lib import h2f, load
    from rela_amd import _capi as capi
    from rela_amd.engine import dev_view
    from synth import synth_obs
    from test_oracle_r2d2 import OracleR2D2Buf, bind

    R, K, A, n, seq, burn, gamma, eta = 3, 3, 6, 3, 80, 40, 0.997, 0.9
    T = burn + seq + n
    steps = 520
    online, _k1 = _net(capi, A, 1)
    target, _k2 = _net(capi, A, 2)
    replay = C.c_void_p()
    capi.check(capi.lib.rela_replay_create(C.byref(replay), 64, 7, 1.0, 0.6, 0, 0), "rela_replay_create")
    rb = (C.c_int64 * 10)(T * 28224, T * 4, T * 4 * A, T * 8, T * 4, T, T * 4, 2048, 2048, 4)
    st = (C.c_int32 * 10)(T, T, T, T, T, T, T, 1, 1, 1)
    capi.check(capi.lib.rela_replay_set_schema_seq(replay, 10, rb, st), "schema")
    actor = C.c_void_p()
    capi.check(capi.lib.rela_r2d2_actor_create(C.byref(actor), R, K, A, n, gamma, seq, burn, eta, replay, 3, 0), "create")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(80)
    # terminals: env 0 ends after 50 steps (episode shorter than the window), then at step 50 + 121 (terminal
    # inside the carried region of its second window); env 1 never before step 300; env 2 random
    term_all = np.zeros((steps, R), np.uint8)
    term_all[49, 0] = term_all[170, 0] = term_all[400, 0] = 1
    term_all[300, 1] = 1
    term_all[:, 2] = rng.uniform(size=steps) < 0.012
    rew_all = rng.integers(-1, 2, (steps, R)).astype(np.float32)
    acts = np.zeros((steps, R), np.int64)
    lib = bind(load())
    obuf = OracleR2D2Buf(R, n, seq, burn)
    n_r, n_b, n_t = (np.zeros((steps, R), np.float32), np.zeros((steps, R), np.float32), np.zeros((steps, R), np.uint8))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    checked, kinds, tail = 0, set(), 0
    base = synth_obs(R, 4242)
    for t in range(steps):
        obs = base.copy()
        tagv = (np.arange(R) + t * 100).astype(np.uint32)
        obs.reshape(R, -1)[:, :4] = tagv.view(np.uint8).reshape(R, 4)
        obs.reshape(R, -1)[:, 4:64] = rng.integers(0, 256, (R, 60), dtype=np.uint8)  # the net must see fresh frames
        eps = np.zeros(R, np.float32)
        legal = np.ones((R, A), np.float32)
        act = np.zeros(R, np.int64)
        capi.check(capi.lib.rela_r2d2_actor_act(actor, online, vp(obs), vp(eps), vp(legal), vp(act), None, stream), "act")
        acts[t] = act
        ns = C.c_int(0)
        capi.check(capi.lib.rela_r2d2_actor_post_step(actor, vp(rew_all[t]), vp(term_all[t]), online, target, 0,
                                                      C.byref(ns), stream), "post_step")
        if t < n:
            assert ns.value == 0
            continue
        s = t - n  # the transition popped from the n-step buffer on this tick (dqn_actor.h:58-106)
        lib.oracle_nstep_pop(n, R, C.c_float(gamma), rew_all[s:s + n + 1].ctypes.data_as(C.POINTER(C.c_float)),
                             term_all[s:s + n + 1].ctypes.data_as(C.POINTER(C.c_uint8)),
                             n_r[s].ctypes.data_as(C.POINTER(C.c_float)), n_b[s].ctypes.data_as(C.POINTER(C.c_float)),
                             n_t[s].ctypes.data_as(C.POINTER(C.c_uint8)))
        torch.cuda.synchronize()
        p = dev_view(capi.lib.rela_r2d2_actor_last_priority_dev(actor), (R,), torch.float32,
                     torch.device("cuda:0")).cpu().numpy().copy()
        can = obuf.push(n_t[s], p)
        assert can == (ns.value > 0), (t, can, ns.value)
        if not can:
            continue
        exp = obuf.pop()
        assert len(exp) == ns.value
        q = len(exp)
        fr = np.zeros((q, T * 28224), np.uint8)
        a_ = np.zeros((q, T), np.int64)
        r_, b_ = np.zeros((q, T), np.float32), np.zeros((q, T), np.float32)
        t_ = np.zeros((q, T), np.uint8)
        ln = np.zeros(q, np.float32)
        for f, arr in ((0, fr), (3, a_), (4, r_), (5, t_), (6, b_), (9, ln)):
            capi.check(capi.lib.rela_replay_debug_read_rows(replay, f, tail, q, vp(arr)), "read_rows")
        st_ = capi.ReplayState()
        capi.check(capi.lib.rela_replay_debug_state(replay, C.byref(st_), None, None, None), "state")
        w = np.zeros(st_.ring, np.float32)
        capi.check(capi.lib.rela_replay_debug_weights(replay, vp(w), None), "weights")
        for i, e in enumerate(exp):
            env = e["env"]
            real = np.array(e["reward"], np.float32) > 0  # the oracle tags real slots with reward = step + 0.5
            step_of = (np.array(e["reward"], np.float32) - 0.5).astype(np.int64)
            got_tag = fr[i].reshape(T, 28224)[:, :4].copy().view(np.uint32).reshape(T)
            for j in range(T):
                if real[j]:
                    sj = int(step_of[j])
                    assert got_tag[j] == env + 100 * sj, (t, i, j)
                    assert a_[i, j] == acts[sj, env] and t_[i, j] == n_t[sj, env]
                    assert r_[i, j] == n_r[sj, env] and b_[i, j] == n_b[sj, env]
                else:  # padLike (types.cc:69-80): zeros, terminal = 1
                    assert got_tag[j] == 0 and not fr[i].reshape(T, 28224)[j].any()
                    assert a_[i, j] == 0 and r_[i, j] == 0 and b_[i, j] == 0 and t_[i, j] == 1
            assert e["terminal"] == t_[i].tolist()
            assert ln[i] == e["len"]
            prow = np.array([h2f(v) for v in e["prio"]], np.float32)[None]
            agg = np.zeros(1, np.float32)
            lib.oracle_r2d2_aggregate(1, seq, burn, C.c_float(eta), vp(prow), vp(np.array([e["len"]], np.float32)), vp(agg))
            np.testing.assert_allclose(w[tail + i], agg[0], rtol=2e-6)
            kinds.add("short" if e["len"] < burn + seq else "full")
            checked += 1
        tail += q
    assert checked >= 12 and kinds == {"short", "full"} and tail < 64
    capi.lib.rela_r2d2_actor_destroy(actor)
    capi.lib.rela_replay_destroy(replay)
    capi.lib.rela_lstmnet_destroy(online)
    capi.lib.rela_lstmnet_destroy(target)
