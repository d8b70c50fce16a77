"""GPU parity tests for the small actor-side ops (n-step return, eps-greedy act, TD priority),
through the C ABI, against the golden vectors of the real reference and the CPU oracle."""
import ctypes as C
import glob
import json
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def h2f(h):
    return struct.unpack(">f", bytes.fromhex(h))[0]


def f2h(x):
    return struct.pack(">f", float(np.float32(x))).hex()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "nstep_*.json"))), ids=os.path.basename)
def test_nstep_golden(path):
    """rela_nstep_return vs MultiStepTransitionBuffer::popTransition (dqn_actor.h:58-106): bit-exact."""
    import torch

    from gpu_util import cur_stream, dev, ptr
    from rela_amd import _capi as capi

    case = json.load(open(path))
    n, K, gamma = case["multi_step"], case["K"], h2f(case["gamma"])
    rh, th = [], []
    for line, exp in zip(case["script"][1:], case["expect"][1:]):
        tok = line.split()[1:]
        rh.append([h2f(t) for t in tok[:K]])
        th.append([int(t) for t in tok[K:]])
        if not exp["pop"]:
            continue
        # store the deque as a ring rotated by `first` to exercise the first_row argument
        first = len(rh) % (n + 1) if n > 0 else 0
        first = (first + exp["obs_step"]) % (n + 1)
        rot = lambda a: np.roll(np.array(a), first, axis=0)
        r, t = dev(rot(rh).astype(np.float32)), dev(rot(th).astype(np.uint8))
        o_r = torch.empty(K, device="cuda")
        o_b = torch.empty(K, device="cuda")
        o_t = torch.empty(K, dtype=torch.uint8, device="cuda")
        capi.check(capi.lib.rela_nstep_return(n, K, gamma, first, ptr(r), ptr(t), ptr(o_r), ptr(o_b), ptr(o_t), cur_stream()),
                   "rela_nstep_return")
        assert [f2h(x) for x in o_r.cpu().numpy()] == exp["reward"]
        assert o_b.cpu().numpy().tolist() == exp["bootstrap"]
        assert o_t.cpu().numpy().tolist() == exp["terminal"]
        rh.pop(0)
        th.pop(0)


@pytest.mark.parametrize("n,A,masked", [(1, 18, False), (80, 18, False), (80, 6, True), (6400, 18, False)])
def test_act_and_td_from_q(n, A, masked):
    """Greedy branch (eps = 0) and TD priority vs the oracle (apex.py:30-78): exact, same f32 ops."""
    import torch

    from gpu_util import cur_stream, dev, ptr
    from oracle_lib import load
    from rela_amd import _capi as capi

    lib = load()
    rng = np.random.default_rng(n * 100 + A)
    q = rng.normal(0, 1, (n, A)).astype(np.float32)
    qn = rng.normal(0, 1, (n, A)).astype(np.float32)
    qt = rng.normal(0, 1, (n, A)).astype(np.float32)
    legal = np.ones((n, A), np.float32)
    if masked:
        legal[:, 1::2] = 0
    action = rng.integers(0, A, n).astype(np.int64)
    reward = rng.integers(-1, 2, n).astype(np.float32)
    boot = (rng.uniform(size=n) < 0.8).astype(np.float32)
    gamma_n = np.float32(0.997 ** 3)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    # oracle greedy
    exp_act = np.zeros(n, np.int64)
    lib.oracle_greedy.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int64)]
    lib.oracle_greedy(n, A, fp(q), fp(legal), exp_act.ctypes.data_as(C.POINTER(C.c_int64)))
    d_act = torch.empty(n, dtype=torch.int64, device="cuda")
    # keep every device tensor alive in a local: a temporary would be freed (and its block
    # reused by the next allocation) before the kernel runs
    d_q, d_qn, d_qt, d_legal, d_eps = dev(q), dev(qn), dev(qt), dev(legal), dev(np.zeros(n, np.float32))
    d_action, d_reward, d_boot = dev(action), dev(reward), dev(boot)
    capi.check(capi.lib.rela_apex_act_from_q(n, A, 0, ptr(d_q), ptr(d_legal), ptr(d_eps), 1, 0, ptr(d_act), cur_stream()),
               "act")
    np.testing.assert_array_equal(d_act.cpu().numpy(), exp_act)
    # oracle td from the three tables
    na = np.zeros(n, np.int64)
    lib.oracle_greedy(n, A, fp(qn), fp(legal), na.ctypes.data_as(C.POINTER(C.c_int64)))
    qa = q[np.arange(n), action]
    bq = qt[np.arange(n), na]
    tgt = (reward + ((boot * gamma_n).astype(np.float32) * bq).astype(np.float32)).astype(np.float32)
    err = (tgt - qa).astype(np.float32)
    d_td = torch.empty(n, device="cuda")
    d_pr = torch.empty(n, device="cuda")
    capi.check(capi.lib.rela_apex_td_from_q(n, A, 0, ptr(d_q), ptr(d_qn), ptr(d_qt), ptr(d_legal), ptr(d_action),
                                            ptr(d_reward), ptr(d_boot), C.c_float(gamma_n), ptr(d_td), ptr(d_pr),
                                            cur_stream()), "td")
    np.testing.assert_array_equal(d_td.cpu().numpy(), err)
    np.testing.assert_array_equal(d_pr.cpu().numpy(), np.abs(err))


def test_eps_greedy_statistics():
    """Random branch: statistical parity only.  With eps = 0.5 about half the rows
    deviate from greedy towards a uniformly random LEGAL action; never an illegal one."""
    import torch

    from gpu_util import cur_stream, dev, ptr
    from rela_amd import _capi as capi

    n, A = 4096, 18
    rng = np.random.default_rng(5)
    q = rng.normal(0, 1, (n, A)).astype(np.float32)
    legal = np.ones((n, A), np.float32)
    legal[:, ::3] = 0
    eps = np.full(n, 0.5, np.float32)
    acts = []
    d_q, d_legal, d_eps = dev(q), dev(legal), dev(eps)
    for off in (0, n):
        d_act = torch.empty(n, dtype=torch.int64, device="cuda")
        capi.check(capi.lib.rela_apex_act_from_q(n, A, 0, ptr(d_q), ptr(d_legal), ptr(d_eps), 99, off, ptr(d_act),
                                                 cur_stream()), "act")
        acts.append(d_act.cpu().numpy())
    greedy = np.argmax((1 + q - q.min()) * legal, 1)
    for a in acts:
        assert (legal[np.arange(n), a] == 1).all()
        frac = (a != greedy).mean()
        assert 0.5 * (11 / 12) - 0.05 < frac < 0.5 * (11 / 12) + 0.05
    assert (acts[0] != acts[1]).any()  # different Philox offsets -> different draws
    hist = np.bincount(acts[0][acts[0] != greedy], minlength=A)[legal[0] == 1]
    assert hist.min() > 0.5 * hist.mean()


@pytest.mark.parametrize("A,R,K,precision", [(6, 16, 8, "f32"), (18, 256, 128, "f32"), (18, 256, 128, "bf16x2")])
def test_post_step_reuses_act_forward_bit_identically(A, R, K, precision):
    """(r4: also at 256 rows in groups of 128 in BOTH precision modes -- the memoised Q tables of the bf16x2 engine
    at >= 128 rows are what the bench's fast mode runs; the launch census asserts the split-bf16 kernels ran.)
    post_step skips the online forwards on obs and next_obs when act() already ran them with the same
    weights (dqn_actor.h:84,161 and apex.py:38,41 evaluate the same network on the same batches, n ticks ago
    and this tick).  Twin actors -- one whose weights are re-loaded (same values, new version) before every
    post_step, one with the reuse switched off, one that reuses only the next_obs forward -- must take the
    recompute paths and end with bit-identical priorities and replay weights.  All of them get NEW online
    weights before tick 4, so the reusing actor has to recompute online(obs) for the n ticks that straddle it."""
    import torch

    from rela_amd import _capi as capi
    from rela_amd.engine import ApexActorEngine, FFNetHandle
    from rela_amd.replay import FFReplay
    from synth import synth_obs, synth_params

    from kernel_names import CONV12

    n = 3
    params = {k: torch.from_numpy(v) for k, v in synth_params(A, 5).items()}
    tparams = {k: torch.from_numpy(v) for k, v in synth_params(A, 6).items()}
    params2 = {k: torch.from_numpy(v) for k, v in synth_params(A, 7).items()}
    eps = np.linspace(0.0, 0.4, R).astype(np.float32)
    runs = []
    for reload_between in (False, True, "switch", "next_only"):
        online, target = FFNetHandle(A, "cuda:0"), FFNetHandle(A, "cuda:0")
        online.load_state_dict(params)
        target.load_state_dict(tparams)
        online.set_precision(precision)
        target.set_precision(precision)
        replay = FFReplay(16 * R, 3, 0.6, 0.4, 0, A, "cuda:0")
        eng = ApexActorEngine(R, K, A, n, 0.99, replay, eps, "cuda:0", seed=11)
        capi.check(capi.lib.rela_prof_count_enable(1), "census")
        eng.legal.fill_(1.0)
        if reload_between == "switch":
            eng.set_reuse(False)
        if reload_between == "next_only":
            eng.set_reuse(2)
        prios, acts = [], []
        cur = params
        for t in range(n + 8):
            if t == 4:
                cur = params2
                online.load_state_dict(cur)
            eng.next_obs_slot().copy_(torch.from_numpy(synth_obs(R, 100 + t)).cuda())
            acts.append(eng.act(online).cpu().numpy().copy())
            v0 = capi.lib.rela_ffnet_version(online.h)
            if reload_between is True:
                online.load_state_dict(cur)
                assert capi.lib.rela_ffnet_version(online.h) == v0 + 1
            rew = torch.full((R,), 0.25 * t, device="cuda")
            term = torch.zeros(R, dtype=torch.uint8, device="cuda")
            if eng.post_step(rew, term, online, target):
                prios.append(eng.prio.cpu().numpy().copy())
        torch.cuda.synchronize()
        cbuf = C.create_string_buffer(1 << 16)
        capi.check(capi.lib.rela_prof_counts_json(cbuf, len(cbuf)), "census")
        capi.lib.rela_prof_count_enable(0)
        ran = set(json.loads(cbuf.value.decode()))
        fast = {CONV12, "conv_bf16s<Conv3F>", "fc_bf16s (split-K)"}
        assert (fast <= ran) if (precision == "bf16x2" and R >= 128) else not (fast & ran), (precision, sorted(ran))
        ring = replay.debug_state()["ring"]
        w, ev = np.zeros(ring, np.float32), np.zeros(ring, np.uint8)
        capi.check(capi.lib.rela_replay_debug_weights(replay.h, w.ctypes.data_as(C.c_void_p),
                                                      ev.ctypes.data_as(C.c_void_p)), "debug_weights")
        runs.append((np.array(acts), np.array(prios), w, replay.size()))
        eng.close()
        replay.close()
        online.close()
        target.close()
    for other in runs[1:]:
        assert runs[0][3] == other[3] == 8 * R
        assert np.array_equal(runs[0][0], other[0])
        assert np.array_equal(runs[0][1].view(np.uint32), other[1].view(np.uint32))
        assert np.array_equal(runs[0][2].view(np.uint32), other[2].view(np.uint32))


def test_transition_carries_eps_and_legal_of_its_own_time_step():
    """obs["eps"] / obs["legal_move"] change from step to step: the stored transition must hold the
    values of time t-n on its obs side and of time t on its next_obs side (the reference pushes the
    whole obs dict into the n-step history, dqn_actor.h:23-29,84-90), and the priority must equal
    a plain fp32 PyTorch evaluation of apex.py:30-45 with those per-time masks."""
    import torch
    import torch.nn.functional as F

    from rela_amd.engine import ApexActorEngine, FFNetHandle
    from rela_amd.replay import FFReplay
    from synth import synth_params

    A, R, n, gamma = 6, 8, 2, 0.9
    p_on, p_tg = synth_params(A, 21), synth_params(A, 22)
    online, target = FFNetHandle(A, "cuda:0"), FFNetHandle(A, "cuda:0")
    online.load_state_dict({k: torch.from_numpy(v) for k, v in p_on.items()})
    target.load_state_dict({k: torch.from_numpy(v) for k, v in p_tg.items()})
    replay = FFReplay(64, 3, 1.0, 0.4, 0, A, "cuda:0")
    eng = ApexActorEngine(R, R, A, n, gamma, replay, np.zeros(R, np.float32), "cuda:0", seed=5)
    rng = np.random.default_rng(0)

    def net(p, s, legal):
        t = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
        x = s.float() / 255.0
        x = F.relu(F.conv2d(x, t["net.0.weight"], t["net.0.bias"], stride=4))
        x = F.relu(F.conv2d(x, t["net.2.weight"], t["net.2.bias"], stride=2))
        x = F.relu(F.conv2d(x, t["net.4.weight"], t["net.4.bias"], stride=1))
        h = F.relu(F.linear(x.reshape(s.shape[0], 3136), t["linear.0.weight"], t["linear.0.bias"]))
        v, a = F.linear(h, t["fc_v.weight"], t["fc_v.bias"]), F.linear(h, t["fc_a.weight"], t["fc_a.bias"]) * legal
        return v + a - a.mean(1, keepdim=True)

    T = n + 3
    obs, legal, eps, acts, rews = [], [], [], [], []
    popped = []
    for t in range(T):
        o = torch.from_numpy(rng.integers(0, 256, (R, 4, 84, 84), dtype=np.uint8)).cuda()
        lg = (rng.uniform(size=(R, A)) < 0.7).astype(np.float32)
        lg[:, t % A] = 1.0  # at least one legal move
        lg = torch.from_numpy(lg).cuda()
        e = torch.full((R, 1), 0.0, device="cuda") + 0.001 * t  # eps tiny: value recorded, acts stay greedy mostly
        obs.append(o), legal.append(lg), eps.append(e)
        eng.next_obs_slot().copy_(o)
        eng.legal.copy_(lg)
        eng.eps.copy_(e)
        acts.append(eng.act(online).clone())
        r = torch.from_numpy(rng.uniform(-1, 1, R).astype(np.float32)).cuda()
        rews.append(r)
        if eng.post_step(r, torch.zeros(R, dtype=torch.uint8, device="cuda"), online, target):
            popped.append((t - n, eng.prio.clone()))
    torch.cuda.synchronize()
    assert [t0 for t0, _ in popped] == list(range(T - n))
    with torch.no_grad():
        for t0, prio in popped:
            t1 = t0 + n
            ret = sum((gamma ** i) * rews[t0 + i] for i in range(n))
            q_on = net(p_on, obs[t0], legal[t0]).gather(1, acts[t0].unsqueeze(1)).squeeze(1)
            qn = net(p_on, obs[t1], legal[t1])
            next_a = ((1 + qn - qn.min()) * legal[t1]).argmax(1)
            boot = net(p_tg, obs[t1], legal[t1]).gather(1, next_a.unsqueeze(1)).squeeze(1)
            ref = (ret + (gamma ** n) * boot - q_on).abs()
            np.testing.assert_allclose(prio.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-4)
    # stored rows: alpha = 1 so the ring weight is the priority; sample everything a few times
    seen = set()
    for _ in range(40):
        batch, w = replay.sample(8)
        s0 = batch.obs["s"].cpu().numpy()
        for i in range(8):
            hits = [t0 for t0 in range(T - n) for r in range(R) if np.array_equal(s0[i], obs[t0][r].cpu().numpy())]
            assert hits, "sampled row is not one of the inserted observations"
            t0 = hits[0]
            r = [r for r in range(R) if np.array_equal(s0[i], obs[t0][r].cpu().numpy())][0]
            assert np.array_equal(batch.obs["legal_move"][i].cpu().numpy(), legal[t0][r].cpu().numpy())
            assert np.array_equal(batch.next_obs["legal_move"][i].cpu().numpy(), legal[t0 + n][r].cpu().numpy())
            assert np.array_equal(batch.obs["eps"][i].cpu().numpy(), eps[t0][r].cpu().numpy())
            assert np.array_equal(batch.next_obs["eps"][i].cpu().numpy(), eps[t0 + n][r].cpu().numpy())
            assert np.array_equal(batch.next_obs["s"][i].cpu().numpy(), obs[t0 + n][r].cpu().numpy())
            seen.add((t0, r))
        replay.update_priority(w.clone())
    assert len(seen) > R
    eng.close()
    replay.close()
    online.close()
    target.close()
