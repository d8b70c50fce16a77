"""GPU parity of the hand-written Ape-X learner step (csrc/learner.hip, through the C ABI) against
PyTorch autograd on the same batch: loss, priorities, every gradient tensor, and the parameters
after clip_grad_norm_ + RMSprop / Adam steps (pyrela/main.py:226-239, pyrela/apex.py:30-91)."""
import glob
import json
import os

import numpy as np
import pytest

from kernel_names import CONV12, CONV12_JOBS  # noqa: F401

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

RTOL, ATOL = 2e-3, 2e-5  # fp32 sums of up to 2e5 terms in different orders (split-K vs MIOpen)


def make_batch(B, A, seed, device="cuda", adjacent=False):
    """adjacent: next_obs right behind obs in ONE tensor (and the legal moves likewise), as rela_amd.replay.FFReplay hands a
    sampled batch over -- the f32x3 learner then runs the online net over [s ; s'] as one forward"""
    import torch
    from types import SimpleNamespace

    g = torch.Generator(device="cpu").manual_seed(seed)
    s = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, generator=g)
    ns = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, generator=g)
    legal = (torch.rand(B, A, generator=g) < 0.8).float()
    legal[:, 0] = 1.0
    nlegal = (torch.rand(B, A, generator=g) < 0.8).float()
    nlegal[:, 1 % A] = 1.0
    a = torch.multinomial(legal, 1, generator=g).squeeze(1)
    # rewards large enough that some TD errors leave the quadratic zone of the Huber loss
    reward = torch.randn(B, generator=g) * 0.7
    boot = (torch.rand(B, generator=g) < 0.9).float()
    w = torch.rand(B, generator=g) * 0.9 + 0.1
    to = lambda x: x.to(device)
    if adjacent:
        frames, moves = to(torch.stack([s, ns])), to(torch.stack([legal, nlegal]))
        s_d, ns_d, legal_d, nlegal_d = frames[0], frames[1], moves[0], moves[1]
    else:
        s_d, ns_d, legal_d, nlegal_d = to(s), to(ns), to(legal), to(nlegal)
    batch = SimpleNamespace(obs={"s": s_d, "eps": to(torch.zeros(B, 1)), "legal_move": legal_d},
                            next_obs={"s": ns_d, "eps": to(torch.zeros(B, 1)), "legal_move": nlegal_d},
                            action={"a": to(a)}, reward=to(reward), terminal=to(boot < 0.5), bootstrap=to(boot))
    return batch, to(w)


def make_agent(A, seed, multi_step=3, gamma=0.99):
    import torch

    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet

    torch.manual_seed(seed)
    agent = ApexAgent(lambda: AtariFFNet(A), multi_step, gamma).to("cuda")
    with torch.no_grad():  # target != online, biases non-zero
        for p in agent.target_net.parameters():
            p.add_(torch.randn_like(p) * 0.01)
        for p in agent.online_net.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
    return agent


@pytest.mark.parametrize("precision", ["f32", "f32x3", "f32x3_adjacent"])
@pytest.mark.parametrize("B,A", [(32, 6), (512, 18), (100, 18)])
def test_loss_priority_and_gradients_match_autograd(B, A, precision):
    """(f32x3: from 512 rows conv2 / conv3 / fc of the three forwards run on the f32-accurate three-part bf16 kernels,
    csrc/gemm_f32emu.h -- asserted through the launch census; same tolerances as the exact f32 mode.)"""
    import torch

    from rela_amd.learner import HipApexLearner

    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    agent = make_agent(A, 3)
    adjacent = precision.endswith("_adjacent")
    precision = precision.split("_")[0]
    batch, w = make_batch(B, A, 11, adjacent=adjacent)
    learner = HipApexLearner.from_agent(agent, B)
    learner.set_precision(precision)
    from rela_amd import _capi as capi

    with capi.launch_census() as census:
        loss, prio = learner.backward(batch, w)
    emu = {"conv12_s3", "conv3_img_s3", "gemm_s3<fc>"}
    if precision == "f32x3" and B >= 512:
        # separate buffers: three launches per layer; s' right behind s (r5): the online net over [s ; s'] as ONE 1,024-row
        # launch per layer + the target net over s'.  (Separately allocated tensors may happen to sit back to back too: the
        # library looks at the addresses, so does the expectation.)
        so, sn = batch.obs["s"], batch.next_obs["s"]
        lo, ln = batch.obs["legal_move"], batch.next_obs["legal_move"]
        behind = sn.data_ptr() == so.data_ptr() + so.numel() and ln.data_ptr() == lo.data_ptr() + 4 * lo.numel()
        assert behind or not adjacent
        assert emu <= set(census.counts) and census.counts["conv12_s3"] == (2 if behind else 3), census.counts
    else:
        assert not (emu & set(census.counts)), census.counts
    per_sample, ref_prio = agent.loss(batch, sync_priority=False)
    ref_loss = (per_sample * w).mean()
    ref_loss.backward()
    np.testing.assert_allclose(prio.cpu().numpy(), ref_prio.cpu().numpy(), rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(loss.item(), ref_loss.item(), rtol=1e-4, atol=1e-5)
    grads = learner.state_dict("grads")
    ref = dict(agent.online_net.named_parameters())
    for key in HipApexLearner.KEYS:
        gr, rr = grads[key].cpu().numpy(), ref[key].grad.cpu().numpy()
        scale = float(np.abs(rr).max()) + 1e-12
        np.testing.assert_allclose(gr, rr, rtol=RTOL, atol=ATOL + 1e-3 * scale, err_msg=key)
    learner.close()


MERGED_MIN_B, MERGED_MAX_B = 128, 1023  # batches the merged split-bf16 forward serves (csrc/ffnet.hip)


def _assert_fast_learner_kernels(counts, B):
    """The kernels a bf16x2 learner step of B rows must launch (csrc/learner.hip, ffnet.hip, learner_common.h): from
    128 rows the three forwards of td_err as ONE split-bf16 launch per layer (conv12_i8_jobs, conv3_bf16s_jobs,
    fc_bf16s split-K) and no f32 trunk kernel at all; at every size conv1's weight gradient and the conv2 / conv3 data
    gradients on bf16 MFMA -- so a silent fall-back to the f32 kernels cannot pass for a test of the fast ones."""
    want = {"wgrad_conv1_bf16", "dgrad_conv2_bf16", "dgrad_conv3_bf16"}
    if MERGED_MIN_B <= B <= MERGED_MAX_B:
        want |= {CONV12_JOBS, "conv3_bf16s_jobs", "fc_bf16s (split-K)", "unsplit_trunk_rows"}
    assert want <= set(counts), "B=%d: launched %s, expected %s" % (B, sorted(counts), sorted(want))
    if MERGED_MIN_B <= B <= MERGED_MAX_B:
        assert not ({"conv1_bf16x3", "conv_mfma<Conv2> (f32)", "conv_mfma<Conv3> (f32)"} & set(counts)), sorted(counts)
        assert counts[CONV12_JOBS] == 1 and counts["fc_bf16s (split-K)"] == 2


def _relu_pattern_disagreement(a, b):
    """fraction of units whose ReLU state differs between two sets of activations [a1, a2, a3, h]"""
    diff = sum(int(((x > 0) != (y > 0)).sum()) for x, y in zip(a, b))
    return diff / float(sum(x.numel() for x in a))


@pytest.mark.parametrize("B", [512, 200, 64])
def test_learner_fast_mode_within_stated_tolerance(B):
    """set_precision("bf16x2") against the all-f32 step on the same batch.  From 128 rows ALL THREE forwards of td_err
    run on split-bf16 MFMA (r3: online over [s ; s'] and target over s', one launch per layer), so the backward pass
    reads the activations and ReLU pattern of a forward whose pre-activations differ from the f32 ones by ~2^-16
    relative: loss and priorities stay within 5e-6, and the ReLU patterns agree on all but a few units in 10^5
    (asserted: < 1e-4) -- units whose pre-activation is within that error of zero, where the gradient of the network is
    discontinuous anyway.  Each flipped unit moves its layer's weight gradient by one term of a random-sign sum, so the
    gradient TENSORS agree to ~5e-3 of their norm (asserted: relative L2 error < 2e-2, cosine > 0.9998), while the
    gradients are exact for the forward they belong to
    (test_fast_mode_gradients_match_autograd_on_the_forwards_own_relu_pattern: 2e-3 of the largest entry, the f32
    mode's own bound).  Below 128 rows only the gradient kernels are bf16 (conv1's weight gradient, conv2 / conv3 data
    gradients) and the gradients stay within 1e-4 of the f32 step's."""
    import torch

    from rela_amd.learner import HipApexLearner

    A = 18
    agent = make_agent(A, 7)
    batch, w = make_batch(B, A, 23)
    learner = HipApexLearner.from_agent(agent, B)
    loss0, prio0 = learner.backward(batch, w)
    loss0, prio0 = loss0.clone(), prio0.clone()
    g0 = {k: v.clone() for k, v in learner.state_dict("grads").items()}
    act0 = [t.clone() for t in learner.debug_activations()]
    learner.set_precision("bf16x2")
    from rela_amd import _capi as capi

    with capi.launch_census() as census:
        loss1, prio1 = learner.backward(batch, w)
    g1 = learner.state_dict("grads")
    _assert_fast_learner_kernels(census.counts, B)
    assert float((prio1 - prio0).abs().max()) < 5e-6
    assert abs(float(loss1) - float(loss0)) < 5e-6 * max(1.0, abs(float(loss0)))
    merged = MERGED_MIN_B <= B <= MERGED_MAX_B
    flips = _relu_pattern_disagreement(act0, learner.debug_activations())
    print("B=%d: ReLU pattern disagreement bf16x2 vs f32 forward: %.3g" % (B, flips))
    assert flips < 1e-4 if merged else flips == 0.0
    for key in HipApexLearner.KEYS:
        if merged:
            d, n0 = float((g1[key] - g0[key]).norm()), float(g0[key].norm()) + 1e-20
            cos = float((g1[key] * g0[key]).sum()) / (float(g1[key].norm()) * n0 + 1e-30)
            assert d <= 2e-2 * n0 and cos > 0.9998, (key, d / n0, cos)
        else:
            scale = float(g0[key].abs().max()) + 1e-12
            assert float((g1[key] - g0[key]).abs().max()) <= 1e-4 * scale, key  # d(Huber) can flip at |err| = 1 only
    learner.set_precision("f32")
    loss2, prio2 = learner.backward(batch, w)
    assert torch.equal(prio2, prio0) and torch.equal(loss2, loss0)  # the parity mode is untouched by the switch
    learner.close()


@pytest.mark.parametrize("B", [512, 200])
def test_fast_mode_gradients_match_autograd_on_the_forwards_own_relu_pattern(B):
    """The bf16x2 step's gradients against PyTorch autograd of the SAME network with the ReLUs replaced by the 0/1
    pattern the HIP forward actually produced (rela_apex_learner_debug_activations): where a pre-activation sits
    within rounding of zero the two forwards may disagree about a unit, and the gradient of the network is
    discontinuous there -- with the pattern fixed the comparison is as tight as the f32 mode's own bound against
    autograd (2e-3 of each tensor's largest entry; RTOL / ATOL above)."""
    import torch
    import torch.nn.functional as F

    from rela_amd.learner import HipApexLearner

    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    A = 18
    agent = make_agent(A, 13)
    batch, w = make_batch(B, A, 31)
    learner = HipApexLearner.from_agent(agent, B)
    learner.set_precision("bf16x2")
    loss, prio = learner.backward(batch, w)
    a1, a2, a3, h = learner.debug_activations()
    m1 = (a1 > 0).float().reshape(B, 20, 20, 32).permute(0, 3, 1, 2)
    m2 = (a2 > 0).float().reshape(B, 9, 9, 64).permute(0, 3, 1, 2)
    m3 = (a3 > 0).float().reshape(B, 7, 7, 64).permute(0, 3, 1, 2)
    mh = (h > 0).float()
    net = agent.online_net
    p = dict(net.named_parameters())

    def q_masked(s, legal):
        x = s.float() / 255.0
        x = F.conv2d(x, p["net.0.weight"], p["net.0.bias"], stride=4) * m1
        x = F.conv2d(x, p["net.2.weight"], p["net.2.bias"], stride=2) * m2
        x = F.conv2d(x, p["net.4.weight"], p["net.4.bias"], stride=1) * m3
        hh = F.linear(x.reshape(B, 3136), p["linear.0.weight"], p["linear.0.bias"]) * mh
        v, a = F.linear(hh, p["fc_v.weight"], p["fc_v.bias"]), F.linear(hh, p["fc_a.weight"], p["fc_a.bias"]) * legal
        return v + a - a.mean(1, keepdim=True)

    with torch.no_grad():  # td target exactly as apex.py:30-45 (the no-grad forwards through the real ReLUs)
        nobs = batch.next_obs
        qn = agent.online_net(nobs)
        na = ((1 + qn - qn.min()) * nobs["legal_move"]).argmax(1)
        qt = agent.target_net(nobs).gather(1, na.unsqueeze(1)).squeeze(1)
        target = batch.reward + batch.bootstrap * (agent.gamma ** agent.multi_step) * qt
    qa = q_masked(batch.obs["s"], batch.obs["legal_move"]).gather(1, batch.action["a"].unsqueeze(1)).squeeze(1)
    err = target - qa
    ref_loss = (F.smooth_l1_loss(err, torch.zeros_like(err), reduction="none") * w).mean()
    ref_loss.backward()
    np.testing.assert_allclose(loss.item(), ref_loss.item(), rtol=1e-4, atol=1e-5)
    grads = learner.state_dict("grads")
    worst = 0.0
    for key in HipApexLearner.KEYS:
        gr, rr = grads[key].cpu().numpy(), p[key].grad.cpu().numpy()
        scale = float(np.abs(rr).max()) + 1e-12
        worst = max(worst, float(np.abs(gr - rr).max()) / scale)
        np.testing.assert_allclose(gr, rr, rtol=RTOL, atol=ATOL + 1e-3 * scale, err_msg=key)
    print("B=%d: worst |dgrad| / max|grad| against autograd on the forward's own ReLU pattern: %.3g" % (B, worst))
    learner.close()


def test_pipelined_step_is_bit_identical_to_the_sequential_one():
    """rela_apex_learner_loss / _grad with the replay in deferred-wait mode: update_priority and the NEXT sample are
    queued between the two halves (the sample path then runs next to the gradient kernels; batches alternate between
    two buffer slots).  Sampled ids, IS weights, priorities, losses and the parameters after six steps must equal the
    strictly sequential sample -> backward -> apply -> update_priority loop bit for bit."""
    import ctypes as C

    import torch

    from rela_amd import _capi as capi
    from rela_amd.learner import HipApexLearner
    from rela_amd.replay import FFReplay

    A, B, cap, rows = 6, 64, 512, 640
    data, _ = make_batch(rows, A, 5)
    prio = torch.rand(rows, device="cuda") * 3 + 0.05
    out = []
    for pipelined in (False, True):
        agent = make_agent(A, 9)
        learner = HipApexLearner.from_agent(agent, B)
        rep = FFReplay(cap, 17, 0.6, 0.4, 0, A, "cuda:0")
        ptrs = [data.obs["s"], data.next_obs["s"], data.obs["eps"], data.next_obs["eps"], data.obs["legal_move"],
                data.next_obs["legal_move"], data.action["a"], data.reward, data.terminal.to(torch.uint8),
                data.bootstrap]
        for lo in range(0, rows, 128):  # more rows than capacity: the first sample evicts
            rep.add_rows(128, [t[lo:lo + 128].contiguous().data_ptr() for t in ptrs], prio[lo:lo + 128].contiguous())
        trace = []
        if not pipelined:
            for k in range(6):
                batch, w = rep.sample(B)
                loss, p = learner.step(batch, w)
                rep.update_priority(p)
                trace.append((batch.action["a"].clone(), batch.reward.clone(), w.clone(), p.clone(), loss.clone()))
        else:
            rep.set_deferred_wait(True)
            nxt = rep.sample(B, slot=0)
            for k in range(6):
                batch, w = nxt
                rep.wait()
                loss, p = learner.loss(batch, w)
                rep.update_priority(p)
                trace.append((batch.action["a"].clone(), batch.reward.clone(), w.clone(), p.clone(), loss.clone()))
                if k < 5:
                    nxt = rep.sample(B, slot=(k + 1) & 1)
                learner.grad()
                learner.apply()
            rep.wait()
        torch.cuda.synchronize()
        assert rep.debug_state()["dev_error"] == 0
        out.append((trace, learner.flat()[0].clone()))
        learner.close()
        rep.close()
    for a, b in zip(out[0][0], out[1][0]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("opt", ["rmsprop", "adam"])
def test_clip_and_optimizer_match_torch_on_identical_gradients(opt):
    """clip_grad_norm_ + optimiser arithmetic in isolation: autograd's gradients are copied into the
    learner's gradient buffer, so both sides update from the same numbers."""
    import torch

    from rela_amd.learner import HipApexLearner

    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    B, A = 64, 6
    agent = make_agent(A, 5)
    lr, eps, clip = 1e-3, 1.5e-4, 0.05  # clip small enough to engage
    learner = HipApexLearner.from_agent(agent, B, optimizer=opt, lr=lr, eps=eps, grad_clip=clip)
    named = dict(agent.online_net.named_parameters())
    params = list(named.values())
    optim = (torch.optim.RMSprop if opt == "rmsprop" else torch.optim.Adam)(params, lr=lr, eps=eps)
    gviews = learner.state_dict("grads")
    for step in range(4):
        batch, w = make_batch(B, A, 100 + step)
        per_sample, _ = agent.loss(batch, sync_priority=False)
        (per_sample * w).mean().backward()
        for key in HipApexLearner.KEYS:
            gviews[key].copy_(named[key].grad)
        gnorm = torch.nn.utils.clip_grad_norm_(params, clip)
        optim.step()
        optim.zero_grad()
        learner.apply()
        st = learner.stats().cpu().numpy()
        np.testing.assert_allclose(st[0], gnorm.item(), rtol=1e-5)
        np.testing.assert_allclose(st[1], min(1.0, clip / (gnorm.item() + 1e-6)), rtol=1e-5)
        sd = learner.state_dict("online")
        for key in HipApexLearner.KEYS:
            np.testing.assert_allclose(sd[key].cpu().numpy(), named[key].detach().cpu().numpy(), rtol=0, atol=2e-6,
                                       err_msg="%s step %d" % (key, step))
    learner.close()


def test_learner_steps_track_the_torch_learner():
    """Three full steps (own backward + clip + RMSprop) next to the PyTorch learner.  RMSprop divides
    by sqrt(E[g^2]): weights whose gradient is ~0 move by +-lr regardless of its size, so rounding
    noise in such a gradient flips a whole step; the comparison allows a fraction of the 3 * lr / (1 - alpha)**0.5
    = 3e-2 the three steps can move a weight, and checks that the bulk agrees far tighter."""
    import torch

    from rela_amd.engine import FFNetHandle
    from rela_amd.learner import HipApexLearner

    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    B, A = 64, 6
    agent = make_agent(A, 5)
    lr, eps, clip = 1e-3, 1.5e-4, 40.0
    learner = HipApexLearner.from_agent(agent, B, lr=lr, eps=eps, grad_clip=clip)
    params = list(agent.online_net.parameters())
    optim = torch.optim.RMSprop(params, lr=lr, eps=eps)
    for step in range(3):
        batch, w = make_batch(B, A, 200 + step)
        _, prio = learner.step(batch, w)
        per_sample, ref_prio = agent.loss(batch, sync_priority=False)
        (per_sample * w).mean().backward()
        torch.nn.utils.clip_grad_norm_(params, clip)
        optim.step()
        optim.zero_grad()
        np.testing.assert_allclose(prio.cpu().numpy(), ref_prio.cpu().numpy(), rtol=5e-3, atol=5e-3)
    sd = learner.state_dict("online")
    ref = agent.online_net.state_dict()
    for key in HipApexLearner.KEYS:
        d = np.abs(sd[key].cpu().numpy() - ref[key].cpu().numpy())
        assert d.max() < 6e-3, (key, d.max())
        assert np.median(d) < 2e-5, (key, np.median(d))
    # sync_target + publish round trip
    learner.sync_target_with_online()
    tsd = learner.state_dict("target")
    for key in HipApexLearner.KEYS:
        assert torch.equal(tsd[key], sd[key])
    net = FFNetHandle(A, "cuda:0")
    learner.publish(net)
    torch.cuda.synchronize()
    net.close()
    learner.close()


def test_learner_errors():
    import ctypes as C

    from rela_amd import _capi as capi

    h = C.c_void_p()
    assert capi.lib.rela_apex_learner_create(C.byref(h), 40, 32, 3, 0.99, 0, 1e-4, 1e-4, 40.0, 0) == capi.EINVAL
    assert capi.lib.rela_apex_learner_create(C.byref(h), 6, 32, 3, 0.99, 0, 1e-4, 1e-4, 40.0, 0) == capi.OK
    assert capi.lib.rela_apex_learner_apply(h, None) == capi.ESTATE  # never loaded
    assert capi.lib.rela_apex_learner_grad(h, None) == capi.ESTATE  # never loaded, and no loss() to differentiate
    capi.lib.rela_apex_learner_destroy(h)
    # grad() needs a loss() before it, once per loss(); a batch larger than the learner's is refused by its nets
    from rela_amd.learner import HipApexLearner

    learner = HipApexLearner.from_agent(make_agent(6, 3), 32)
    with pytest.raises(RuntimeError):
        learner.grad()
    batch, w = make_batch(32, 6, 4)
    learner.loss(batch, w)
    learner.grad()
    with pytest.raises(RuntimeError):
        learner.grad()
    big, wb = make_batch(48, 6, 4)
    with pytest.raises(RuntimeError):
        learner.loss(big, wb)
    learner.close()


@pytest.mark.parametrize("precision", ["f32", "f32x3", "bf16x2"])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "learner_*.json"))), ids=os.path.basename)
def test_learner_step_matches_the_reference_golden(path, precision):
    """One learner step against vectors recorded from the REAL reference on CPU
    (tests/golden/make_golden.py learner_cases: pyrela/apex.py loss -> backward -> clip_grad_norm_ ->
    RMSprop.step of pyrela/main.py:226-239): loss, priorities, gradient norm, and for every parameter
    tensor its gradient / updated value (l2 norm, sum, 48 sampled entries) -- in BOTH precision modes, with the
    launched kernels asserted (learner_apex_A18_B128 reaches the split-bf16 forward trunk; the bf16 gradient
    kernels run at every batch size)."""
    import torch
    from types import SimpleNamespace

    from rela_amd.learner import HipApexLearner
    from synth import synth_obs, synth_params

    g = json.load(open(path))
    A, B = g["num_action"], g["B"]
    dev = "cuda:0"
    learner = HipApexLearner(A, B, g["multi_step"], g["gamma"], lr=g["lr"], eps=g["eps"], grad_clip=g["grad_clip"],
                             device=dev)
    to = lambda sd: {k: torch.from_numpy(v).to(dev) for k, v in sd.items()}
    learner.load_state_dicts(to(synth_params(A, g["online_seed"])), to(synth_params(A, g["target_seed"])))
    f32 = lambda x: torch.tensor(x, dtype=torch.float32, device=dev)
    batch = SimpleNamespace(
        obs={"s": torch.from_numpy(synth_obs(B, g["obs_seed"])).to(dev), "eps": torch.zeros(B, 1, device=dev),
             "legal_move": f32(g["legal"])},
        next_obs={"s": torch.from_numpy(synth_obs(B, g["next_obs_seed"])).to(dev),
                  "eps": torch.zeros(B, 1, device=dev), "legal_move": f32(g["next_legal"])},
        action={"a": torch.tensor(g["action"], dtype=torch.int64, device=dev)}, reward=f32(g["reward"]),
        terminal=torch.zeros(B, dtype=torch.bool, device=dev), bootstrap=f32(g["bootstrap"]))
    from rela_amd import _capi as capi

    learner.set_precision(precision)
    with capi.launch_census() as census:
        loss, prio = learner.backward(batch, f32(g["weight"]))
    if precision == "bf16x2":
        _assert_fast_learner_kernels(census.counts, B)
    else:
        assert not ({"wgrad_conv1_bf16", "dgrad_conv2_bf16", "conv12_i8"} & set(census.counts)), census.counts
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(prio.cpu().numpy(), np.array(g["priority"]), rtol=1e-4, atol=1e-4)

    def check(views, gold, tol_scale, what):
        for key, rec in gold.items():
            t = views[key].double().reshape(-1).cpu()
            scale = rec["absmax"] + 1e-12
            np.testing.assert_allclose(float(t.norm()), rec["l2"], rtol=2e-3, atol=tol_scale * scale,
                                       err_msg="%s l2 %s" % (what, key))
            np.testing.assert_allclose(t[torch.tensor(rec["idx"])].numpy(), np.array(rec["val"]), rtol=2e-3,
                                       atol=tol_scale * scale, err_msg="%s %s" % (what, key))

    # (bf16x2 from 128 rows: the backward differentiates the split-bf16 forward, whose ReLU pattern differs from the
    # reference's f32 forward in a few units per 10^5 -- see test_learner_fast_mode_within_stated_tolerance)
    merged = precision == "bf16x2" and MERGED_MIN_B <= B <= MERGED_MAX_B
    check(learner.state_dict("grads"), g["grads"], 1e-2 if merged else 2e-3, "grad")
    learner.apply()
    st = learner.stats().cpu().numpy()
    np.testing.assert_allclose(st[0], g["grad_norm"], rtol=1e-3)
    # one RMSprop step moves a weight by at most lr / sqrt(1 - alpha) = 10 lr
    check(learner.state_dict("online"), g["params_after"], 0.02 * 10 * g["lr"] / 1.0, "param")
    learner.close()


def test_actor_net_loaded_from_flat_buffer_matches_publish():
    """load_net_from_flat (what an actor-only rank does after broadcast_weights) gives the same
    forward as HipApexLearner.publish, and the Python layout equals the library's."""
    import torch

    from rela_amd import _capi as capi
    from rela_amd.engine import FFNetHandle
    from rela_amd.learner import HipApexLearner, ffnet_flat_layout, load_net_from_flat
    from synth import synth_obs, synth_params
    import ctypes as C

    A, N = 18, 40
    learner = HipApexLearner(A, 32, 3, 0.99)
    learner.load_state_dicts({k: torch.from_numpy(v).cuda() for k, v in synth_params(A, 9).items()})
    flat_p, _ = learner.flat()
    layout, total = ffnet_flat_layout(A)
    assert flat_p.numel() == total
    sd = learner.state_dict("online")
    for key, shape, off in layout:
        assert sd[key].data_ptr() == flat_p.data_ptr() + 4 * off and tuple(sd[key].shape) == tuple(shape)
    received = flat_p.clone()  # stands for the broadcast destination on another rank
    a, b = FFNetHandle(A, "cuda:0"), FFNetHandle(A, "cuda:0")
    learner.publish(a)
    load_net_from_flat(b, received, A)
    s = torch.from_numpy(synth_obs(N, 3)).cuda()
    legal = torch.ones(N, A, device="cuda")
    nb = capi.lib.rela_ffnet_workspace_bytes(a.h, N)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    qs = []
    for net in (a, b):
        q = torch.empty(N, A, device="cuda")
        capi.check(capi.lib.rela_ffnet_forward(net.h, N, C.c_void_p(s.data_ptr()), C.c_void_p(legal.data_ptr()),
                                               C.c_void_p(q.data_ptr()), C.c_void_p(ws.data_ptr()), nb,
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fwd")
        qs.append(q)
    torch.cuda.synchronize()
    assert torch.equal(qs[0], qs[1])
    a.close()
    b.close()
    learner.close()


def test_f32_fc_layout_is_packed_on_demand_after_a_fast_mode_step():
    """In the split-bf16 mode no kernel reads the f32 split-K fc's weight layout (BfT) from 128 rows up, so a learner
    whose batch can reach 128 rows skips it in the re-pack after every step and the f32 path packs it when it is next
    needed (csrc/ffnet.hip: rela_ffnet::bft_stale).  After a fast-mode step at B = 256, a 64-row batch -- whose forwards
    run the f32 kernels, BfT included -- must give exactly what a learner with the SAME parameters and a 64-row maximum
    (which packs BfT eagerly) gives, and so must the f32 mode at 256 rows."""
    import torch

    from rela_amd.learner import HipApexLearner

    A = 18
    agent = make_agent(A, 11)
    big = HipApexLearner.from_agent(agent, 256)
    big.set_precision("bf16x2")
    b256, w256 = make_batch(256, A, 31)
    big.step(b256, w256)  # backward + optimiser + re-pack in the fast mode: BfT is now stale
    on = {k: v.clone() for k, v in big.state_dict("online").items()}
    tg = {k: v.clone() for k, v in big.state_dict("target").items()}
    small = HipApexLearner(A, 64, agent.multi_step, agent.gamma)
    small.load_state_dicts(on, tg)
    small.set_precision("bf16x2")
    b64, w64 = make_batch(64, A, 32)
    loss_a, prio_a = big.backward(b64, w64)
    loss_b, prio_b = small.backward(b64, w64)
    assert torch.equal(prio_a, prio_b) and torch.equal(loss_a, loss_b)
    ref = HipApexLearner(A, 256, agent.multi_step, agent.gamma)  # f32 from the start: every layout packed at load
    ref.load_state_dicts(on, tg)
    big.step(b256, w256)  # stale again ...
    on2 = {k: v.clone() for k, v in big.state_dict("online").items()}
    ref.load_state_dicts(on2, tg)
    big.set_precision("f32")  # ... and now the f32 mode needs it at 256 rows
    loss_c, prio_c = big.backward(b256, w256)
    loss_d, prio_d = ref.backward(b256, w256)
    assert torch.equal(prio_c, prio_d) and torch.equal(loss_c, loss_d)
    for l in (big, small, ref):
        l.close()
