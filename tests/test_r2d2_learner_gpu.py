"""R2D2 learner step on the device.

  1. rela_amd/pyrela/r2d2.py on PyTorch-ROCm autograd ON THE GPU against vectors recorded from the REAL
     reference's R2D2Agent on CPU (tests/golden/r2d2_loss_A6_B3.json: loss per sequence, aggregated priority,
     gradients of (loss * weight).mean() w.r.t. every online parameter).
  2. the hand-written HIP step (csrc/learner_r2d2.hip via rela_amd.learner.HipR2D2Learner) against the same
     reference vectors: loss / priority 1e-4, every gradient tensor 2e-3.
  3. the HIP step against PyTorch autograd on the device at BASELINE config C4's learner shape
     (B = 64, T = 40 + 80 + 3 = 123, A = 18): every gradient tensor 2e-3 relative to its largest entry.
  4. clip_grad_norm_ + Adam arithmetic and a 2-step trajectory against torch.optim.Adam.
"""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

from kernel_names import CONV12, CONV12_JOBS  # noqa: F401

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _agent(A, n, gamma, eta, seq, burn, on_seed, tg_seed, device):
    import torch

    from rela_amd.pyrela.net import AtariLSTMNet
    from rela_amd.pyrela.r2d2 import R2D2Agent
    from synth import synth_lstm_params

    agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, A), "cpu", n, gamma, eta, seq, burn, 0)
    sd = {}
    for prefix, seed in (("online_net.", on_seed), ("target_net.", tg_seed)):
        for k, v in synth_lstm_params(A, seed).items():
            sd[prefix + k] = torch.from_numpy(v)
    agent.load_state_dict(sd)
    return agent.to(device)


def _golden_batch(g, device):
    import torch

    from oracle_lib import h2f
    from synth import synth_obs

    B, seq, burn, n = g["B"], g["seq_len"], g["burn_in"], g["multi_step"]
    T = burn + seq + n
    m = g["batch"]
    f32 = lambda x: torch.tensor(x, dtype=torch.float32, device=device)
    hid = lambda key: torch.tensor([h2f(v) for v in m[key]], dtype=torch.float32, device=device).reshape(1, B, 512)
    batch = SimpleNamespace(
        obs={"s": torch.from_numpy(synth_obs(T * B, m["obs_seed"]).reshape(T, B, 4, 84, 84)).to(device),
             "legal_move": f32(m["legal"]), "eps": torch.zeros(T, B, 1, device=device)},
        h0={"h0": hid("h0"), "c0": hid("c0")}, action={"a": torch.tensor(m["action"], dtype=torch.int64, device=device)},
        reward=f32(m["reward"]), terminal=f32(m["terminal"]).bool(), bootstrap=f32(m["bootstrap"]),
        seq_len=f32(m["seq_len"]))
    return batch, f32(m["weight"])


def _check_grads_vs_golden(named_grads, g, tol=2e-3):
    import torch

    assert set(named_grads) == set(g["grads"])
    for key, rec in g["grads"].items():
        t = named_grads[key].detach().double().reshape(-1).cpu()
        scale = rec["absmax"] + 1e-12
        np.testing.assert_allclose(float(t.norm()), rec["l2"], rtol=tol, atol=1e-4 * scale, err_msg=key)
        np.testing.assert_allclose(t[torch.tensor(rec["idx"])].numpy(), np.array(rec["val"]), rtol=tol,
                                   atol=tol * scale, err_msg=key)


def test_pytorch_r2d2_loss_on_gpu_matches_reference_golden():
    import torch

    g = json.load(open(os.path.join(GOLD, "r2d2_loss_A6_B3.json")))
    torch.backends.cudnn.allow_tf32 = False
    agent = _agent(g["num_action"], g["multi_step"], g["gamma"], g["eta"], g["seq_len"], g["burn_in"], g["online_seed"],
                   g["target_seed"], "cuda:0")
    batch, weight = _golden_batch(g, "cuda:0")
    loss, prio = agent.loss(batch)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), np.array(g["loss"]), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(prio.cpu().numpy(), np.array(g["priority"]), rtol=1e-4, atol=1e-5)
    (loss * weight).mean().backward()
    _check_grads_vs_golden({k: v.grad for k, v in agent.online_net.named_parameters()}, g)


def _synth_batch(g, device):
    """the batch of a golden whose inputs are re-derived from tests/synth.py (synth_r2d2_batch)"""
    import torch

    from synth import synth_r2d2_batch

    B, seq, burn, n = g["B"], g["seq_len"], g["burn_in"], g["multi_step"]
    d = synth_r2d2_batch(g["batch_seed"], g["num_action"], B, seq, burn, n)
    assert d["seq_len"].tolist() == g["seq_lens"]
    T = burn + seq + n
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    batch = SimpleNamespace(
        obs={"s": tt(d["s"]), "legal_move": tt(d["legal"]), "eps": torch.zeros(T, B, 1, device=device)},
        h0={"h0": tt(d["h0"]), "c0": tt(d["c0"])}, action={"a": tt(d["action"])}, reward=tt(d["reward"]),
        terminal=tt(d["terminal"]).bool(), bootstrap=tt(d["bootstrap"]), seq_len=tt(d["seq_len"]))
    return batch, tt(d["weight"]), d["weight"]


# kernels a bf16x2 R2D2 learner step must launch once T * B >= 128 frames (csrc/learner_r2d2.hip)
R2D2_FAST_KERNELS = {CONV12, "conv_bf16s<Conv3F>", "gemm_rec64_nt", "wgrad_conv1_bf16", "dgrad_conv2_bf16",
                     "dgrad_conv3_bf16",
                     # r3: the ONLINE trunk too (conv1's records kept for the backward, then turned back into f32)
                     CONV12_JOBS, "conv3_bf16s_jobs", "unsplit_trunk_rows"}


@pytest.mark.parametrize("precision", ["f32", "f32x3", "bf16x2"])
def test_hip_r2d2_learner_matches_reference_golden_c4_shape(precision):
    """BASELINE config C4's sequence shape (seq 80 / burn-in 40 / n 3: T = 123) with B = 16, A = 18 -- 1,968 frames,
    1,328 training rows: loss per sequence, aggregated priority and every gradient tensor of the hand-written step
    against vectors recorded from the REAL reference's R2D2Agent.loss + backward on CPU
    (tests/golden/r2d2_loss_A18_B16_T123.json), in BOTH precision modes.  In bf16x2 mode the launch census must show
    the split-bf16 kernels (target trunk, rec64 GEMMs of the LSTM's input side, bf16 conv gradient kernels): this is
    the direct pin of those kernels to the reference."""
    import torch

    from rela_amd import _capi as capi
    from rela_amd.learner import HipR2D2Learner

    g = json.load(open(os.path.join(GOLD, "r2d2_loss_A18_B16_T123.json")))
    agent = _agent(g["num_action"], g["multi_step"], g["gamma"], g["eta"], g["seq_len"], g["burn_in"], g["online_seed"],
                   g["target_seed"], "cuda:0")
    learner = HipR2D2Learner.from_agent(agent, g["B"], grad_clip=1e9)
    learner.set_precision(precision)
    batch, weight, w_np = _synth_batch(g, "cuda:0")
    with capi.launch_census() as census:
        loss, prio, loss_seq = learner.backward(batch, weight)
    learner.check()
    torch.cuda.synchronize()
    if precision == "bf16x2":
        assert R2D2_FAST_KERNELS <= set(census.counts), sorted(census.counts)
    else:
        assert not (R2D2_FAST_KERNELS & set(census.counts)), sorted(census.counts)
    # (f32x3: both trunks, 1,968 frames each, and the x part of both gate GEMMs on the three-part kernels)
    emu = {"conv12_s3", "conv3_img_s3", "gemm_s3<gates_x>"}
    assert (emu <= set(census.counts)) if precision == "f32x3" else not (emu & set(census.counts)), sorted(census.counts)
    np.testing.assert_allclose(loss_seq.cpu().numpy(), np.array(g["loss"]), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(prio.cpu().numpy(), np.array(g["priority"]), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(float(loss.cpu()[0]), float((np.array(g["loss"]) * w_np).mean()), rtol=2e-4)
    # bf16x2: the backward differentiates the split-bf16 forward, whose ReLU pattern differs from the reference's f32
    # forward in a few units per 10^5 (tests/test_learner_gpu.py::test_learner_fast_mode_within_stated_tolerance and
    # ..._on_the_forwards_own_relu_pattern pin that mechanism on the shared trunk kernels)
    _check_grads_vs_golden(learner.state_dict("grads"), g, 1e-2 if precision == "bf16x2" else 2e-3)
    learner.close()


def test_hip_r2d2_learner_matches_reference_golden():
    """Loss per sequence, aggregated priority and every gradient tensor of the hand-written step against the
    REAL reference (padded short sequence, dummy burn-in with zeroed state, illegal actions)."""
    import torch

    from rela_amd.learner import HipR2D2Learner

    g = json.load(open(os.path.join(GOLD, "r2d2_loss_A6_B3.json")))
    agent = _agent(g["num_action"], g["multi_step"], g["gamma"], g["eta"], g["seq_len"], g["burn_in"], g["online_seed"],
                   g["target_seed"], "cuda:0")
    learner = HipR2D2Learner.from_agent(agent, g["B"], grad_clip=1e9)
    batch, weight = _golden_batch(g, "cuda:0")
    loss, prio, loss_seq = learner.backward(batch, weight)
    learner.check()  # no grid-barrier timeout in the persistent recurrent kernels
    torch.cuda.synchronize()
    np.testing.assert_allclose(loss_seq.cpu().numpy(), np.array(g["loss"]), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(prio.cpu().numpy(), np.array(g["priority"]), rtol=1e-4, atol=1e-5)
    exp_total = float((np.array(g["loss"]) * np.array(g["batch"]["weight"])).mean())
    np.testing.assert_allclose(float(loss.cpu()[0]), exp_total, rtol=1e-4)
    _check_grads_vs_golden(learner.state_dict("grads"), g)
    learner.close()


def _random_batch(rng, A, B, seq, burn, n, device):
    """RNNTransition-shaped batch with consistent padding (as tests/golden/make_golden.py:_r2d2_batch)."""
    import torch

    T = burn + seq + n
    s = torch.randint(0, 256, (T, B, 4, 84, 84), dtype=torch.uint8, device=device)
    legal = (rng.uniform(size=(T, B, A)) < 0.85).astype(np.float32)
    legal[:, :, 0] = 1.0
    lens = rng.integers(burn + 2, burn + seq + 1, B).astype(np.float32)
    lens[0] = burn + seq
    term = np.zeros((T, B), np.float32)
    for b in range(B):
        if lens[b] < burn + seq:
            term[int(lens[b]) - 1:, b] = 1.0
    start = rng.uniform(size=B) < 0.25  # sequences that start an episode: padLike'd burn-in (terminal = 1)
    term[:burn, start] = 1.0
    boot = (1.0 - np.maximum.reduce([np.roll(term, -k, 0) for k in range(n)])).astype(np.float32)
    boot[T - n:] = 0.0
    action = np.zeros((T, B), np.int64)
    for t in range(T):
        for b in range(B):
            action[t, b] = rng.choice(np.flatnonzero(legal[t, b]))
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    batch = SimpleNamespace(
        obs={"s": s, "legal_move": tt(legal), "eps": torch.zeros(T, B, 1, device=device)},
        h0={"h0": tt(rng.normal(0, 0.3, (1, B, 512)).astype(np.float32)),
            "c0": tt(rng.normal(0, 0.3, (1, B, 512)).astype(np.float32))},
        action={"a": tt(action)}, reward=tt(rng.normal(0, 1.2, (T, B)).astype(np.float32)), terminal=tt(term).bool(),
        bootstrap=tt(boot), seq_len=tt(lens))
    return batch, tt(rng.uniform(0.2, 1.0, B).astype(np.float32))


@pytest.mark.parametrize("A,B,seq,burn,n", [(18, 64, 80, 40, 3), (6, 5, 7, 0, 2), (18, 33, 12, 6, 3)])
def test_hip_r2d2_learner_matches_autograd(A, B, seq, burn, n):
    """C4's learner shape (B = 64, seq 80 / burn-in 40 / n 3 -> 7,872 frames, 83 BPTT steps), a batch without
    burn-in and a ragged batch: loss, priorities and every gradient tensor against PyTorch autograd of
    rela_amd/pyrela/r2d2.py on the same device."""
    import torch

    from rela_amd.learner import HipR2D2Learner

    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    rng = np.random.default_rng(A * 1000 + B)
    agent = _agent(A, n, 0.997, 0.9, seq, burn, 71, 72, "cuda:0")
    batch, weight = _random_batch(rng, A, B, seq, burn, n, "cuda:0")
    learner = HipR2D2Learner.from_agent(agent, B, grad_clip=1e9)
    loss, prio, loss_seq = learner.backward(batch, weight)
    learner.check()  # no grid-barrier timeout in the persistent recurrent kernels
    torch.cuda.synchronize()
    ref_loss, ref_prio = agent.loss(batch, sync_priority=False)
    (ref_loss * weight).mean().backward()
    np.testing.assert_allclose(loss_seq.cpu().numpy(), ref_loss.detach().cpu().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(prio.cpu().numpy(), ref_prio.cpu().numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(float(loss.cpu()[0]), float((ref_loss * weight).mean().detach()), rtol=2e-4)
    grads = learner.state_dict("grads")
    for key, p in agent.online_net.named_parameters():
        ref = p.grad.detach()
        scale = float(ref.abs().max()) + 1e-12
        err = float((grads[key] - ref).abs().max())
        assert err <= 2e-3 * scale, (key, err, scale)
    learner.close()


def test_hip_r2d2_learner_adam_step_and_trajectory():
    """clip_grad_norm_(40) + Adam(lr, eps) (pyrela/main.py:124-126,233-238) on the flat buffers, two
    consecutive steps on fresh batches, against torch.optim.Adam driven by autograd."""
    import torch

    from rela_amd.learner import HipR2D2Learner

    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    A, B, seq, burn, n = 6, 8, 10, 4, 3
    rng = np.random.default_rng(5)
    agent = _agent(A, n, 0.997, 0.9, seq, burn, 81, 82, "cuda:0")
    lr, eps, clip = 1e-3, 1.5e-4, 0.5  # a clip small enough to bite
    learner = HipR2D2Learner.from_agent(agent, B, lr=lr, eps=eps, grad_clip=clip)
    optim = torch.optim.Adam(agent.online_net.parameters(), lr=lr, eps=eps)
    for step in range(2):
        batch, weight = _random_batch(rng, A, B, seq, burn, n, "cuda:0")
        learner.step(batch, weight)
        loss, _ = agent.loss(batch, sync_priority=False)
        (loss * weight).mean().backward()
        norm = torch.nn.utils.clip_grad_norm_(agent.online_net.parameters(), clip)
        optim.step()
        optim.zero_grad()
        torch.cuda.synchronize()
        st = learner.stats().cpu().numpy()
        np.testing.assert_allclose(st[0], float(norm), rtol=2e-3)
        assert st[1] < 1.0
        sd = learner.state_dict("online")
        for key, p in agent.online_net.named_parameters():
            # an Adam step moves every weight by ~lr; compare the UPDATE, not the weight
            np.testing.assert_allclose(sd[key].cpu().numpy(), p.detach().cpu().numpy(), rtol=0, atol=0.05 * lr,
                                       err_msg="%s after step %d" % (key, step))
    learner.sync_target_with_online()
    torch.cuda.synchronize()
    tg = learner.state_dict("target")
    on = learner.state_dict("online")
    for key in on:
        assert torch.equal(tg[key], on[key])
    learner.close()


def test_hip_r2d2_learner_loss_then_grad_equals_backward():
    """rela_r2d2_learner_loss + rela_r2d2_learner_grad (the step in two halves, so that update_priority and the next
    sample can be queued in between) leave the same loss, priorities and gradients as rela_r2d2_learner_backward,
    bit for bit; grad() without a loss() before it is refused."""
    import torch

    from rela_amd.learner import HipR2D2Learner

    A, B, seq, burn, n = 6, 5, 8, 4, 3
    rng = np.random.default_rng(31)
    agent = _agent(A, n, 0.997, 0.9, seq, burn, 41, 42, "cuda:0")
    batch, weight = _random_batch(rng, A, B, seq, burn, n, "cuda:0")
    learner = HipR2D2Learner.from_agent(agent, B, grad_clip=1e9)
    loss0, prio0, ls0 = learner.backward(batch, weight)
    loss0, prio0, ls0 = loss0.clone(), prio0.clone(), ls0.clone()
    g0 = learner.flat()[1].clone()
    learner.flat()[1].zero_()
    loss1, prio1, ls1 = learner.loss(batch, weight)
    assert torch.equal(loss1, loss0) and torch.equal(prio1, prio0) and torch.equal(ls1, ls0)
    learner.grad()
    learner.check()
    assert torch.equal(learner.flat()[1], g0)
    with pytest.raises(RuntimeError):
        learner.grad()
    learner.close()


@pytest.mark.parametrize("B,seq,burn", [(16, 12, 6), (32, 80, 40)])
def test_hip_r2d2_learner_fast_target_trunk_within_tolerance(B, seq, burn):
    """set_precision("bf16x2"): the target net's conv trunk (no gradient, its activations are never read back) and the
    three large GEMMs of the LSTM's input side -- the gate GEMM of both nets, its data gradient and its weight gradient
    (csrc/gemm_bf16s.h: hi + lo bf16 operands, three MFMAs per product, f32 accumulation) -- run on split-bf16 MFMA;
    loss, priorities and gradients stay within the fast mode's tolerance of the all-f32 step.  T * B = 336 rows and
    240 training rows: every GEMM has a ragged last row block and the weight gradient a zero-padded last k-chunk.
    conv1's weight gradient runs on bf16 MFMA at both shapes (csrc/wgrad_conv1_bf16.h), conv2's and conv3's
    (wgrad_conv2_bf16.h, wgrad_conv3_bf16.h) from 2,048 training frames up: the second shape, 83 x 32 = 2,656."""
    import torch

    from rela_amd.learner import HipR2D2Learner

    A, n = 18, 3
    rng = np.random.default_rng(77)
    agent = _agent(A, n, 0.997, 0.9, seq, burn, 71, 72, "cuda:0")
    batch, weight = _random_batch(rng, A, B, seq, burn, n, "cuda:0")
    learner = HipR2D2Learner.from_agent(agent, B, grad_clip=1e9)
    loss0, prio0, ls0 = learner.backward(batch, weight)
    loss0, prio0 = loss0.clone(), prio0.clone()
    g0 = {k: v.clone() for k, v in learner.state_dict("grads").items()}
    learner.set_precision("bf16x2")  # T * B = 336 frames >= 128: the fast trunk is taken
    from rela_amd import _capi as capi

    with capi.launch_census() as census:
        loss1, prio1, _ = learner.backward(batch, weight)
    learner.check()
    assert R2D2_FAST_KERNELS <= set(census.counts), sorted(census.counts)
    if B * (seq + 3) >= 2048:
        assert {"wgrad_conv2_bf16", "wgrad_conv3_bf16"} <= set(census.counts), sorted(census.counts)
    assert float((prio1 - prio0).abs().max()) < 2e-5
    assert abs(float(loss1) - float(loss0)) < 2e-5 * max(1.0, abs(float(loss0)))
    g1 = learner.state_dict("grads")
    for key in HipR2D2Learner.KEYS:
        # r3: the online trunk runs on split-bf16 MFMA too, so a few ReLUs in 10^5 sit on the other side of zero and
        # every gradient tensor moves by a few 1e-3 of its norm (see tests/test_learner_gpu.py for the mechanism)
        d, n0 = float((g1[key] - g0[key]).norm()), float(g0[key].norm()) + 1e-20
        cos = float((g1[key] * g0[key]).sum()) / (float(g1[key].norm()) * n0 + 1e-30)
        assert d <= 2e-2 * n0 and cos > 0.9998, (key, d / n0, cos)
    learner.set_precision("f32")
    loss2, prio2, _ = learner.backward(batch, weight)
    assert torch.equal(prio2, prio0)
    learner.close()


def test_hip_r2d2_learner_is_bit_reproducible_under_uneven_load():
    """The persistent recurrent kernels hand h_t / the gate gradients from CU to CU inside one launch (write-through
    stores, one counter per step, sc1 loads in place of an acquire fence: csrc/learner_r2d2.hip).  A stale read would be
    a race: it would come and go with timing.  The same batch at C4's shape is therefore differentiated twelve times --
    alone, and next to another stream that keeps the chip unevenly busy with GEMMs and copies -- and loss, priorities
    and every gradient must be bit-identical each time (and match autograd, test_hip_r2d2_learner_matches_autograd)."""
    import torch

    from rela_amd.learner import HipR2D2Learner

    A, B, seq, burn, n = 18, 64, 80, 40, 3
    rng = np.random.default_rng(123)
    agent = _agent(A, n, 0.997, 0.9, seq, burn, 71, 72, "cuda:0")
    batch, weight = _random_batch(rng, A, B, seq, burn, n, "cuda:0")
    learner = HipR2D2Learner.from_agent(agent, B, grad_clip=1e9)
    learner.set_precision("bf16x2")
    side = torch.cuda.Stream()
    x = torch.randn(3072, 3072, device="cuda")
    big = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    ref = None
    for rep in range(12):
        if rep % 2 == 1:  # uneven load: a burst of other work on another stream, started just before the step
            with torch.cuda.stream(side):
                for _ in range(1 + rep % 5):
                    x = (x @ x).clamp_(-1, 1)
                    big.fill_(rep)
        loss, prio, loss_seq = learner.backward(batch, weight)
        learner.check()
        torch.cuda.synchronize()
        cur = (loss.clone(), prio.clone(), loss_seq.clone(), learner.flat()[1].clone())
        if ref is None:
            ref = cur
        else:
            for a, b, name in zip(cur, ref, ("loss", "priority", "loss_seq", "gradients")):
                assert torch.equal(a, b), "run %d differs from run 0 in %s" % (rep, name)
    learner.close()
