"""CPU checks of the PyTorch learner path (`--hip_learner 0`, rela_amd/pyrela/apex.py) against the vectors
recorded from the REAL reference's learner step (tests/golden/learner_*.json, make_golden.py learner)."""
import glob
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
GOLD = os.path.join(HERE, "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "learner_*.json"))), ids=os.path.basename)
def test_autograd_learner_step_matches_reference_golden(path):
    import torch
    from types import SimpleNamespace

    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet
    from synth import synth_obs, synth_params

    g = json.load(open(path))
    A, B = g["num_action"], g["B"]
    torch.set_num_threads(4)
    agent = ApexAgent(lambda: AtariFFNet(A), g["multi_step"], g["gamma"])
    agent.online_net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_params(A, g["online_seed"]).items()})
    agent.target_net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_params(A, g["target_seed"]).items()})
    f32 = lambda x: torch.tensor(x, dtype=torch.float32)
    batch = SimpleNamespace(
        obs={"s": torch.from_numpy(synth_obs(B, g["obs_seed"])), "eps": torch.zeros(B, 1), "legal_move": f32(g["legal"])},
        next_obs={"s": torch.from_numpy(synth_obs(B, g["next_obs_seed"])), "eps": torch.zeros(B, 1),
                  "legal_move": f32(g["next_legal"])},
        action={"a": torch.tensor(g["action"], dtype=torch.int64)}, reward=f32(g["reward"]),
        terminal=torch.zeros(B, dtype=torch.bool), bootstrap=f32(g["bootstrap"]))
    params = list(agent.online_net.parameters())
    optim = torch.optim.RMSprop(params, lr=g["lr"], eps=g["eps"])
    per_sample, prio = agent.loss(batch)
    loss = (per_sample * f32(g["weight"])).mean()
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(prio.numpy(), np.array(g["priority"]), rtol=1e-5, atol=1e-6)
    named = dict(agent.online_net.named_parameters())
    for key, rec in g["grads"].items():
        t = named[key].grad.detach().double().reshape(-1)
        np.testing.assert_allclose(float(t.norm()), rec["l2"], rtol=1e-4, err_msg=key)
        np.testing.assert_allclose(t[torch.tensor(rec["idx"])].numpy(), np.array(rec["val"]), rtol=1e-3,
                                   atol=1e-4 * (rec["absmax"] + 1e-12), err_msg=key)
    gnorm = torch.nn.utils.clip_grad_norm_(params, g["grad_clip"])
    np.testing.assert_allclose(gnorm.item(), g["grad_norm"], rtol=1e-4)
    optim.step()
    for key, rec in g["params_after"].items():
        t = named[key].detach().double().reshape(-1)
        np.testing.assert_allclose(t[torch.tensor(rec["idx"])].numpy(), np.array(rec["val"]), rtol=0,
                                   atol=2e-2 * 10 * g["lr"] * (rec["absmax"] + 1e-12) + 1e-7, err_msg=key)
