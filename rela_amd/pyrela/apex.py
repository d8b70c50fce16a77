"""Ape-X agent for the learner: double-DQN n-step TD error, Huber loss, priorities.

Arithmetic follows the reference's ApexAgent (pyrela/apex.py:30-91); the actor-side methods
(`act`, `compute_priority`) exist here for API parity and tests, but in this engine the actors
run the HIP kernels (rela_amd/csrc/ffnet.hip + agent_ops.hip), not these methods.
"""
import copy
from typing import Dict

import torch
from torch import nn
from torch.nn import functional as F


def masked_greedy(q: torch.Tensor, legal: torch.Tensor) -> torch.Tensor:
    """argmax over legal moves; the shift uses the minimum of the WHOLE batch (apex.py:48-54)."""
    shifted = (1 + q - q.min()) * legal
    return shifted.argmax(1)


class ApexAgent(nn.Module):
    def __init__(self, net_cons, multi_step: int, gamma: float):
        super().__init__()
        self.net_cons = net_cons
        self.multi_step = multi_step
        self.gamma = gamma
        self.online_net = net_cons()
        self.target_net = net_cons()

    @classmethod
    def clone(cls, model, device):
        twin = cls(model.net_cons, model.multi_step, model.gamma)
        twin.load_state_dict(model.state_dict())
        return twin.to(device)

    def sync_target_with_online(self):
        self.target_net.load_state_dict(self.online_net.state_dict())

    def greedy_act(self, obs: Dict[str, torch.Tensor]) -> torch.Tensor:
        with torch.no_grad():
            return masked_greedy(self.online_net(obs), obs["legal_move"])

    def td_err(self, obs, action, reward, bootstrap, next_obs) -> torch.Tensor:
        q_taken = self.online_net(obs).gather(1, action["a"].unsqueeze(1)).squeeze(1)
        with torch.no_grad():
            next_a = masked_greedy(self.online_net(next_obs), next_obs["legal_move"])
            next_q = self.target_net(next_obs).gather(1, next_a.unsqueeze(1)).squeeze(1)
            target = reward + bootstrap * (self.gamma ** self.multi_step) * next_q
        return target - q_taken

    @torch.no_grad()
    def act(self, obs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        greedy = self.greedy_act(obs)
        eps = obs["eps"].squeeze(1)
        explore = obs["legal_move"].multinomial(1).squeeze(1)
        coin = (torch.rand(greedy.size(0), device=greedy.device) < eps).long()
        return {"a": (greedy * (1 - coin) + explore * coin).long().cpu()}

    @torch.no_grad()
    def compute_priority(self, obs, action, reward, terminal, bootstrap, next_obs) -> torch.Tensor:
        return self.td_err(obs, action, reward, bootstrap, next_obs).abs().cpu()

    def loss(self, batch, sync_priority: bool = True):
        """Per-sample Huber loss and |td| priority.  sync_priority=False keeps the priority on
        the device (the C ABI accepts a device pointer), removing the reference's per-step
        `.cpu()` sync (apex.py:90)."""
        err = self.td_err(batch.obs, batch.action, batch.reward, batch.bootstrap, batch.next_obs)
        per_sample = F.smooth_l1_loss(err, torch.zeros_like(err), reduction="none")
        prio = err.detach().abs()
        return per_sample, (prio.cpu() if sync_priority else prio)
