"""R2D2 agent for the learner: burn-in unroll, per-timestep n-step TD error, eta-mixed priority.

Arithmetic follows the reference's R2D2Agent (pyrela/r2d2.py:58-206).  As with Ape-X, the actor
side runs HIP kernels; `act` / `compute_priority` here exist for parity tests.
"""
from typing import Dict

import torch
from torch import nn
from torch.nn import functional as F


class R2D2Agent(nn.Module):
    def __init__(self, net_cons, device, multi_step, gamma, eta, seq_len, burn_in, same_hid=0):
        super().__init__()
        self.net_cons = net_cons
        self.multi_step, self.gamma, self.eta = multi_step, gamma, eta
        self.seq_len, self.burn_in, self.same_hid = seq_len, burn_in, same_hid
        self.online_net = net_cons(device)
        self.target_net = net_cons(device)

    def get_h0(self, batchsize: int) -> Dict[str, torch.Tensor]:
        return self.online_net.get_h0(batchsize)

    @classmethod
    def clone(cls, model, device):
        twin = cls(model.net_cons, device, model.multi_step, model.gamma, model.eta, model.seq_len, model.burn_in,
                   model.same_hid)
        twin.load_state_dict(model.state_dict())
        return twin.to(device)

    def sync_target_with_online(self):
        self.target_net.load_state_dict(self.online_net.state_dict())

    @torch.no_grad()
    def act(self, obs, hid):
        greedy, new_hid = self.online_net.act(obs, hid)
        eps = obs["eps"].squeeze(1)
        explore = obs["legal_move"].multinomial(1).squeeze(1)
        coin = (torch.rand(greedy.size(0), device=greedy.device) < eps).long()
        return {"a": (greedy * (1 - coin) + explore * coin).long().cpu()}, new_hid

    @torch.no_grad()
    def compute_priority(self, obs, action, reward, terminal, bootstrap, next_obs, hid, next_hid):
        """One-step priority of a K-row batch (r2d2.py:76-100)."""
        lift = lambda d: {k: v.unsqueeze(0) for k, v in d.items()}
        online_q = self.online_net(lift(obs), hid, action["a"].unsqueeze(0))[0].squeeze(0)
        next_a = self.online_net.act(next_obs, next_hid)[0].unsqueeze(0)
        boot_q = self.target_net(lift(next_obs), next_hid, next_a)[0].squeeze(0)
        target = reward + bootstrap * (self.gamma ** self.multi_step) * boot_q
        return (target - online_q).abs().cpu()

    @torch.no_grad()
    def aggregate_priority(self, priority: torch.Tensor, seq_len: torch.Tensor) -> torch.Tensor:
        """eta * max_t + (1 - eta) * sum_t / (seq_len - burn_in) over the masked row (r2d2.py:103-120)."""
        t = torch.arange(priority.size(1), device=seq_len.device)
        masked = priority * (t.unsqueeze(0) < seq_len.unsqueeze(1)).float()
        mean = masked.sum(1) / (seq_len - self.burn_in)
        return (self.eta * masked.max(1)[0] + (1.0 - self.eta) * mean).cpu()

    def td_err(self, obs, hid, action, reward, terminal, bootstrap, seq_len) -> torch.Tensor:
        """[batch, seq_len] TD errors of the training part of each sequence (r2d2.py:122-187)."""
        terminal = terminal.float()
        b = self.burn_in
        warm = {k: v[:b] for k, v in obs.items()}
        train = {k: v[b:] for k, v in obs.items()}
        if b == 0:
            on_hid, tg_hid = hid, hid
        else:
            with torch.no_grad():
                _, on_hid = self.online_net.unroll_rnn(warm, hid)
                _, tg_hid = self.target_net.unroll_rnn(warm, hid)
            keep = (1 - terminal[b - 1]).unsqueeze(0).unsqueeze(2)  # dummy burn-in at episode start
            on_hid = {k: v * keep for k, v in on_hid.items()}
            tg_hid = {k: v * keep for k, v in tg_hid.items()}
        a_train = action["a"][b:]
        online_qa, greedy = self.online_net(train, on_hid, a_train)
        with torch.no_grad():
            target_qa, _ = self.target_net(train, tg_hid, greedy)
        reward, bootstrap = reward[b:], bootstrap[b:]
        gamma_n = self.gamma ** self.multi_step
        cols = []
        for i in range(self.seq_len):
            target = reward[i] + bootstrap[i] * (gamma_n * target_qa[i + self.multi_step])
            pad = (i >= (seq_len - b)).float()
            cols.append((target.detach() - online_qa[i]) * (1 - pad))
        return torch.stack(cols, 1)

    def loss(self, batch, sync_priority: bool = True):
        err = self.td_err(batch.obs, batch.h0, batch.action, batch.reward, batch.terminal, batch.bootstrap,
                          batch.seq_len)
        per_seq = F.smooth_l1_loss(err, torch.zeros_like(err), reduction="none").sum(1)
        prio = self.aggregate_priority(err.detach().abs(), batch.seq_len)
        return per_seq, (prio if sync_priority else prio.to(err.device))
