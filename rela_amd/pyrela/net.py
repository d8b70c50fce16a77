"""Learner-side Atari Q-network (PyTorch-ROCm autograd).

Parameter contract = the reference's AtariFFNet state_dict (pyrela/net.py:18-31):
  net.{0,2,4}.{weight,bias}  (32,4,8,8) / (64,32,4,4) / (64,64,3,3)
  linear.0.{weight,bias}     (512,3136)
  fc_v.{weight,bias}         (1,512)      fc_a.{weight,bias}  (A,512)
so weights move freely between this module, the reference's agents and the HIP actor
(rela_amd/csrc/ffnet.hip, which consumes exactly these tensors through rela_ffnet_load).
"""
from typing import Dict

import torch
from torch import nn

FRAME_STACK = 4
FLAT = 64 * 7 * 7  # 3136
HIDDEN = 512


def _conv_trunk() -> nn.Sequential:
    spec = [(FRAME_STACK, 32, 8, 4), (32, 64, 4, 2), (64, 64, 3, 1)]
    layers = []
    for cin, cout, k, stride in spec:
        layers += [nn.Conv2d(cin, cout, k, stride=stride), nn.ReLU()]
    return nn.Sequential(*layers)


def dueling_q(v: torch.Tensor, a: torch.Tensor, legal: torch.Tensor, dim: int) -> torch.Tensor:
    """q = v + a*legal - mean(a*legal): the mean runs over ALL actions, legal or not (net.py:33-39)."""
    masked = a * legal
    return v + masked - masked.mean(dim, keepdim=True)


class AtariFFNet(nn.Module):
    def __init__(self, num_action: int):
        super().__init__()
        self.num_action = num_action
        self.net = _conv_trunk()
        self.linear = nn.Sequential(nn.Linear(FLAT, HIDDEN), nn.ReLU())
        self.fc_v = nn.Linear(HIDDEN, 1)
        self.fc_a = nn.Linear(HIDDEN, num_action)

    def forward(self, obs: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Q-values [N, A]; entries of illegal moves are unspecified, as in the reference."""
        x = obs["s"].float() / 255.0
        feat = self.net(x).flatten(1)
        hid = self.linear(feat)
        return dueling_q(self.fc_v(hid), self.fc_a(hid), obs["legal_move"], 1)


class AtariLSTMNet(nn.Module):
    """R2D2 network: conv trunk -> one LSTM layer (3136 -> 512) -> dueling heads.

    Parameter contract = the reference's AtariLSTMNet state_dict (pyrela/net.py:69-84; SURVEY 8a
    N2): net.{0,2,4}.*, lstm.{weight_ih_l0 (2048,3136), weight_hh_l0 (2048,512), bias_ih_l0,
    bias_hh_l0 (2048)}, fc_v.*, fc_a.*; torch gate order i,f,g,o.
    """

    def __init__(self, device, num_action: int):
        super().__init__()
        self.num_action = num_action
        self.net = _conv_trunk()
        self.lstm = nn.LSTM(FLAT, HIDDEN, num_layers=1).to(device)
        self.fc_v = nn.Linear(HIDDEN, 1)
        self.fc_a = nn.Linear(HIDDEN, num_action)

    def get_h0(self, batchsize: int) -> Dict[str, torch.Tensor]:
        z = torch.zeros(1, batchsize, HIDDEN)
        return {"h0": z, "c0": z.clone()}

    def _features(self, s: torch.Tensor) -> torch.Tensor:
        return self.net(s.float() / 255.0).flatten(1)

    def act(self, obs: Dict[str, torch.Tensor], hid: Dict[str, torch.Tensor]):
        """One step; ranks the raw ADVANTAGES shifted by their batch minimum (net.py:110-124)."""
        x = self._features(obs["s"]).unsqueeze(0)
        o, (h, c) = self.lstm(x, (hid["h0"], hid["c0"]))
        adv = self.fc_a(o).squeeze(0)
        greedy = ((1 + adv - adv.min()) * obs["legal_move"]).argmax(1)
        return greedy.detach(), {"h0": h.detach(), "c0": c.detach()}

    def unroll_rnn(self, obs: Dict[str, torch.Tensor], hid: Dict[str, torch.Tensor]):
        s = obs["s"]
        seq, batch = s.shape[:2]
        x = self._features(s.reshape(seq * batch, *s.shape[2:])).view(seq, batch, FLAT)
        o, (h, c) = self.lstm(x, (hid["h0"], hid["c0"]))
        return o, {"h0": h, "c0": c}

    def forward(self, obs, hid, action):
        """-> (Q(s_t, a_t) [seq,batch], greedy action [seq,batch]) for a [seq,batch,...] unroll."""
        o, _ = self.unroll_rnn(obs, hid)
        q = dueling_q(self.fc_v(o), self.fc_a(o), obs["legal_move"], 2)
        qa = q.gather(2, action.unsqueeze(2)).squeeze(2)
        greedy = ((1 + q - q.min()) * obs["legal_move"]).argmax(2)
        return qa, greedy.detach()
