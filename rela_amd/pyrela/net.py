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
