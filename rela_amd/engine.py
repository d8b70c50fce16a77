"""Device-resident Ape-X actor engine: the work of T actor threads x K envs as batched launches.

One engine row = one env.  Rows are grouped in blocks of K (`group_rows`) that correspond to the
K envs of one reference actor thread, so every batch-global reduction of the reference (the
q.min() of greedy_act, apex.py:51) keeps its original scope.  Per env-step the engine runs what
BasicThreadLoop::mainLoop (rela/thread_loop.h:74-105) makes DQNActor do:

  act       (dqn_actor.h:153-171)  Q(obs) -> eps-greedy action                    1 trunk forward
  post_step (dqn_actor.h:181-203)  n-step pop (:58-106) -> TD priority (apex.py:68-78)
                                   -> replay add (prioritized_replay.h:186-200)   3 trunk forwards

The observation history (n+1 frame stacks per env) lives in HBM, so obs_t / obs_{t+n} are never
re-uploaded for the priority pass and the replay insert is a device-to-device row copy.
"""
import ctypes as C

import torch

from . import _capi as capi
from .replay import OBS_BYTES


class FFNetHandle:
    """One immutable device copy of AtariFFNet parameters in kernel layout (rela_ffnet_*)."""

    KEYS = ("net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias",
            "linear.0.weight", "linear.0.bias", "fc_v.weight", "fc_v.bias", "fc_a.weight", "fc_a.bias")

    def __init__(self, num_action, device="cuda:0"):
        self.device = torch.device(device)
        self.num_action = num_action
        h = C.c_void_p()
        capi.check(capi.lib.rela_ffnet_create(C.byref(h), num_action, self.device.index or 0), "rela_ffnet_create")
        self.h = h

    def load_state_dict(self, sd, prefix=""):
        """sd: a (sub-)state_dict with the N1 keys; tensors may live on the CPU or on this GPU."""
        p = capi.FFNetParams()
        keep = []
        for (field, _), key in zip(capi.FFNetParams._fields_, self.KEYS):
            t = sd[prefix + key].detach().to(self.device, torch.float32).contiguous()
            keep.append(t)
            setattr(p, field, t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        capi.check(capi.lib.rela_ffnet_load(self.h, C.byref(p), 1, stream), "rela_ffnet_load")
        self._keep = keep  # packing kernels are stream-ordered; keep sources alive until the next load

    def close(self):
        if getattr(self, "h", None):
            capi.lib.rela_ffnet_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class ApexActorEngine:
    def __init__(self, rows, group_rows, num_action, multi_step, gamma, replay, eps, device="cuda:0", seed=1):
        assert rows % group_rows == 0
        self.R, self.K, self.A, self.n = rows, group_rows, num_action, multi_step
        self.gamma = float(gamma)
        self.gamma_n = float(torch.tensor(self.gamma ** multi_step, dtype=torch.float32))
        self.replay = replay
        self.device = torch.device(device)
        self.seed = seed
        dev = self.device
        H = multi_step + 1
        self.obs_hist = torch.zeros((H, rows, 4, 84, 84), dtype=torch.uint8, device=dev)
        self.act_hist = torch.zeros((H, rows), dtype=torch.int64, device=dev)
        self.rew_hist = torch.zeros((H, rows), dtype=torch.float32, device=dev)
        self.term_hist = torch.zeros((H, rows), dtype=torch.uint8, device=dev)
        self.eps = torch.as_tensor(eps, dtype=torch.float32, device=dev).reshape(rows, 1).contiguous()
        self.legal = torch.ones((rows, num_action), dtype=torch.float32, device=dev)
        self.q = torch.empty((4, rows, num_action), dtype=torch.float32, device=dev)
        self.out_r = torch.empty(rows, dtype=torch.float32, device=dev)
        self.out_b = torch.empty(rows, dtype=torch.float32, device=dev)
        self.out_t = torch.empty(rows, dtype=torch.uint8, device=dev)
        self.prio = torch.empty(rows, dtype=torch.float32, device=dev)
        self.ws_bytes = 0
        self.ws = None
        self.count = 0      # entries currently in the history (<= n+1)
        self.head = 0       # ring row of the oldest entry
        self.num_act = 0
        self.act_calls = 0

    # -- plumbing -----------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _forward(self, net, obs, q_out):
        if self.ws is None:
            self.ws_bytes = capi.lib.rela_ffnet_workspace_bytes(net.h, self.R)
            self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)
        capi.check(capi.lib.rela_ffnet_forward(net.h, self.R, C.c_void_p(obs.data_ptr()),
                                               C.c_void_p(self.legal.data_ptr()), C.c_void_p(q_out.data_ptr()),
                                               C.c_void_p(self.ws.data_ptr()), self.ws_bytes, self._stream()),
                   "rela_ffnet_forward")

    def next_obs_slot(self):
        """The HBM slot the env layer writes the next observation batch into ([R,4,84,84] u8)."""
        return self.obs_hist[(self.head + self.count) % (self.n + 1)]

    # -- DQNActor::act ------------------------------------------------------------------
    def act(self, online):
        """Consumes the observation already written to next_obs_slot(); returns actions (cuda i64[R])."""
        slot = (self.head + self.count) % (self.n + 1)
        self._forward(online, self.obs_hist[slot], self.q[0])
        a = self.act_hist[slot]
        capi.check(capi.lib.rela_apex_act_from_q(self.R, self.A, self.K, C.c_void_p(self.q[0].data_ptr()),
                                                 C.c_void_p(self.legal.data_ptr()), C.c_void_p(self.eps.data_ptr()),
                                                 self.seed, self.act_calls * self.R, C.c_void_p(a.data_ptr()),
                                                 self._stream()), "rela_apex_act_from_q")
        self.act_calls += 1
        self.num_act += self.R
        self._cur = slot
        return a

    # -- setRewardAndTerminal + postStep --------------------------------------------------
    def post_step(self, reward, terminal, online, target, nonblocking=False):
        """reward f32[R], terminal u8/bool[R] on the device.  Returns True if a block was inserted."""
        self.rew_hist[self._cur].copy_(reward)
        self.term_hist[self._cur].copy_(terminal)
        self.count += 1
        if self.count < self.n + 1:
            return False
        H = self.n + 1
        first, last = self.head, (self.head + self.n) % H
        s = self._stream()
        capi.check(capi.lib.rela_nstep_return(self.n, self.R, self.gamma, first, C.c_void_p(self.rew_hist.data_ptr()),
                                              C.c_void_p(self.term_hist.data_ptr()), C.c_void_p(self.out_r.data_ptr()),
                                              C.c_void_p(self.out_b.data_ptr()), C.c_void_p(self.out_t.data_ptr()), s),
                   "rela_nstep_return")
        obs_t, obs_n = self.obs_hist[first], self.obs_hist[last]
        self._forward(online, obs_t, self.q[1])   # online_net(obs)       apex.py:38
        self._forward(online, obs_n, self.q[2])   # greedy_act(next_obs)  apex.py:41
        self._forward(target, obs_n, self.q[3])   # target_net(next_obs)  apex.py:42
        capi.check(capi.lib.rela_apex_td_from_q(self.R, self.A, self.K, C.c_void_p(self.q[1].data_ptr()),
                                                C.c_void_p(self.q[2].data_ptr()), C.c_void_p(self.q[3].data_ptr()),
                                                C.c_void_p(self.legal.data_ptr()),
                                                C.c_void_p(self.act_hist[first].data_ptr()),
                                                C.c_void_p(self.out_r.data_ptr()), C.c_void_p(self.out_b.data_ptr()),
                                                C.c_float(self.gamma_n), None, C.c_void_p(self.prio.data_ptr()), s),
                   "rela_apex_td_from_q")
        ptrs = [obs_t.data_ptr(), obs_n.data_ptr(), self.eps.data_ptr(), self.eps.data_ptr(), self.legal.data_ptr(),
                self.legal.data_ptr(), self.act_hist[first].data_ptr(), self.out_r.data_ptr(), self.out_t.data_ptr(),
                self.out_b.data_ptr()]
        rc = self.replay.add_rows(self.R, ptrs, self.prio, nonblocking=nonblocking)
        self.head = (self.head + 1) % H  # pop_front, dqn_actor.h:101-104
        self.count -= 1
        return rc == 0
