"""Device-resident Ape-X actor engine: the work of T actor threads x K envs as batched launches.

One engine row = one env.  Rows are grouped in blocks of K (`group_rows`) that correspond to the
K envs of one reference actor thread, so every batch-global reduction of the reference (the
q.min() of greedy_act, apex.py:51) keeps its original scope.  Per env-step the engine runs what
BasicThreadLoop::mainLoop (rela/thread_loop.h:74-105) makes DQNActor do:

  act       (dqn_actor.h:153-171)  Q(obs) -> eps-greedy action                    1 trunk forward
  post_step (dqn_actor.h:181-203)  n-step pop (:58-106) -> TD priority (apex.py:68-78)
                                   -> replay add (prioritized_replay.h:186-200)   2 trunk forwards (+1 if the weights
                                                                                  changed since act(): the third is act's own)

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

    def set_precision(self, mode):
        """"f32" (default: exact f32 MFMA), "f32x3" (conv2 / conv3 / fc with both f32 operands as three bf16 parts on
        the bf16 matrix cores: f32 accuracy, csrc/gemm_f32emu.h) or "bf16x2" (the fast mode: two bf16 parts, 16-bit
        significands, |dQ| < 2e-5 max|Q|); include/rela_amd.h rela_ffnet_set_precision."""
        capi.check(capi.lib.rela_ffnet_set_precision(self.h, {"f32": 0, "bf16x2": 1, "f32x3": 2}[mode]),
                   "rela_ffnet_set_precision")

    def close(self):
        if getattr(self, "h", None):
            capi.lib.rela_ffnet_destroy(self.h)
            self.h = None

    def __del__(self):
        if capi is not None and getattr(capi, "lib", None) is not None:  # module globals die first at exit
            self.close()


class _DevMem:
    """Exposes a raw device allocation owned by the C library through __cuda_array_interface__."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def dev_view(ptr, shape, dtype, device):
    typestr = {torch.uint8: "|u1", torch.float32: "<f4", torch.int64: "<i8", torch.int32: "<i4"}[dtype]
    return torch.as_tensor(_DevMem(ptr, shape, typestr), device=device)


class ApexActorEngine:
    """Thin Python handle on the native actor shard (rela_apex_actor_*, csrc/actor.hip)."""

    def __init__(self, rows, group_rows, num_action, multi_step, gamma, replay, eps, device="cuda:0", seed=1):
        self.R, self.K, self.A, self.n = rows, group_rows, num_action, multi_step
        self.replay = replay
        self.device = torch.device(device)
        h = C.c_void_p()
        capi.check(capi.lib.rela_apex_actor_create(C.byref(h), rows, group_rows, num_action, multi_step, gamma,
                                                   replay.h if replay is not None else None, seed,
                                                   self.device.index or 0), "rela_apex_actor_create")
        self.h = h
        if replay is not None and getattr(replay, "dedup", None):  # frame-stack de-duplication on the way in
            capi.check(capi.lib.rela_apex_actor_set_dedup(h, {"stack": 1, "plane": 4}[replay.dedup]),
                       "rela_apex_actor_set_dedup")
        dev = self.device
        base = capi.lib.rela_apex_actor_obs_slot(h)  # head = count = 0 -> slot 0 = base of the history
        self.obs_hist = dev_view(base, (multi_step + 1, rows, 4, 84, 84), torch.uint8, dev)
        self.eps = dev_view(capi.lib.rela_apex_actor_eps_dev(h), (rows, 1), torch.float32, dev)
        self.legal = dev_view(capi.lib.rela_apex_actor_legal_dev(h), (rows, num_action), torch.float32, dev)
        self._rows, self._A = rows, num_action
        self.prio = dev_view(capi.lib.rela_apex_actor_last_priority_dev(h), (rows,), torch.float32, dev)
        self.eps.copy_(torch.as_tensor(eps, dtype=torch.float32).reshape(rows, 1))
        self._slot = 0

    def close(self):
        if getattr(self, "h", None):
            capi.lib.rela_apex_actor_destroy(self.h)
            self.h = None

    def __del__(self):
        if capi is not None and getattr(capi, "lib", None) is not None:  # module globals die first at exit
            self.close()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def num_act(self):
        return capi.lib.rela_apex_actor_num_act(self.h)

    @property
    def q(self):
        """[1, rows, A] view of the Q table of the last act() (rela_apex_actor_last_q_dev: the table lives in the
        history slot act() wrote, so the pointer is looked up per call)."""
        return dev_view(capi.lib.rela_apex_actor_last_q_dev(self.h), (1, self._rows, self._A), torch.float32, self.device)

    def set_reuse(self, on):
        """on=0 / False: post_step always recomputes (the reference's 4 forwards per step); 1 / True: reuses act()'s
        forwards of this tick and of n ticks ago; 2: only the one of this tick."""
        capi.check(capi.lib.rela_apex_actor_set_reuse(self.h, int(on)), "rela_apex_actor_set_reuse")

    def next_obs_slot(self):
        """The HBM slot the env layer writes the next observation batch into ([R,4,84,84] u8)."""
        ptr = capi.lib.rela_apex_actor_obs_slot(self.h)
        return dev_view(ptr, (self.R, 4, 84, 84), torch.uint8, self.device)

    def act(self, online):
        """DQNActor::act on the observation already written to next_obs_slot(); cuda i64[R]."""
        out = C.c_void_p()
        capi.check(capi.lib.rela_apex_actor_act(self.h, online.h, None, None, None, None, C.byref(out), self._stream()),
                   "rela_apex_actor_act")
        return dev_view(out.value, (self.R,), torch.int64, self.device)

    def post_step(self, reward, terminal, online, target, nonblocking=False):
        """setRewardAndTerminal + postStep; reward f32[R] / terminal u8[R] on the device."""
        ins = C.c_int(0)
        self._keep = (reward, terminal)
        rc = capi.lib.rela_apex_actor_post_step(self.h, C.c_void_p(reward.data_ptr()), C.c_void_p(terminal.data_ptr()),
                                                1, online.h, target.h, int(nonblocking), C.byref(ins), self._stream())
        if rc != capi.EWOULDBLOCK:
            capi.check(rc, "rela_apex_actor_post_step")
        return bool(ins.value)


class LSTMNetHandle:
    """One immutable device copy of AtariLSTMNet parameters in kernel layout (rela_lstmnet_*)."""

    KEYS = ("net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias",
            "lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0", "fc_v.weight", "fc_v.bias",
            "fc_a.weight", "fc_a.bias")

    def __init__(self, num_action, device="cuda:0"):
        self.device = torch.device(device)
        self.num_action = num_action
        h = C.c_void_p()
        capi.check(capi.lib.rela_lstmnet_create(C.byref(h), num_action, self.device.index or 0), "rela_lstmnet_create")
        self.h = h

    def load_state_dict(self, sd, prefix=""):
        p = capi.LSTMNetParams()
        keep = []
        for (field, _), key in zip(capi.LSTMNetParams._fields_, self.KEYS):
            t = sd[prefix + key].detach().to(self.device, torch.float32).contiguous()
            keep.append(t)
            setattr(p, field, t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        capi.check(capi.lib.rela_lstmnet_load(self.h, C.byref(p), 1, stream), "rela_lstmnet_load")
        self._keep = keep

    def set_precision(self, mode):
        """"f32" (default) or "bf16x2": the conv trunk on split-bf16 MFMA for batches of 128 rows and more and, from
        1,024 rows up, the input side of the LSTM gate GEMM (h x W_hh, the cell and the heads stay f32);
        rela_lstmnet_set_precision."""
        capi.check(capi.lib.rela_lstmnet_set_precision(self.h, {"f32": 0, "bf16x2": 1, "f32x3": 2}[mode]),
                   "rela_lstmnet_set_precision")  # (f32x3: conv2 / conv3 of the trunk, from 512 rows; the rest exact f32)

    def close(self):
        if getattr(self, "h", None):
            capi.lib.rela_lstmnet_destroy(self.h)
            self.h = None

    def __del__(self):
        if capi is not None and getattr(capi, "lib", None) is not None:
            self.close()


class R2D2ActorEngine:
    """Thin Python handle on the native R2D2 actor shard (rela_r2d2_actor_*, csrc/actor_r2d2.hip): `rows` envs in
    groups of K with their n-step rings, recurrent state and sequence windows in HBM."""

    def __init__(self, rows, group_rows, num_action, multi_step, gamma, seq_len, burn_in, eta, replay, eps,
                 device="cuda:0", seed=1):
        self.R, self.K, self.A = rows, group_rows, num_action
        self.device = torch.device(device)
        self.replay = replay
        h = C.c_void_p()
        capi.check(capi.lib.rela_r2d2_actor_create(C.byref(h), rows, group_rows, num_action, multi_step, gamma, seq_len,
                                                   burn_in, float(eta), replay.h if replay is not None else None, seed,
                                                   self.device.index or 0), "rela_r2d2_actor_create")
        self.h = h
        self._eps = torch.as_tensor(eps, dtype=torch.float32).reshape(rows).contiguous()
        self._legal = torch.ones((rows, num_action), dtype=torch.float32)
        self._first = True

    def close(self):
        if getattr(self, "h", None):
            capi.lib.rela_r2d2_actor_destroy(self.h)
            self.h = None

    def __del__(self):
        if capi is not None and getattr(capi, "lib", None) is not None:
            self.close()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def num_act(self):
        return capi.lib.rela_r2d2_actor_num_act(self.h)

    def set_reuse(self, on):
        capi.check(capi.lib.rela_r2d2_actor_set_reuse(self.h, int(on)), "rela_r2d2_actor_set_reuse")

    def next_obs_slot(self):
        return dev_view(capi.lib.rela_r2d2_actor_obs_slot(self.h), (self.R, 4, 84, 84), torch.uint8, self.device)

    def act(self, online):
        """R2D2Actor::act on the observation already resident in next_obs_slot(); cuda i64[R]."""
        out = C.c_void_p()
        e = C.c_void_p(self._eps.data_ptr()) if self._first else None
        l = C.c_void_p(self._legal.data_ptr()) if self._first else None
        capi.check(capi.lib.rela_r2d2_actor_act(self.h, online.h, None, e, l, None, C.byref(out), self._stream()),
                   "rela_r2d2_actor_act")
        self._first = False
        return dev_view(out.value, (self.R,), torch.int64, self.device)

    def post_step(self, reward_host, terminal_host, online, target, nonblocking=False):
        """setRewardAndTerminal + postStep; reward f32[R] / terminal u8[R] are HOST numpy arrays (the window
        bookkeeping branches on the terminal flags).  -> sequences appended by this call."""
        n = C.c_int(0)
        rc = capi.lib.rela_r2d2_actor_post_step(self.h, reward_host.ctypes.data_as(C.c_void_p),
                                                terminal_host.ctypes.data_as(C.c_void_p), online.h, target.h,
                                                int(nonblocking), C.byref(n), self._stream())
        if rc != capi.EWOULDBLOCK:
            capi.check(rc, "rela_r2d2_actor_post_step")
        return n.value
