"""Python view of the device-resident prioritized replay (C ABI: include/rela_amd.h).

`FFReplay` stores FFTransition records (rela/types.h:18-51) as ten SoA fields and hands batches
back as torch tensors that alias preallocated device buffers (zero host copies).  It mirrors
FFPrioritizedReplay's Python surface (rela/pybind.cc:37-47): size / num_add / sample /
update_priority; `add` is reachable only from the actor engine, as in the reference.
"""
import ctypes as C
from types import SimpleNamespace

import torch

from . import _capi as capi

OBS_BYTES = 4 * 84 * 84


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class FFReplay:
    """FFPrioritizedReplay(capacity, seed, alpha, beta, prefetch) -- prioritized_replay.h:175-184."""

    FIELDS = ("s", "next_s", "eps", "next_eps", "legal_move", "next_legal_move", "a", "reward", "terminal",
              "bootstrap")

    def __init__(self, capacity, seed, alpha, beta, prefetch, num_action, device="cuda:0", dedup=None,
                 guard_units=0):
        """dedup: None (s and next_s stored in full, 2 x 28,224 B per transition), "stack" (every stack stored once:
        28,224 B per env-step) or "plane" (one new 84x84 plane per env-step: 7,056 B; only for envs that slide their
        frame stack like atari/game_state.h:53-82).  guard_units: the units producers store ahead of the transitions
        that use them, (multi_step + 8) * rows of all producers is always enough."""
        self.device = torch.device(device)
        self.num_action = num_action
        h = C.c_void_p()
        capi.check(capi.lib.rela_replay_create(C.byref(h), capacity, seed, alpha, beta, prefetch,
                                               self.device.index or 0), "rela_replay_create")
        self.h = h
        A = num_action
        self.row_bytes = [OBS_BYTES, OBS_BYTES, 4, 4, 4 * A, 4 * A, 8, 4, 1, 4]
        rb = (C.c_int64 * len(self.row_bytes))(*self.row_bytes)
        self.dedup = dedup
        if dedup is None:
            capi.check(capi.lib.rela_replay_set_schema(h, len(self.row_bytes), rb), "rela_replay_set_schema")
        else:
            ups = {"stack": 1, "plane": 4}[dedup]
            capi.check(capi.lib.rela_replay_set_schema_dedup(h, len(self.row_bytes), rb, 0, 1, OBS_BYTES // ups, ups,
                                                             int(guard_units)), "rela_replay_set_schema_dedup")
        self._out = {}
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            capi.lib.rela_replay_destroy(self.h)
            self.h = None

    def __del__(self):
        if capi is not None and getattr(capi, "lib", None) is not None:  # module globals die first at exit
            self.close()

    def size(self):
        return capi.lib.rela_replay_size(self.h)

    def num_add(self):
        return capi.lib.rela_replay_num_add(self.h)

    # -- actor side ---------------------------------------------------------------------
    def add_rows(self, n, ptrs, priority, nonblocking=False):
        """ptrs: ten device pointers in FIELDS order, n rows each; priority: cuda f32[n]."""
        rows = (C.c_void_p * len(ptrs))(*ptrs)
        rc = capi.lib.rela_replay_add(self.h, n, rows, C.c_void_p(priority.data_ptr()), int(nonblocking),
                                      _stream_ptr(self.device))
        if rc != capi.EWOULDBLOCK:
            capi.check(rc, "rela_replay_add")
        return rc

    # -- learner side -------------------------------------------------------------------
    def _buffers(self, batch, slot=0):
        if (batch, slot) not in self._out:
            dev, A = self.device, self.num_action
            mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
            # next_s right behind s, next_legal_move right behind legal_move: the f32x3 learner then runs the online net over
            # [s ; s'] as ONE forward without copying anything (csrc/learner.hip: rela_apex_learner_loss)
            frames, legal = mk((2, batch, 4, 84, 84), torch.uint8), mk((2, batch, A), torch.float32)
            self._out[(batch, slot)] = dict(
                s=frames[0], next_s=frames[1],
                eps=mk((batch, 1), torch.float32), next_eps=mk((batch, 1), torch.float32),
                legal_move=legal[0], next_legal_move=legal[1],
                a=mk((batch,), torch.int64), reward=mk((batch,), torch.float32), terminal=mk((batch,), torch.bool),
                bootstrap=mk((batch,), torch.float32), weight=mk((batch,), torch.float32))
        return self._out[(batch, slot)]

    def sample(self, batchsize, device=None, gather=True, slot=0):
        """-> (FFTransition-like namespace, IS weights); tensors live on the replay's GPU.  The batch is written into
        the output buffers of `slot`: a caller that samples the next batch while the previous one is still being
        read (HipApexLearner.loss / grad) alternates between two slots."""
        b = self._buffers(batchsize, slot)
        rows = (C.c_void_p * len(self.FIELDS))(*[b[f].data_ptr() for f in self.FIELDS]) if gather else None
        capi.check(capi.lib.rela_replay_sample(self.h, batchsize, rows, C.c_void_p(b["weight"].data_ptr()),
                                               _stream_ptr(self.device)), "rela_replay_sample")
        batch = SimpleNamespace(
            obs={"s": b["s"], "eps": b["eps"], "legal_move": b["legal_move"]}, action={"a": b["a"]},
            reward=b["reward"], terminal=b["terminal"], bootstrap=b["bootstrap"],
            next_obs={"s": b["next_s"], "eps": b["next_eps"], "legal_move": b["next_legal_move"]})
        return batch, b["weight"]

    def update_priority(self, priority):
        """CPU tensor as in the reference, or a CUDA tensor (no host sync)."""
        p = priority.detach().contiguous().float()
        self._keep = p
        if p.is_cuda:
            rc = capi.lib.rela_replay_update_priority(self.h, p.numel(), C.c_void_p(p.data_ptr()), 1,
                                                      _stream_ptr(self.device))
        else:
            rc = capi.lib.rela_replay_update_priority(self.h, p.numel(), C.c_void_p(p.data_ptr()), 0, None)
        capi.check(rc, "rela_replay_update_priority")

    def set_deferred_wait(self, on):
        """on: sample / update_priority no longer stall the caller's stream; call wait() before reading a batch."""
        capi.check(capi.lib.rela_replay_set_deferred_wait(self.h, int(bool(on))), "rela_replay_set_deferred_wait")

    def wait(self):
        capi.check(capi.lib.rela_replay_wait(self.h, _stream_ptr(self.device)), "rela_replay_wait")

    def debug_state(self):
        st = capi.ReplayState()
        capi.check(capi.lib.rela_replay_debug_state(self.h, C.byref(st), None, None, None), "rela_replay_debug_state")
        return {k: getattr(st, k) for k, _ in capi.ReplayState._fields_}


class RNNReplay:
    """RNNPrioritizedReplay(capacity, seed, alpha, beta, prefetch) over RNNTransition records
    (rela/types.h:53-73): one slot = one sequence of T = burn_in + seq_len + multi_step steps; `sample`
    returns the time-major batch RNNTransition::makeBatch builds (rela/types.cc:140-182): per-step fields
    [T, B, ...], h0 / c0 [1, B, 512], seq_len [B]."""

    FIELDS = ("s", "eps", "legal_move", "a", "reward", "terminal", "bootstrap", "h0", "c0", "seq_len")

    def __init__(self, capacity, seed, alpha, beta, prefetch, num_action, steps, device="cuda:0"):
        self.device = torch.device(device)
        self.num_action, self.T = num_action, steps
        h = C.c_void_p()
        capi.check(capi.lib.rela_replay_create(C.byref(h), capacity, seed, alpha, beta, prefetch,
                                               self.device.index or 0), "rela_replay_create")
        self.h = h
        A, T = num_action, steps
        self.row_bytes = [T * OBS_BYTES, T * 4, T * 4 * A, T * 8, T * 4, T, T * 4, 2048, 2048, 4]
        rb = (C.c_int64 * 10)(*self.row_bytes)
        st = (C.c_int32 * 10)(T, T, T, T, T, T, T, 1, 1, 1)
        capi.check(capi.lib.rela_replay_set_schema_seq(h, 10, rb, st), "rela_replay_set_schema_seq")
        self._out = {}
        self._keep = None

    close = FFReplay.close
    __del__ = FFReplay.__del__
    size = FFReplay.size
    num_add = FFReplay.num_add
    update_priority = FFReplay.update_priority
    set_deferred_wait = FFReplay.set_deferred_wait
    wait = FFReplay.wait
    debug_state = FFReplay.debug_state

    def _buffers(self, batch, slot=0):
        if (batch, slot) not in self._out:
            dev, A, T = self.device, self.num_action, self.T
            mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
            self._out[(batch, slot)] = dict(
                s=mk((T, batch, 4, 84, 84), torch.uint8), eps=mk((T, batch, 1), torch.float32),
                legal_move=mk((T, batch, A), torch.float32), a=mk((T, batch), torch.int64),
                reward=mk((T, batch), torch.float32), terminal=mk((T, batch), torch.bool),
                bootstrap=mk((T, batch), torch.float32), h0=mk((1, batch, 512), torch.float32),
                c0=mk((1, batch, 512), torch.float32), seq_len=mk((batch,), torch.float32),
                weight=mk((batch,), torch.float32))
        return self._out[(batch, slot)]

    def sample(self, batchsize, device=None, gather=True, slot=0):
        b = self._buffers(batchsize, slot)
        rows = (C.c_void_p * len(self.FIELDS))(*[b[f].data_ptr() for f in self.FIELDS]) if gather else None
        capi.check(capi.lib.rela_replay_sample(self.h, batchsize, rows, C.c_void_p(b["weight"].data_ptr()),
                                               _stream_ptr(self.device)), "rela_replay_sample")
        batch = SimpleNamespace(obs={"s": b["s"], "eps": b["eps"], "legal_move": b["legal_move"]},
                                h0={"h0": b["h0"], "c0": b["c0"]}, action={"a": b["a"]}, reward=b["reward"],
                                terminal=b["terminal"], bootstrap=b["bootstrap"], seq_len=b["seq_len"])
        return batch, b["weight"]
