"""The reference's multi-GPU layout, one process per GPU (BASELINE configs C3 / C4).

pyrela/main.py:131-166 gives every act device its own ModelLocker and spreads the actor threads
round-robin over them, next to ONE learner on train_device; data flows through one host-RAM replay
and weights through load_state_dict.  Here every actor GPU is its own process with its own
device-resident replay PARTITION (the ring fed by its own actors), and the learner process sees one
logical replay:

    learner rank 0                            actor ranks 1..G
    ------------------------------------      ------------------------------------------------
    PartitionedReplay.sample(B)         -->   partition.sample(B / G)          (bit-identical to a
        gather of the sampled rows      <--       rows, raw weights             PrioritizedReplay(capacity / G,
        (RCCL gather over xGMI; gloo in           IS weights normalised over     seed_g) fed the same stream)
        the CPU tests)                            ALL partitions: SUM of (sum, size), MAX of the maximum
    learner step
    PartitionedReplay.update_priority   -->   partition.update_priority(B / G priorities)   (scatter)
    PartitionedReplay.publish(flat)     -->   one broadcast of the flat parameter buffer(s); the actor rank
                                              loads them into its nets (ModelLocker.update_model)

A rank-0 command word keeps the collectives of all ranks in the same order (RCCL requires it).  The
exchange is backend-agnostic: a partition is any object with

    sample(n) -> (fields: dict[str, Tensor] with leading dimension n (or [T, n, ...] when time_major),
                  raw_w: Tensor[n], partition_sum: float | Tensor, partition_size: int)
    update_priority(Tensor[n])

so the CPU tests drive it over gloo with partitions built on the oracle (tests/test_dist_cpu.py) and the
GPU paths with rela_amd.replay.FFReplay / the `rela` module's replays (FFPartition below).
"""
import torch
import torch.distributed as dist

from .learner import global_is_weights

CMD_SAMPLE, CMD_UPDATE, CMD_PUBLISH, CMD_STOP = 1, 2, 3, 4


class FieldSpec:
    """Shape (without the batch dimension), dtype and batch axis of one transition field."""

    def __init__(self, name, shape, dtype, batch_dim=0):
        self.name, self.shape, self.dtype, self.batch_dim = name, tuple(shape), dtype, batch_dim

    def empty(self, n, device):
        shape = list(self.shape)
        shape.insert(self.batch_dim, n)
        return torch.empty(shape, dtype=self.dtype, device=device)


def ff_field_specs(num_action):
    """FFTransition (rela/types.h:18-51) as the ten SoA fields of rela_amd.replay.FFReplay."""
    return [FieldSpec("s", (4, 84, 84), torch.uint8), FieldSpec("next_s", (4, 84, 84), torch.uint8),
            FieldSpec("eps", (1,), torch.float32), FieldSpec("next_eps", (1,), torch.float32),
            FieldSpec("legal_move", (num_action,), torch.float32), FieldSpec("next_legal_move", (num_action,), torch.float32),
            FieldSpec("a", (), torch.int64), FieldSpec("reward", (), torch.float32), FieldSpec("terminal", (), torch.uint8),
            FieldSpec("bootstrap", (), torch.float32)]


def rnn_field_specs(num_action, steps):
    """RNNTransition (rela/types.h:53-73), time-major as makeBatch builds it (rela/types.cc:140-182)."""
    T = steps
    return [FieldSpec("s", (T, 4, 84, 84), torch.uint8, 1), FieldSpec("eps", (T, 1), torch.float32, 1),
            FieldSpec("legal_move", (T, num_action), torch.float32, 1), FieldSpec("a", (T,), torch.int64, 1),
            FieldSpec("reward", (T,), torch.float32, 1), FieldSpec("terminal", (T,), torch.uint8, 1),
            FieldSpec("bootstrap", (T,), torch.float32, 1), FieldSpec("h0", (1, 512), torch.float32, 1),
            FieldSpec("c0", (1, 512), torch.float32, 1), FieldSpec("seq_len", (), torch.float32)]


class _Exchange:
    """State shared by both sides: ranks, groups, specs."""

    def __init__(self, specs, batch, beta, device, learner_rank=0, group=None):
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.learner_rank = learner_rank
        self.group = group
        self.actor_ranks = [r for r in range(self.world) if r != learner_rank]
        self.G = len(self.actor_ranks)
        assert self.G >= 1 and batch % self.G == 0, "the learner batch (%d) must split over %d partitions" % (batch, self.G)
        self.batch, self.b_local, self.beta = batch, batch // self.G, beta
        self.specs, self.device = specs, torch.device(device)
        # every rank creates the actor-only subgroup (new_group is collective over the default group)
        self.actor_group = dist.new_group(ranks=self.actor_ranks)
        self._cmd = torch.zeros(2, dtype=torch.int64, device=self.device)

    def _bcast_cmd(self, code=0, arg=0):
        if self.rank == self.learner_rank:
            self._cmd[0], self._cmd[1] = code, arg
        dist.broadcast(self._cmd, src=self.learner_rank, group=self.group)
        return int(self._cmd[0]), int(self._cmd[1])


class PartitionedReplay(_Exchange):
    """Learner side: one logical replay over the actor ranks' partitions, with the reference's surface
    (sample / update_priority, one outstanding batch, rela/prioritized_replay.h:202-245)."""

    def __init__(self, specs, batch, beta, device, learner_rank=0, group=None):
        super().__init__(specs, batch, beta, device, learner_rank, group)
        self._outstanding = False

    def _gather(self, local):
        """rows of every rank's `local` (this rank contributes a dummy) -> list ordered by actor rank"""
        bufs = [torch.empty_like(local) for _ in range(self.world)]
        dist.gather(local, bufs, dst=self.learner_rank, group=self.group)
        return [bufs[r] for r in self.actor_ranks]

    def sample(self):
        """-> (fields: dict of [B, ...] (or [T, B, ...]) tensors on the learner device, IS weights [B])"""
        assert not self._outstanding, "Error: previous samples' priority has not been updated."  # :203-206
        self._bcast_cmd(CMD_SAMPLE)
        out = {}
        for sp in self.specs:
            parts = self._gather(sp.empty(self.b_local, self.device))
            out[sp.name] = torch.cat(parts, dim=sp.batch_dim)
        weight = torch.cat(self._gather(torch.empty(self.b_local, dtype=torch.float32, device=self.device)))
        self._outstanding = True
        return out, weight

    def update_priority(self, priority):
        assert self._outstanding and priority.numel() == self.batch
        self._bcast_cmd(CMD_UPDATE)
        p = priority.detach().to(self.device, torch.float32).reshape(self.batch)
        chunks = list(p.split(self.b_local))
        scatter = []
        it = iter(chunks)
        for r in range(self.world):
            scatter.append(torch.zeros(self.b_local, device=self.device) if r == self.learner_rank else next(it).contiguous())
        recv = torch.empty(self.b_local, dtype=torch.float32, device=self.device)
        dist.scatter(recv, scatter, src=self.learner_rank, group=self.group)
        self._outstanding = False

    def publish(self, *flats):
        """ModelLocker.update_model across processes: one broadcast per flat parameter buffer
        (online, target; 6.8 MB each for AtariFFNet, 30 MB for AtariLSTMNet)."""
        self._bcast_cmd(CMD_PUBLISH, len(flats))
        for f in flats:
            dist.broadcast(f, src=self.learner_rank, group=self.group)

    def stop(self):
        self._bcast_cmd(CMD_STOP)


class PartitionServer(_Exchange):
    """Actor-rank side: serves the learner's commands against this rank's replay partition."""

    def __init__(self, partition, specs, batch, beta, device, flat_sizes=(), on_weights=None, learner_rank=0, group=None):
        super().__init__(specs, batch, beta, device, learner_rank, group)
        self.partition = partition
        self.on_weights = on_weights
        self._flats = [torch.empty(n, dtype=torch.float32, device=self.device) for n in flat_sizes]
        self.served = 0

    def _gather(self, local):
        dist.gather(local.contiguous(), None, dst=self.learner_rank, group=self.group)

    def serve_one(self):
        """-> False after CMD_STOP"""
        code, arg = self._bcast_cmd()
        if code == CMD_SAMPLE:
            fields, raw_w, part_sum, part_size = self.partition.sample(self.b_local)
            # the "priority all-reduce": totals of (sum, size) and the global maximum over the actor ranks
            weight = global_is_weights(raw_w.to(self.device), part_sum, part_size, self.beta, group=self.actor_group)
            for sp in self.specs:
                t = fields[sp.name]
                t = t.to(self.device, sp.dtype) if (t.dtype != sp.dtype or t.device != self.device) else t
                self._gather(t.reshape(sp.empty(self.b_local, "meta").shape))
            self._gather(weight.float())
            self.served += 1
        elif code == CMD_UPDATE:
            recv = torch.empty(self.b_local, dtype=torch.float32, device=self.device)
            dist.scatter(recv, None, src=self.learner_rank, group=self.group)
            self.partition.update_priority(recv)
        elif code == CMD_PUBLISH:
            assert arg == len(self._flats), "publish of %d buffers, %d expected" % (arg, len(self._flats))
            for f in self._flats:
                dist.broadcast(f, src=self.learner_rank, group=self.group)
            if self.on_weights is not None:
                self.on_weights(*self._flats)
        elif code == CMD_STOP:
            return False
        return True

    def serve_forever(self):
        while self.serve_one():
            pass


class FFPartition:
    """Adapter: rela_amd.replay.FFReplay (one device-resident partition) -> the partition protocol."""

    def __init__(self, replay):
        import ctypes as C

        from . import _capi as capi
        from .engine import dev_view

        self.replay, self._C, self._capi, self._dev_view = replay, C, capi, dev_view

    def sample(self, n):
        C, capi = self._C, self._capi
        batch, _ = self.replay.sample(n)
        raw_p, sum_p = C.c_void_p(), C.c_void_p()
        capi.check(capi.lib.rela_replay_last_sample_dev(self.replay.h, C.byref(raw_p), C.byref(sum_p)), "last_sample")
        dev = self.replay.device
        raw_w = self._dev_view(raw_p.value, (n,), torch.float32, dev)
        part_sum = self._dev_view(sum_p.value, (1,), torch.float32, dev)
        fields = {"s": batch.obs["s"], "next_s": batch.next_obs["s"], "eps": batch.obs["eps"],
                  "next_eps": batch.next_obs["eps"], "legal_move": batch.obs["legal_move"],
                  "next_legal_move": batch.next_obs["legal_move"], "a": batch.action["a"], "reward": batch.reward,
                  "terminal": batch.terminal.to(torch.uint8), "bootstrap": batch.bootstrap}
        return fields, raw_w, part_sum, self._capi.lib.rela_replay_last_sample_size(self.replay.h)

    def update_priority(self, p):
        self.replay.update_priority(p)


def rnn_batch_namespace(fields):
    """dict of gathered RNN fields (time-major) -> the RNNTransition-shaped namespace the R2D2 learners consume"""
    from types import SimpleNamespace

    return SimpleNamespace(
        obs={"s": fields["s"], "eps": fields["eps"], "legal_move": fields["legal_move"]}, action={"a": fields["a"]},
        reward=fields["reward"], terminal=fields["terminal"].bool(), bootstrap=fields["bootstrap"],
        h0={"h0": fields["h0"], "c0": fields["c0"]}, seq_len=fields["seq_len"])


def ff_batch_namespace(fields):
    """dict of gathered FF fields -> the FFTransition-shaped namespace the learners consume"""
    from types import SimpleNamespace

    return SimpleNamespace(
        obs={"s": fields["s"], "eps": fields["eps"], "legal_move": fields["legal_move"]}, action={"a": fields["a"]},
        reward=fields["reward"], terminal=fields["terminal"].bool(), bootstrap=fields["bootstrap"],
        next_obs={"s": fields["next_s"], "eps": fields["next_eps"], "legal_move": fields["next_legal_move"]})
