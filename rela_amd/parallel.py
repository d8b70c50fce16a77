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

A rank-0 command word keeps the collectives of all ranks in the same order (RCCL requires it); in `scheduled`
mode only the weight publish (every actor_sync_freq steps) and the final stop carry one, the sample / update pairs in
between follow the announced cycle without any host synchronisation.  A sample is ONE gather of a packed per-rank
record into a pre-allocated buffer; `sample(async_op=True)` lets it overlap the learner's backward pass.  The
exchange is backend-agnostic: a partition is any object with

    sample(n) -> (fields: dict[str, Tensor] with leading dimension n (or [T, n, ...] when time_major),
                  raw_w: Tensor[n], partition_sum: float | Tensor, partition_size: int)
    update_priority(Tensor[n])

so the CPU tests drive it over gloo with partitions built on the oracle (tests/test_dist_cpu.py) and the
GPU paths with rela_amd.replay.FFReplay / the `rela` module's replays (FFPartition below).

r4, NATIVE data plane (NativePartitionedReplay / NativePartitionServer, partitions = rela_amd.replay.FFReplay /
RNNReplay): the sampled ROWS no longer travel through a collective.  Every partition is exported once through HIP IPC
(include/rela_amd.h: rela_replay_export_ipc) and mapped into the learner process; per step the owner only SAMPLES
(ids, raw weights, eviction -- no gather), the learner's own gather kernel reads the B / G rows of every partition out
of the owner's HBM (peer reads over xGMI between GPUs) straight into its batch tensors, and the weight publish maps
the learner's flat parameter buffers into the actor processes, which load their nets from them directly.  What was left
on torch.distributed in r4: rendezvous, the command words, the B / G importance weights (the gather of 4 B / G bytes per
rank doubled as "my sample has completed") and the B / G priorities back (doubled as "the learner has read your rows").

r5, control = "slots" (the default): the per-step collectives are gone as well.  A page of step counters shared by the
processes and by their GPUs' command processors (include/rela_amd.h: rela_ipc_page_*) carries the ordering -- the owner's
stream writes sampled[g] = k behind its sample, the learner waits for that word (a host wait at the moment it needs the
batch; nothing is parked in a stream), gathers rows AND raw weights, computes
the importance weights over all partitions itself (the partitions' sizes ride in the page), writes the priorities into
a buffer the owners have mapped and writes consumed[g] = k; the owner's serving thread waits for that word on the HOST
(the release of the slots it held is host bookkeeping) and updates from the mapped buffer.  No GPU synchronisation, no
collective, no command word per step; torch.distributed keeps rendezvous and the publish / stop words.
"""
import os

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


def _pad16(n):
    return (n + 15) // 16 * 16


class _Exchange:
    """State shared by both sides: ranks, groups, specs and the PACKED per-rank sample record -- every field of the
    b_local rows a partition contributes, then its b_local IS weights, each at a 16-byte aligned offset of ONE
    contiguous uint8 buffer, so that a sample is ONE gather into a pre-allocated buffer (r3; it was one gather per
    field, each allocating `world` receive buffers)."""

    def __init__(self, specs, batch, beta, device, learner_rank=0, group=None, scheduled=False):
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.learner_rank = learner_rank
        self.group = group
        self.actor_ranks = [r for r in range(self.world) if r != learner_rank]
        self.G = len(self.actor_ranks)
        assert self.G >= 1 and batch % self.G == 0, "the learner batch (%d) must split over %d partitions" % (batch, self.G)
        self.batch, self.b_local, self.beta = batch, batch // self.G, beta
        self.specs, self.device = specs, torch.device(device)
        self.scheduled = scheduled
        # every rank creates the actor-only subgroup (new_group is collective over the default group)
        self.actor_group = dist.new_group(ranks=self.actor_ranks)
        self._cmd = torch.zeros(2, dtype=torch.int64, device=self.device)
        self._layout = []  # (spec, byte offset, bytes, local shape)
        off = 0
        for sp in specs:
            shape = list(sp.shape)
            shape.insert(sp.batch_dim, self.b_local)
            nb = int(torch.empty(shape, dtype=sp.dtype, device="meta").numel()) * torch.empty((), dtype=sp.dtype).element_size()
            self._layout.append((sp, off, nb, tuple(shape)))
            off += _pad16(nb)
        self._w_off = off
        self.rank_bytes = _pad16(off + 4 * self.b_local)

    def _bcast_cmd(self, code=0, arg=0):
        """the one host-synchronising exchange: in scheduled mode only publish / stop use it"""
        if self.rank == self.learner_rank:
            self._cmd[0], self._cmd[1] = code, arg
        dist.broadcast(self._cmd, src=self.learner_rank, group=self.group)
        return int(self._cmd[0]), int(self._cmd[1])


class _PendingSample:
    def __init__(self, owner, slot, work):
        self.owner, self.slot, self.work = owner, slot, work

    def wait(self):
        """-> (fields, IS weights) once the packed gather has landed"""
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.owner._unpack(self.slot)


class PartitionedReplay(_Exchange):
    """Learner side: one logical replay over the actor ranks' partitions, with the reference's surface
    (sample / update_priority, one outstanding batch, rela/prioritized_replay.h:202-245).

    scheduled = False: every call is announced by a rank-0 command word (a host synchronisation on every rank).
    scheduled = True (the steady state of a training run): only `publish(..., steps=n)` and `stop()` carry a command
    word; the n sample / update_priority pairs that follow a publish need none -- the actor ranks know the cycle."""

    def __init__(self, specs, batch, beta, device, learner_rank=0, group=None, scheduled=False):
        super().__init__(specs, batch, beta, device, learner_rank, group, scheduled)
        self._outstanding = False
        self._left = 0  # scheduled mode: sample / update pairs left in the announced cycle
        # two receive slots (a prefetched sample lands while the previous batch is still being read), allocated once
        self._recv = [torch.empty(self.world * self.rank_bytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self._out = [{sp.name: sp.empty(self.batch, self.device) for sp in specs} for _ in range(2)]
        self._w = [torch.empty(self.batch, dtype=torch.float32, device=self.device) for _ in range(2)]
        self._slot = 0
        self._prio = torch.zeros(self.world * self.b_local, dtype=torch.float32, device=self.device)
        self._prio_recv = torch.empty(self.b_local, dtype=torch.float32, device=self.device)
        ar = self.actor_ranks
        self._contig = ar == list(range(ar[0], ar[0] + self.G))

    def _unpack(self, slot):
        rows = self._recv[slot].view(self.world, self.rank_bytes)
        rows = rows[self.actor_ranks[0]:self.actor_ranks[0] + self.G] if self._contig else rows[self.actor_ranks]
        out = self._out[slot]
        for sp, off, nb, shape in self._layout:
            src = rows[:, off:off + nb].view(sp.dtype).reshape((self.G,) + shape)  # [G, *local shape]
            dst = out[sp.name]
            dshape = list(dst.shape)
            dshape[sp.batch_dim:sp.batch_dim + 1] = [self.G, self.b_local]
            perm = list(range(1, sp.batch_dim + 1)) + [0] + list(range(sp.batch_dim + 1, len(shape) + 1))
            dst.view(dshape).copy_(src.permute(perm))  # ONE strided copy per field
        self._w[slot].view(self.G, self.b_local).copy_(rows[:, self._w_off:self._w_off + 4 * self.b_local].view(torch.float32))
        return out, self._w[slot]

    def sample(self, async_op=False):
        """-> (fields: dict of [B, ...] (or [T, B, ...]) tensors on the learner device, IS weights [B]); with
        async_op a handle whose wait() returns them (the gather then overlaps whatever the caller does meanwhile,
        e.g. the backward half of the previous step).  The returned tensors are valid until the sample after next."""
        assert not self._outstanding, "Error: previous samples' priority has not been updated."  # :203-206
        if self.scheduled:
            assert self._left > 0, "scheduled mode: publish(..., steps=n) announces the next n sample / update pairs"
        else:
            self._bcast_cmd(CMD_SAMPLE)
        slot = self._slot
        self._slot ^= 1
        recv = self._recv[slot]
        views = [recv[r * self.rank_bytes:(r + 1) * self.rank_bytes] for r in range(self.world)]
        work = dist.gather(views[self.rank], views, dst=self.learner_rank, group=self.group, async_op=async_op)
        self._outstanding = True
        pending = _PendingSample(self, slot, work if async_op else None)
        return pending if async_op else pending.wait()

    def update_priority(self, priority):
        assert self._outstanding and priority.numel() == self.batch
        if self.scheduled:
            self._left -= 1
        else:
            self._bcast_cmd(CMD_UPDATE)
        p = priority.detach().to(self.device, torch.float32).reshape(self.G, self.b_local)
        full = self._prio.view(self.world, self.b_local)
        if self._contig:
            full[self.actor_ranks[0]:self.actor_ranks[0] + self.G].copy_(p)
        else:
            full[self.actor_ranks] = p
        dist.scatter(self._prio_recv, list(full.unbind(0)), src=self.learner_rank, group=self.group)
        self._outstanding = False

    def publish(self, *flats, steps=0):
        """ModelLocker.update_model across processes: one broadcast per flat parameter buffer (online, target; 6.8 MB
        each for AtariFFNet, 30 MB for AtariLSTMNet).  scheduled mode: `steps` = the sample / update_priority pairs that
        follow before the next publish or stop."""
        assert not self.scheduled or self._left == 0, "scheduled mode: %d announced steps are still to run" % self._left
        self._bcast_cmd(CMD_PUBLISH, len(flats) + 1000 * int(steps))
        self._left = int(steps)
        for f in flats:
            dist.broadcast(f, src=self.learner_rank, group=self.group)

    def stop(self):
        assert not self.scheduled or self._left == 0, "scheduled mode: stop() in the middle of an announced cycle"
        self._bcast_cmd(CMD_STOP)


class PartitionServer(_Exchange):
    """Actor-rank side: serves the learner's commands against this rank's replay partition."""

    def __init__(self, partition, specs, batch, beta, device, flat_sizes=(), on_weights=None, learner_rank=0, group=None,
                 scheduled=False):
        super().__init__(specs, batch, beta, device, learner_rank, group, scheduled)
        self.partition = partition
        self.on_weights = on_weights
        self._flats = [torch.empty(n, dtype=torch.float32, device=self.device) for n in flat_sizes]
        self._send = torch.zeros(self.rank_bytes, dtype=torch.uint8, device=self.device)
        self._prio_recv = torch.empty(self.b_local, dtype=torch.float32, device=self.device)
        self.served = 0

    def _sample_step(self):
        fields, raw_w, part_sum, part_size = self.partition.sample(self.b_local)
        # the "priority all-reduce": total size and the global maximum over the actor ranks
        weight = global_is_weights(raw_w.to(self.device), part_sum, part_size, self.beta, group=self.actor_group)
        for sp, off, nb, shape in self._layout:
            t = fields[sp.name]
            t = t.to(self.device, sp.dtype) if (t.dtype != sp.dtype or t.device != self.device) else t
            self._send[off:off + nb].view(sp.dtype).view(shape).copy_(t.reshape(shape))
        self._send[self._w_off:self._w_off + 4 * self.b_local].view(torch.float32).copy_(weight.float())
        dist.gather(self._send, None, dst=self.learner_rank, group=self.group)
        self.served += 1

    def _update_step(self):
        dist.scatter(self._prio_recv, None, src=self.learner_rank, group=self.group)
        self.partition.update_priority(self._prio_recv)

    def _publish_step(self, arg):
        assert arg % 1000 == len(self._flats), "publish of %d buffers, %d expected" % (arg % 1000, len(self._flats))
        for f in self._flats:
            dist.broadcast(f, src=self.learner_rank, group=self.group)
        if self.on_weights is not None:
            self.on_weights(*self._flats)
        return arg // 1000

    def serve_one(self):
        """-> False after CMD_STOP.  scheduled mode: one command = a publish and the whole cycle it announces."""
        code, arg = self._bcast_cmd()
        if code == CMD_SAMPLE:
            self._sample_step()
        elif code == CMD_UPDATE:
            self._update_step()
        elif code == CMD_PUBLISH:
            steps = self._publish_step(arg)
            if self.scheduled:
                for _ in range(steps):  # no command words inside the cycle: no host synchronisation per step
                    self._sample_step()
                    self._update_step()
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


# ---- r4: native data plane over HIP IPC ---------------------------------------------------------------------------


class _FdServer:
    """Owner side of the descriptor hand-over (r5; include/rela_amd.h: rela_replay_export_chunks).  The chunks of a
    partition's large fields are POSIX file descriptors, and a descriptor only crosses processes as SCM_RIGHTS ancillary
    data of a Unix socket -- not through torch.distributed.  This serves `fds` ONCE to `clients` connections on an
    abstract-namespace socket (no file to clean up; `name` goes into the rendezvous object) from a daemon thread, then
    closes its copies."""

    _count = 0

    def __init__(self, fds, clients=1, timeout=300.0):
        import socket
        import threading

        _FdServer._count += 1
        self.name = "rela-amd-fds-%d-%d" % (os.getpid(), _FdServer._count)
        self.fds = list(fds)
        self._sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        self._sock.bind("\0" + self.name)
        self._sock.listen(max(1, clients))
        self._sock.settimeout(timeout)
        self.error = None
        self._thread = threading.Thread(target=self._serve, args=(clients,), daemon=True)
        self._thread.start()

    def _serve(self, clients):
        import socket

        try:
            for _ in range(clients):
                conn, _ = self._sock.accept()
                with conn:
                    socket.send_fds(conn, [len(self.fds).to_bytes(4, "little")], self.fds)
                    conn.recv(1)  # the importer has the descriptors (its copies) before ours are closed
        except OSError as e:  # (a learner that never came: the partition keeps working locally)
            self.error = e
        finally:
            self._sock.close()
            for fd in self.fds:
                os.close(fd)
            self.fds = []

    def join(self, timeout=None):
        self._thread.join(timeout)


def _fetch_fds(name, timeout=300.0):
    import socket

    with socket.socket(socket.AF_UNIX, socket.SOCK_STREAM) as sk:
        sk.settimeout(timeout)
        sk.connect("\0" + name)
        msg, fds, _, _ = socket.recv_fds(sk, 4, 253)
        sk.send(b"k")
    assert len(msg) == 4 and int.from_bytes(msg, "little") == len(fds), "descriptor hand-over: %r, %d fds" % (msg, len(fds))
    return fds


def _export_desc(replay_handle):
    """-> (descriptor bytes, [fd of every chunk]) of a rela_replay*: rela_replay_export_chunks serves partitions with and
    without chunked fields (none: no descriptors, the fields travel as IPC handles inside the bytes)"""
    import ctypes as C

    from . import _capi as capi

    desc = capi.ReplayChunkDesc()
    fds = (C.c_int * capi.IPC_MAX_FDS)()
    capi.check(capi.lib.rela_replay_export_chunks(replay_handle, C.byref(desc), fds, capi.IPC_MAX_FDS),
               "rela_replay_export_chunks")
    return bytes(desc), [fds[i] for i in range(desc.nfds)]


def _import_partition(capi, C, entry, dev_index):
    """learner side: the rendezvous entry of one actor rank -> rela_replay_remote*"""
    raw = entry["partition"]
    assert len(raw) == C.sizeof(capi.ReplayChunkDesc), "partition descriptor of %d bytes" % len(raw)
    buf = capi.ReplayChunkDesc.from_buffer_copy(raw)
    fds = _fetch_fds(entry["fd_socket"]) if entry.get("fd_socket") else []
    rr = C.c_void_p()
    try:
        arr = (C.c_int * max(1, len(fds)))(*fds)
        capi.check(capi.lib.rela_replay_import_chunks(C.byref(rr), C.byref(buf), arr, len(fds), dev_index), "rela_replay_import_chunks")
    finally:
        for fd in fds:  # the mapping holds its own references
            os.close(fd)
    return rr


# word indices in the shared control page (include/rela_amd.h: rela_ipc_page_*), one 64-byte line per partition and use
_W_SAMPLED, _W_CONSUMED, _W_SIZE, _W_TEST = 0, 256, 512, 768


class _ControlPage:
    """ctypes handle on a rela_ipc_page: step counters written / waited for by STREAMS of any process of the host"""

    def __init__(self, device, name=None):
        import ctypes as C

        from . import _capi as capi
        from .engine import dev_view

        self._C, self._capi, self.device = C, capi, torch.device(device)
        self.h = C.c_void_p()
        if name is None:
            buf = C.create_string_buffer(64)
            capi.check(capi.lib.rela_ipc_page_create(C.byref(self.h), buf, self.device.index or 0), "rela_ipc_page_create")
            self.name = buf.value.decode()
        else:
            capi.check(capi.lib.rela_ipc_page_open(C.byref(self.h), name.encode(), self.device.index or 0), "rela_ipc_page_open")
            self.name = name
        # the page as kernels of this process see it (sizes are read by the learner's weight computation)
        self.words = dev_view(capi.lib.rela_ipc_page_dev_ptr(self.h), (1024,), torch.int32, self.device)

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def write(self, word, value):
        self._capi.check(self._capi.lib.rela_ipc_page_write32(self.h, word, value, self._stream()), "rela_ipc_page_write32")

    def wait(self, word, value):
        self._capi.check(self._capi.lib.rela_ipc_page_wait32(self.h, word, value, self._stream()), "rela_ipc_page_wait32")

    def host_wait(self, word, value, timeout=600.0):
        self._capi.check(self._capi.lib.rela_ipc_page_host_wait32(self.h, word, value, timeout), "rela_ipc_page_host_wait32")

    def selftest(self, word):
        ok = self._C.c_int(0)
        self._capi.check(self._capi.lib.rela_ipc_page_selftest(self.h, word, 1, self._C.byref(ok)), "rela_ipc_page_selftest")
        return bool(ok.value)

    def unlink(self):
        self._capi.lib.rela_ipc_page_unlink(self.h)

    def close(self):
        if self.h:
            self.words = None
            self._capi.lib.rela_ipc_page_close(self.h)
            self.h = None


def _agree_on_slots(page, rank, group):
    """every rank tests the stream operations on the page; the slot protocol is used only if ALL can"""
    ok = [None] * dist.get_world_size(group)
    dist.all_gather_object(ok, bool(page is not None and page.selftest(_W_TEST + 16 * rank)), group=group)
    return all(ok)


class NativePartitionedReplay(PartitionedReplay):
    """Learner side of the native exchange: as PartitionedReplay, but a sample gathers the rows itself out of the
    partitions' memory (rela_replay_remote_gather); only the importance weights arrive through the collective."""

    def __init__(self, specs, batch, beta, device, learner_rank=0, group=None, scheduled=False, flats=(), data_device=None,
                 control="slots"):
        """device: where the few control tensors of the collectives live (the GPU under RCCL, "cpu" under gloo);
        data_device: the learner's GPU, where the batch tensors live and the gather kernels run (default: device);
        control: "slots" (r5) -- per step NOTHING goes through torch.distributed: the owners' streams publish "sampled"
        step counters and their partition sizes in a shared page, this rank waits for those words, gathers rows and raw
        weights itself, computes the importance weights over all partitions, writes the priorities into a buffer the
        owners have mapped and publishes "consumed" counters (include/rela_amd.h: rela_ipc_page_*) -- or "collective"
        (r4: a gather of the weights and a scatter of the priorities per step).  Falls back to "collective" when the
        stream operations fail any rank's self-test."""
        import ctypes as C

        from . import _capi as capi
        from .engine import dev_view

        super().__init__(specs, batch, beta, device, learner_rank, group, scheduled)
        self._C, self._capi = C, capi
        self.data_device = torch.device(data_device if data_device is not None else device)
        assert self.data_device.type == "cuda", "the native exchange gathers on a GPU"
        if self.data_device != self.device:
            self._out = [{sp.name: sp.empty(self.batch, self.data_device) for sp in specs} for _ in range(2)]
        dev_index = self.data_device.index or 0
        control = os.environ.get("RELA_EXCHANGE_CONTROL", control)
        assert control in ("slots", "collective")
        # rendezvous of the descriptors: every actor rank contributes its partition's, the learner its flat buffers'
        mine = {"flats": [], "control": control}
        self._page = self._prio_ptr = None
        if control == "slots":
            self._page = _ControlPage(self.data_device)
            self._prio_ptr = C.c_void_p()  # (library memory: torch's allocator hands out pieces of blocks that do not export)
            capi.check(capi.lib.rela_ipc_alloc_buffer(C.byref(self._prio_ptr), max(4096, 4 * self.batch), dev_index), "alloc")
            h = (C.c_ubyte * 64)()
            capi.check(capi.lib.rela_ipc_export_buffer(self._prio_ptr, h), "rela_ipc_export_buffer")
            self._prio_ipc = dev_view(self._prio_ptr.value, (self.batch,), torch.float32, self.data_device)
            mine["page"], mine["prio"] = self._page.name, bytes(h)
        for f in flats:  # device buffers of the learner (rela_*_learner_flat): mapped by the actor ranks for publish
            h = (C.c_ubyte * 64)()
            capi.check(capi.lib.rela_ipc_export_buffer(C.c_void_p(f.data_ptr()), h), "rela_ipc_export_buffer")
            mine["flats"].append((bytes(h), f.numel()))
        descs = [None] * self.world
        dist.all_gather_object(descs, mine, group=group)
        self._remote = []
        for r in self.actor_ranks:  # IPC handles for the small arrays, file descriptors for the chunks of the large ones
            self._remote.append(_import_partition(capi, C, descs[r], dev_index))
        # the packed record of a rank shrinks to its importance weights
        self._layout, self._w_off = [], 0
        self.rank_bytes = _pad16(4 * self.b_local)
        self._recv = [torch.empty(self.world * self.rank_bytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.native = True
        self.slots = control == "slots" and _agree_on_slots(self._page, self.rank, group)
        if self._page is not None:
            self._page.unlink()  # every rank has opened it by now
        if self.slots:
            dd = self.data_device
            self._k = 0                       # steps sampled so far
            self._slot_step = [0, 0]
            self._raw = torch.empty((self.G, self.b_local), dtype=torch.float32, device=dd)
            self._sum = torch.empty((self.G, 1), dtype=torch.float32, device=dd)
            self._w = [torch.empty(self.batch, dtype=torch.float32, device=dd) for _ in range(2)]
            self._size_idx = torch.tensor([_W_SIZE + 16 * g for g in range(self.G)], dtype=torch.int64, device=dd)

    def sample(self, async_op=False):
        if not self.slots:
            return super().sample(async_op)
        assert not self._outstanding, "Error: previous samples' priority has not been updated."  # :203-206
        if self.scheduled:
            assert self._left > 0, "scheduled mode: publish(..., steps=n) announces the next n sample / update pairs"
        else:
            self._bcast_cmd(CMD_SAMPLE)
        slot = self._slot
        self._slot ^= 1
        self._k += 1
        self._slot_step[slot] = self._k
        self._outstanding = True
        pending = _PendingSample(self, slot, None)
        return pending if async_op else pending.wait()

    def _gather(self, slot):
        """wait -- on the HOST, at the last moment -- until every owner's stream has published `sampled` = this step, then
        queue on the caller's stream: rows and raw weights out of the partitions, importance weights over all of them
        (learner.global_is_weights' arithmetic, computed here).  The wait is not parked in a stream: a stream that
        waits for a word blocks the hardware queue it shares with other streams of the process, and a prefetched sample
        measured twice as slow that way (one-GPU rehearsal: 20.7 ms per step against 10.5)."""
        C, capi, k = self._C, self._capi, self._slot_step[slot]
        out = self._out[slot]
        for g in range(self.G):
            self._page.host_wait(_W_SAMPLED + 16 * g, k)
        with torch.cuda.device(self.data_device):
            stream = C.c_void_p(torch.cuda.current_stream(self.data_device).cuda_stream)
            rows = (C.c_void_p * len(self.specs))(*[out[sp.name].data_ptr() for sp in self.specs])
            for g, rr in enumerate(self._remote):
                capi.check(capi.lib.rela_replay_remote_gather(
                    rr, self.b_local, rows, C.c_void_p(self._raw[g].data_ptr()), C.c_void_p(self._sum[g].data_ptr()), self.batch,
                    g * self.b_local, stream), "rela_replay_remote_gather")
            total = self._page.words[self._size_idx].to(torch.float64).sum()
            w = (total.float() * (self._raw / (float(self.G) * self._sum))).pow(-self.beta)
            torch.div(w, w.max(), out=self._w[slot].view(self.G, self.b_local))

    def update_priority(self, priority):
        if not self.slots:
            return super().update_priority(priority)
        assert self._outstanding and priority.numel() == self.batch
        if self.scheduled:
            self._left -= 1
        else:
            self._bcast_cmd(CMD_UPDATE)
        self._prio_ipc.copy_(priority.detach().reshape(-1))  # the owners read their slices out of this buffer
        for g in range(self.G):
            self._page.write(_W_CONSUMED + 16 * g, self._k)  # ... once their host threads have seen this counter
        self._outstanding = False

    def _unpack(self, slot):
        C, capi = self._C, self._capi
        out = self._out[slot]
        if self.slots:
            self._gather(slot)
            return out, self._w[slot]
        stream = C.c_void_p(torch.cuda.current_stream(self.data_device).cuda_stream)
        rows = (C.c_void_p * len(self.specs))(*[out[sp.name].data_ptr() for sp in self.specs])
        for g, rr in enumerate(self._remote):  # peer reads of B / G rows per partition, straight into the batch
            capi.check(capi.lib.rela_replay_remote_gather(rr, self.b_local, rows, None, None, self.batch, g * self.b_local,
                                                          stream), "rela_replay_remote_gather")
        w = self._recv[slot].view(self.world, self.rank_bytes)
        w = w[self.actor_ranks[0]:self.actor_ranks[0] + self.G] if self._contig else w[self.actor_ranks]
        self._w[slot].view(self.G, self.b_local).copy_(w[:, :4 * self.b_local].view(torch.float32))
        return out, (self._w[slot] if self.data_device == self.device else self._w[slot].to(self.data_device))

    def publish(self, *flats, steps=0):
        """the actor ranks read the exported flat buffers themselves: the command word is all that travels"""
        assert not self.scheduled or self._left == 0, "scheduled mode: %d announced steps are still to run" % self._left
        torch.cuda.synchronize(self.data_device)  # the buffers hold the weights to publish before the actors are told
        self._bcast_cmd(CMD_PUBLISH, len(flats) + 1000 * int(steps))
        self._left = int(steps)
        # (the actors signal completion of their reads with the first collective of the cycle; a learner that
        # overwrites its parameters before that -- its next apply() -- waits for the barrier below)
        dist.barrier(group=self.group)

    def close(self):
        for rr in self._remote:
            self._capi.lib.rela_replay_remote_close(rr)
        self._remote = []
        if self._page is not None:
            torch.cuda.synchronize(self.data_device)
            self._page.close()
            self._capi.lib.rela_ipc_free_buffer(self._prio_ptr, self.data_device.index or 0)
            self._page = None


class _CapiNativePartition:
    """rela_amd.replay.FFReplay / RNNReplay (a rela_replay* behind ctypes) as the owner side of the native exchange"""

    def __init__(self, replay):
        import ctypes as C

        from . import _capi as capi
        from .engine import dev_view

        self.replay, self._C, self._capi, self._dev_view = replay, C, capi, dev_view
        self._scratch = None

    def export_desc(self):
        return _export_desc(self.replay.h)

    def sample_ids(self, n):
        C, capi, dd = self._C, self._capi, self.replay.device
        if self._scratch is None or self._scratch.numel() != n:
            self._scratch = torch.empty(n, dtype=torch.float32, device=dd)
        stream = C.c_void_p(torch.cuda.current_stream(dd).cuda_stream)
        capi.check(capi.lib.rela_replay_sample(self.replay.h, n, None, C.c_void_p(self._scratch.data_ptr()), stream),
                   "rela_replay_sample")
        raw_p, sum_p = C.c_void_p(), C.c_void_p()
        capi.check(capi.lib.rela_replay_last_sample_dev(self.replay.h, C.byref(raw_p), C.byref(sum_p)), "last_sample")
        return (self._dev_view(raw_p.value, (n,), torch.float32, dd), self._dev_view(sum_p.value, (1,), torch.float32, dd),
                capi.lib.rela_replay_last_sample_size(self.replay.h))

    def update_priority(self, p):
        self.replay.update_priority(p)


class _ModuleNativePartition:
    """the `rela` module's FFPrioritizedReplay / RNNPrioritizedReplay (one partition in this process) as the owner side"""

    def __init__(self, replay, device):
        self.replay, self.device = replay, device

    def export_desc(self):
        return self.replay.export_chunks()

    def sample_ids(self, n):
        return self.replay.sample_ids(n)

    def update_priority(self, p):
        self.replay.update_priority(p.to(self.device))


class NativePartitionServer(PartitionServer):
    """Actor-rank side of the native exchange: `replay` is a rela_amd.replay.FFReplay / RNNReplay, or an adapter with
    export_desc() / sample_ids(n) -> (raw weights, float sum, size) / update_priority(p) (the `rela` module's replays:
    _ModuleNativePartition).  Ordering against the actors' INSERTS (which keep running while this thread serves) is the
    library's: the slots a gather-less sample evicts stay reserved until update_priority (csrc/replay.hip:
    rela_replay::held), so nothing the learner's peer read can touch is rewritten before `_update_step`."""

    def __init__(self, replay, specs, batch, beta, device, flat_sizes=(), on_weights=None, learner_rank=0, group=None,
                 scheduled=False, data_device=None):
        import ctypes as C

        from . import _capi as capi
        from .engine import dev_view

        super().__init__(None, specs, batch, beta, device, flat_sizes=(), on_weights=on_weights, learner_rank=learner_rank,
                         group=group, scheduled=scheduled)
        self._C, self._capi, self._dev_view = C, capi, dev_view
        self.replay = replay if hasattr(replay, "sample_ids") else _CapiNativePartition(replay)
        self.data_device = torch.device(data_device if data_device is not None else device)
        descs = [None] * self.world
        desc, fds = self.replay.export_desc()
        self._fd_server = _FdServer(fds) if fds else None  # listening BEFORE the learner learns its name
        dist.all_gather_object(descs, {"partition": desc, "fd_socket": self._fd_server.name if fds else None}, group=group)
        dev_index = self.data_device.index or 0
        # the slot protocol (NativePartitionedReplay: control): the learner's control page and priority buffer
        ld = descs[learner_rank]
        self._page = self._prio_ptr = None
        self._g = self.actor_ranks.index(self.rank)
        self._k = 0
        if ld.get("control") == "slots":
            self._page = _ControlPage(self.data_device, ld["page"])
            self._prio_ptr = C.c_void_p()
            capi.check(capi.lib.rela_ipc_import_buffer((C.c_ubyte * 64).from_buffer_copy(ld["prio"]), C.byref(self._prio_ptr),
                                                       dev_index), "rela_ipc_import_buffer")
            self._prio_view = dev_view(self._prio_ptr.value, (self.batch,), torch.float32, self.data_device)[
                self._g * self.b_local:(self._g + 1) * self.b_local]
        self.slots = ld.get("control") == "slots" and _agree_on_slots(self._page, self.rank, group)
        self._flat_ptrs, self._flats = [], []
        for h, n in descs[learner_rank]["flats"]:  # map the learner's flat parameter buffers
            p = C.c_void_p()
            capi.check(capi.lib.rela_ipc_import_buffer((C.c_ubyte * 64).from_buffer_copy(h), C.byref(p), dev_index),
                       "rela_ipc_import_buffer")
            self._flat_ptrs.append(p)
            self._flats.append(dev_view(p.value, (n,), torch.float32, self.data_device))
        self._layout, self._w_off = [], 0
        self.rank_bytes = _pad16(4 * self.b_local)
        self._send = torch.zeros(self.rank_bytes, dtype=torch.uint8, device=self.device)

    def _sample_step(self):
        dd = self.data_device
        # ids, raw weights and eviction only: the rows stay where they are, the learner reads them
        raw_w, part_sum, size = self.replay.sample_ids(self.b_local)
        if self.slots:  # size and step counter follow the sample in stream order; the host does not wait for anything
            self._k += 1
            with torch.cuda.device(dd):
                self._page.write(_W_SIZE + 16 * self._g, int(size))
                self._page.write(_W_SAMPLED + 16 * self._g, self._k)
            self.served += 1
            return
        raw_w, part_sum = raw_w.to(self.device), part_sum.to(self.device)
        weight = global_is_weights(raw_w, part_sum, size, self.beta, group=self.actor_group)
        self._send[:4 * self.b_local].view(torch.float32).copy_(weight.float())
        torch.cuda.current_stream(dd).synchronize()  # the sample has COMPLETED before the learner hears of it
        dist.gather(self._send, None, dst=self.learner_rank, group=self.group)
        self.served += 1

    def _update_step(self):
        if self.slots:  # a HOST wait (the release of the held slots is host bookkeeping), but no GPU synchronisation
            self._page.host_wait(_W_CONSUMED + 16 * self._g, self._k)  # = "your rows were read, your priorities are written"
            self.replay.update_priority(self._prio_view)
            return
        dist.scatter(self._prio_recv, None, src=self.learner_rank, group=self.group)  # = "your rows were read"
        self.replay.update_priority(self._prio_recv)

    def _publish_step(self, arg):
        assert arg % 1000 == len(self._flats), "publish of %d buffers, %d expected" % (arg % 1000, len(self._flats))
        if self.on_weights is not None:
            self.on_weights(*self._flats)  # loads the nets straight from the learner's mapped buffers
            torch.cuda.synchronize(self.data_device)
        dist.barrier(group=self.group)
        return arg // 1000

    def close(self):
        for p in self._flat_ptrs:
            self._capi.lib.rela_ipc_close_buffer(p, self.data_device.index or 0)
        self._flat_ptrs = []
        if self._page is not None:
            torch.cuda.synchronize(self.data_device)
            self._prio_view = None
            self._capi.lib.rela_ipc_close_buffer(self._prio_ptr, self.data_device.index or 0)
            self._page.close()
            self._page = None


# ---- r5: gradient all-reduce over IPC-mapped buffers ---------------------------------------------------------------
class IpcAllReduce:
    """The replicated layout's gradient sum without a collective library (include/rela_amd.h: rela_ipc_allreduce_*):
    `bucket` -- a flat f32 device buffer THIS LIBRARY allocated (HipApexLearner.flat()[1]) -- is summed in place over the
    ranks of `group`, in rank order, bit-identically on every rank.  The descriptors (IPC handles of the buckets, the name
    of the shared page of step counters) cross through one all_gather_object at construction; run() is then a collective
    call without torch.distributed -- and, in mode 1 (stream value operations), without any host wait.  Same host only;
    RCCL remains the default of bench.py."""

    def __init__(self, bucket, group=None, device_flags=True):
        import ctypes as C

        from . import _capi as capi

        assert bucket.is_cuda and bucket.dtype == torch.float32 and bucket.is_contiguous()
        self._C, self._capi, self.bucket = C, capi, bucket
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = bucket.device
        self.h = C.c_void_p()
        desc = capi.IpcAllreduceDesc()
        capi.check(capi.lib.rela_ipc_allreduce_create(C.byref(self.h), self.rank, self.world, C.c_void_p(bucket.data_ptr()),
                                                      bucket.numel(), self.device.index or 0, int(bool(device_flags)),
                                                      C.byref(desc)), "rela_ipc_allreduce_create")
        raw = [None] * self.world
        dist.all_gather_object(raw, bytes(desc), group=group)
        descs = (capi.IpcAllreduceDesc * self.world)(*[capi.IpcAllreduceDesc.from_buffer_copy(b) for b in raw])
        capi.check(capi.lib.rela_ipc_allreduce_connect(self.h, descs), "rela_ipc_allreduce_connect")
        self.mode = capi.lib.rela_ipc_allreduce_mode(self.h)  # 1 = stream value operations, 0 = host synchronisation

    def run(self, stream=None):
        """bucket := sum over ranks, ordered on `stream` (default: torch's current stream of the bucket's device)"""
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._capi.check(self._capi.lib.rela_ipc_allreduce_run(self.h, self._C.c_void_p(s.cuda_stream)), "rela_ipc_allreduce_run")

    def close(self):
        if self.h:
            self._capi.lib.rela_ipc_allreduce_destroy(self.h)
            self.h = None
