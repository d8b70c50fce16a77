"""Learner-side helpers: data-parallel gradient exchange for replicated learners.

The reference has a single learner (pyrela/main.py:54,114); with one learner replica per GPU
 the only exchange on the learner path is a SUM all-reduce of the fp32 gradients
(1,693,875 floats = 6.8 MB for AtariFFNet, A=18) before clipping and the optimiser step.
torch.distributed's "nccl" backend is RCCL over xGMI on ROCm; the gradients are flattened into
ONE bucket so the exchange is a single latency-bound collective instead of 12 small ones.
"""
import torch
import torch.distributed as dist


def allreduce_grads(params, world_size, group=None):
    """In-place average of .grad over all ranks with one flat all-reduce."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or world_size <= 1:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world_size)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def sum_grads(learner, world_size, group=None):
    """mean of the learner's flat gradient bucket over the ranks, in place.  learner.ipc_allreduce (an
    rela_amd.parallel.IpcAllReduce over that bucket, set by the caller) replaces the collective library: peer reads of
    IPC-mapped buckets, summed in rank order (csrc/ipc_allreduce.hip); otherwise torch.distributed (RCCL)."""
    g = learner.flat()[1]
    ar = getattr(learner, "ipc_allreduce", None)
    if ar is not None:
        ar.run()
    else:
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    g.div_(world_size)


def global_is_weights(raw_w, partition_sum, partition_size, beta, group=None):
    """Importance weights of one replay PARTITION normalised over ALL partitions.

    The reference computes  w = (size * P(i)) ** -beta;  w /= w.max()  with P(i) = raw_i / sum, the probability of
    drawing item i from its single buffer (rela/prioritized_replay.h:320-322).  With one partition per actor GPU every
    partition contributes exactly B / G of the B draws, so item i of partition g is drawn with probability
    P(i) = raw_i / (G * sum_g) -- NOT raw_i / sum_total unless all partition sums are equal -- and the correction is
        w_i = (N_total * raw_i / (G * sum_g)) ** -beta,   normalised by the maximum over every rank's batch.
    One SUM all-reduce of the partition size and one MAX all-reduce of a scalar per learner step: the "priority
    all-reduce" of the north star.  raw_w: f32[B] un-normalised weights of this rank's sample; partition_sum /
    partition_size: this partition's weight sum (tensor or float) and item count.
    """
    dev = raw_w.device
    G = dist.get_world_size(group)
    total = torch.as_tensor(float(partition_size), dtype=torch.float64, device=dev).reshape(1)
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    part_sum = torch.as_tensor(partition_sum, dtype=torch.float32, device=dev).reshape(())
    w = (total[0].float() * (raw_w / (float(G) * part_sum))).pow(-beta)
    top = w.max()
    dist.all_reduce(top, op=dist.ReduceOp.MAX, group=group)
    return w / top


class HipApexLearner:
    """The Ape-X learner step in HIP (rela_apex_learner_*, csrc/learner.hip): the counterpart of
    `loss, priority = agent.loss(batch); (loss * weight).mean().backward(); clip_grad_norm_;
    optim.step()` of pyrela/main.py:226-239 for ApexAgent + AtariFFNet, without PyTorch autograd.

        learner = HipApexLearner.from_agent(agent, batch, lr=6.25e-5, eps=1.5e-4)
        batch, weight = replay.sample(B)            # rela_amd.replay.FFReplay
        loss, priority = learner.step(batch, weight)
        replay.update_priority(priority)
        learner.publish(actor_net)                  # every actor_sync_freq steps (ModelLocker.update_model)
    """

    KEYS = ("net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias",
            "linear.0.weight", "linear.0.bias", "fc_v.weight", "fc_v.bias", "fc_a.weight", "fc_a.bias")
    SHAPES = lambda A: ((32, 4, 8, 8), (32,), (64, 32, 4, 4), (64,), (64, 64, 3, 3), (64,), (512, 3136), (512,),  # noqa: E731
                        (1, 512), (1,), (A, 512), (A,))
    FIELDS = ("s", "next_s", "eps", "next_eps", "legal_move", "next_legal_move", "a", "reward", "terminal",
              "bootstrap")

    def __init__(self, num_action, max_batch, multi_step, gamma, optimizer="rmsprop", lr=6.25e-5, eps=1.5e-4,
                 grad_clip=40.0, device="cuda:0"):
        import ctypes as C

        from . import _capi as capi

        self._C, self._capi = C, capi
        self.device = torch.device(device)
        self.num_action, self.max_batch = num_action, max_batch
        h = C.c_void_p()
        capi.check(capi.lib.rela_apex_learner_create(C.byref(h), num_action, max_batch, multi_step, gamma,
                                                     {"rmsprop": 0, "adam": 1}[optimizer], lr, eps, grad_clip,
                                                     self.device.index or 0), "rela_apex_learner_create")
        self.h = h
        self._prio = torch.empty(max_batch, dtype=torch.float32, device=self.device)
        self._loss = torch.empty(1, dtype=torch.float32, device=self.device)

    @classmethod
    def from_agent(cls, agent, max_batch, **kw):
        """agent: pyrela ApexAgent (online_net / target_net AtariFFNet, multi_step, gamma)."""
        dev = next(agent.online_net.parameters()).device
        num_action = agent.online_net.fc_a.weight.shape[0]
        self = cls(num_action, max_batch, agent.multi_step, agent.gamma, device=str(dev), **kw)
        self.load_state_dicts(agent.online_net.state_dict(), agent.target_net.state_dict())
        return self

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _params(self, sd, keep):
        p = self._capi.FFNetParams()
        for (field, _), key in zip(self._capi.FFNetParams._fields_, self.KEYS):
            t = sd[key].detach().to(self.device, torch.float32).contiguous()
            keep.append(t)
            setattr(p, field, t.data_ptr())
        return p

    def load_state_dicts(self, online_sd, target_sd=None):
        C, capi = self._C, self._capi
        keep = []
        po = self._params(online_sd, keep)
        pt = self._params(target_sd, keep) if target_sd is not None else None
        capi.check(capi.lib.rela_apex_learner_load(self.h, C.byref(po), C.byref(pt) if pt is not None else None, 1,
                                                   self._stream()), "rela_apex_learner_load")
        torch.cuda.current_stream(self.device).synchronize()  # sources may be temporaries

    def sync_target_with_online(self):
        self._capi.check(self._capi.lib.rela_apex_learner_sync_target(self.h, self._stream()), "sync_target")

    def _views(self, which):
        from .engine import dev_view

        C, capi = self._C, self._capi
        p = capi.FFNetParams()
        if which == "grads":
            capi.check(capi.lib.rela_apex_learner_grads(self.h, C.byref(p)), "rela_apex_learner_grads")
        elif which == "online":
            capi.check(capi.lib.rela_apex_learner_params(self.h, C.byref(p), None), "rela_apex_learner_params")
        else:
            capi.check(capi.lib.rela_apex_learner_params(self.h, None, C.byref(p)), "rela_apex_learner_params")
        shapes = HipApexLearner.SHAPES(self.num_action)
        return {key: dev_view(getattr(p, field), shape, torch.float32, self.device)
                for (field, _), key, shape in zip(capi.FFNetParams._fields_, self.KEYS, shapes)}, p

    def state_dict(self, which="online"):
        """Zero-copy views of the flat parameter buffer as a state_dict ("online" | "target" | "grads")."""
        return self._views(which)[0]

    def flat(self):
        """(params, grads) as flat f32 views -- the all-reduce bucket of data-parallel learners."""
        from .engine import dev_view

        C, capi = self._C, self._capi
        pp, gp, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        capi.check(capi.lib.rela_apex_learner_flat(self.h, C.byref(pp), C.byref(gp), C.byref(n)), "flat")
        return (dev_view(pp.value, (n.value,), torch.float32, self.device),
                dev_view(gp.value, (n.value,), torch.float32, self.device))

    def set_precision(self, mode):
        """"f32" (default) or "bf16x2": the arithmetic of the two gradient-free forwards of td_err (the pass whose
        activations feed the backward kernels stays f32).  "f32x3": conv2 / conv3 of all three forwards on the f32-accurate
        bf16 kernels (csrc/gemm_f32emu.h, from 512 rows); fc, heads, loss, backward and optimiser in exact f32."""
        self._capi.check(self._capi.lib.rela_apex_learner_set_precision(self.h, {"f32": 0, "bf16x2": 1, "f32x3": 2}[mode]),
                         "rela_apex_learner_set_precision")

    def flat_target(self):
        """The target net's flat parameter buffer (same layout as flat()[0]) -- what a publish sends along."""
        from .engine import dev_view

        C, capi = self._C, self._capi
        p, n = capi.FFNetParams(), C.c_int64()
        capi.check(capi.lib.rela_apex_learner_params(self.h, None, C.byref(p)), "rela_apex_learner_params")
        capi.check(capi.lib.rela_apex_learner_flat(self.h, None, None, C.byref(n)), "flat")
        return dev_view(p.conv1_w, (n.value,), torch.float32, self.device)  # conv1_w sits at offset 0

    def stats(self):
        """cuda f32[2]: gradient norm before clipping, clip coefficient of the last apply()."""
        from .engine import dev_view

        return dev_view(self._capi.lib.rela_apex_learner_stats_dev(self.h), (2,), torch.float32, self.device)

    def debug_activations(self):
        """f32 views of online(obs)'s activations the last loss() left for the backward pass (channel-last):
        a1 [B,400,32], a2 [B,81,64], a3 [B,49,64], h [B,512]; their > 0 pattern is the ReLU mask of the gradients."""
        from .engine import dev_view

        C, capi = self._C, self._capi
        p = [C.c_void_p() for _ in range(4)]
        b = C.c_int()
        capi.check(capi.lib.rela_apex_learner_debug_activations(self.h, *[C.byref(x) for x in p], C.byref(b)), "debug_activations")
        shapes = [(b.value, 400, 32), (b.value, 81, 64), (b.value, 49, 64), (b.value, 512)]
        return [dev_view(x.value, sh, torch.float32, self.device) for x, sh in zip(p, shapes)]

    def backward(self, batch, weight):
        """batch: the namespace FFReplay.sample returns (or any object with obs / next_obs / action /
        reward / terminal / bootstrap of cuda tensors); weight: cuda f32[B].  -> (loss[1], priority[B])."""
        return self._forward_half(batch, weight, "rela_apex_learner_backward")

    def loss(self, batch, weight):
        """The forward half of `backward` (forwards, priorities, loss); `grad()` runs the backward pass of it.  In
        between the caller may update_priority and sample the next batch into OTHER buffers (FFReplay.sample(...,
        slot=1 - slot)): the replay's sample path then runs next to the gradient kernels."""
        return self._forward_half(batch, weight, "rela_apex_learner_loss")

    def grad(self):
        self._capi.check(self._capi.lib.rela_apex_learner_grad(self.h, self._stream()), "rela_apex_learner_grad")

    def _forward_half(self, batch, weight, entry):
        C, capi = self._C, self._capi
        B = weight.numel()
        t = {"s": batch.obs["s"], "next_s": batch.next_obs["s"], "eps": batch.obs["eps"],
             "next_eps": batch.next_obs["eps"], "legal_move": batch.obs["legal_move"],
             "next_legal_move": batch.next_obs["legal_move"], "a": batch.action["a"], "reward": batch.reward,
             "terminal": batch.terminal, "bootstrap": batch.bootstrap}
        keep = [t[f].contiguous() for f in self.FIELDS]
        w = weight.detach().float().contiguous()
        rows = (C.c_void_p * 10)(*[x.data_ptr() for x in keep])
        self._keep = (keep, w)
        capi.check(getattr(capi.lib, entry)(self.h, B, rows, C.c_void_p(w.data_ptr()),
                                            C.c_void_p(self._prio.data_ptr()),
                                            C.c_void_p(self._loss.data_ptr()), self._stream()), entry)
        return self._loss, self._prio[:B]

    def apply(self):
        self._capi.check(self._capi.lib.rela_apex_learner_apply(self.h, self._stream()), "rela_apex_learner_apply")

    def step(self, batch, weight, world_size=1, group=None):
        loss, prio = self.backward(batch, weight)
        if world_size > 1:  # replicated learners: one flat SUM all-reduce, then the mean
            sum_grads(self, world_size, group)
        self.apply()
        return loss, prio

    def publish(self, online_handle, target_handle=None):
        """ModelLocker.update_model for device nets: repack the current weights into actor-side
        FFNetHandle objects (rela_ffnet_load from device pointers, no host copy)."""
        C, capi = self._C, self._capi
        _, po = self._views("online")
        capi.check(capi.lib.rela_ffnet_load(online_handle.h, C.byref(po), 1, self._stream()), "rela_ffnet_load")
        if target_handle is not None:
            _, pt = self._views("target")
            capi.check(capi.lib.rela_ffnet_load(target_handle.h, C.byref(pt), 1, self._stream()), "rela_ffnet_load")

    def close(self):
        if getattr(self, "h", None):
            self._capi.lib.rela_apex_learner_destroy(self.h)
            self.h = None

    def __del__(self):
        capi = getattr(self, "_capi", None)
        if capi is not None and getattr(capi, "lib", None) is not None:
            self.close()


class HipR2D2Learner:
    """The R2D2 learner step in HIP (rela_r2d2_learner_*, csrc/learner_r2d2.hip): the counterpart of
    `loss, priority = agent.loss(batch); (loss * weight).mean().backward(); clip_grad_norm_; optim.step()`
    of pyrela/main.py:226-239 for R2D2Agent + AtariLSTMNet (pyrela/r2d2.py:122-206, net.py:127-163), without
    PyTorch autograd: burn-in unroll, training unroll of online and target nets, sequence TD error, BPTT, Adam.

        learner = HipR2D2Learner.from_agent(agent, batch, lr=..., eps=...)
        batch, weight = replay.sample(B, device)          # rela.RNNPrioritizedReplay
        loss, priority = learner.step(batch, weight)
        replay.update_priority(priority)
    """

    KEYS = ("net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight", "net.4.bias",
            "lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0", "fc_v.weight", "fc_v.bias",
            "fc_a.weight", "fc_a.bias")
    SHAPES = lambda A: ((32, 4, 8, 8), (32,), (64, 32, 4, 4), (64,), (64, 64, 3, 3), (64,), (2048, 3136), (2048, 512),  # noqa: E731
                        (2048,), (2048,), (1, 512), (1,), (A, 512), (A,))

    def __init__(self, num_action, max_batch, multi_step, gamma, seq_len, burn_in, eta, optimizer="adam", lr=6.25e-5,
                 eps=1.5e-4, grad_clip=40.0, device="cuda:0"):
        import ctypes as C

        from . import _capi as capi

        self._C, self._capi = C, capi
        self.device = torch.device(device)
        self.num_action, self.max_batch = num_action, max_batch
        self.seq_len, self.burn_in, self.multi_step = seq_len, burn_in, multi_step
        h = C.c_void_p()
        capi.check(capi.lib.rela_r2d2_learner_create(C.byref(h), num_action, max_batch, multi_step, gamma, seq_len,
                                                     burn_in, float(eta), {"rmsprop": 0, "adam": 1}[optimizer], lr, eps,
                                                     grad_clip, self.device.index or 0), "rela_r2d2_learner_create")
        self.h = h
        self._prio = torch.empty(max_batch, dtype=torch.float32, device=self.device)
        self._loss = torch.empty(1, dtype=torch.float32, device=self.device)
        self._loss_seq = torch.empty(max_batch, dtype=torch.float32, device=self.device)

    @classmethod
    def from_agent(cls, agent, max_batch, **kw):
        """agent: pyrela R2D2Agent (online_net / target_net AtariLSTMNet, multi_step, gamma, eta, seq_len, burn_in)."""
        dev = next(agent.online_net.parameters()).device
        num_action = agent.online_net.fc_a.weight.shape[0]
        self = cls(num_action, max_batch, agent.multi_step, agent.gamma, agent.seq_len, agent.burn_in, agent.eta,
                   device=str(dev), **kw)
        self.load_state_dicts(agent.online_net.state_dict(), agent.target_net.state_dict())
        return self

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _params(self, sd, keep):
        p = self._capi.LSTMNetParams()
        for (field, _), key in zip(self._capi.LSTMNetParams._fields_, self.KEYS):
            t = sd[key].detach().to(self.device, torch.float32).contiguous()
            keep.append(t)
            setattr(p, field, t.data_ptr())
        return p

    def load_state_dicts(self, online_sd, target_sd=None):
        C, capi = self._C, self._capi
        keep = []
        po = self._params(online_sd, keep)
        pt = self._params(target_sd, keep) if target_sd is not None else None
        capi.check(capi.lib.rela_r2d2_learner_load(self.h, C.byref(po), C.byref(pt) if pt is not None else None, 1,
                                                   self._stream()), "rela_r2d2_learner_load")
        torch.cuda.current_stream(self.device).synchronize()  # sources may be temporaries

    def sync_target_with_online(self):
        self._capi.check(self._capi.lib.rela_r2d2_learner_sync_target(self.h, self._stream()), "sync_target")

    def state_dict(self, which="online"):
        """Zero-copy views of the flat buffers as a state_dict ("online" | "target" | "grads")."""
        from .engine import dev_view

        C, capi = self._C, self._capi
        p = capi.LSTMNetParams()
        if which == "grads":
            capi.check(capi.lib.rela_r2d2_learner_grads(self.h, C.byref(p)), "rela_r2d2_learner_grads")
        elif which == "online":
            capi.check(capi.lib.rela_r2d2_learner_params(self.h, C.byref(p), None), "rela_r2d2_learner_params")
        else:
            capi.check(capi.lib.rela_r2d2_learner_params(self.h, None, C.byref(p)), "rela_r2d2_learner_params")
        shapes = HipR2D2Learner.SHAPES(self.num_action)
        return {key: dev_view(getattr(p, field), shape, torch.float32, self.device)
                for (field, _), key, shape in zip(capi.LSTMNetParams._fields_, self.KEYS, shapes)}

    def flat(self):
        """(params, grads) as flat f32 views -- the all-reduce bucket of data-parallel learners (30 MB)."""
        from .engine import dev_view

        C, capi = self._C, self._capi
        pp, gp, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        capi.check(capi.lib.rela_r2d2_learner_flat(self.h, C.byref(pp), C.byref(gp), C.byref(n)), "flat")
        return (dev_view(pp.value, (n.value,), torch.float32, self.device),
                dev_view(gp.value, (n.value,), torch.float32, self.device))

    def stats(self):
        from .engine import dev_view

        return dev_view(self._capi.lib.rela_r2d2_learner_stats_dev(self.h), (2,), torch.float32, self.device)

    def flat_target(self):
        """The target net's flat parameter buffer (same layout as flat()[0]) -- what a publish sends along."""
        from .engine import dev_view

        C, capi = self._C, self._capi
        p, n = capi.LSTMNetParams(), C.c_int64()
        capi.check(capi.lib.rela_r2d2_learner_params(self.h, None, C.byref(p)), "rela_r2d2_learner_params")
        capi.check(capi.lib.rela_r2d2_learner_flat(self.h, None, None, C.byref(n)), "flat")
        return dev_view(getattr(p, capi.LSTMNetParams._fields_[0][0]), (n.value,), torch.float32, self.device)

    def set_precision(self, mode):
        """"f32" (default); "bf16x2": the target net's conv trunk on split-bf16 MFMA (no gradient flows through it);
        "f32x3": conv2 / conv3 of both trunks on the f32-accurate three-part bf16 kernels (csrc/gemm_f32emu.h)."""
        self._capi.check(self._capi.lib.rela_r2d2_learner_set_precision(self.h, {"f32": 0, "bf16x2": 1, "f32x3": 2}[mode]),
                         "rela_r2d2_learner_set_precision")

    def check(self):
        """Synchronises and raises if a grid barrier of the persistent recurrent kernels gave up since the last check."""
        stream = self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._capi.check(self._capi.lib.rela_r2d2_learner_check(self.h, stream), "rela_r2d2_learner_check")

    def backward(self, batch, weight):
        """batch: RNNTransition-shaped (time-major [T, B, ...] cuda tensors: obs{s, eps, legal_move}, h0{h0, c0},
        action{a}, reward, terminal, bootstrap, seq_len); weight: cuda f32[B].
        -> (mean(loss * weight)[1], aggregated priority[B], per-sequence loss[B])."""
        return self._forward_half(batch, weight, "rela_r2d2_learner_backward")

    def loss(self, batch, weight):
        """The forward half of `backward` (unrolls, TD errors, priorities, loss); `grad()` runs the backward pass of
        it.  In between the caller may update_priority and sample the next batch into OTHER buffers
        (RNNReplay.sample(..., slot=1 - slot))."""
        return self._forward_half(batch, weight, "rela_r2d2_learner_loss")

    def grad(self):
        self._capi.check(self._capi.lib.rela_r2d2_learner_grad(self.h, self._stream()), "rela_r2d2_learner_grad")

    def _forward_half(self, batch, weight, entry):
        C, capi = self._C, self._capi
        B = weight.numel()
        dev = self.device
        fields = [batch.obs["s"], batch.obs["eps"].float(), batch.obs["legal_move"].float(), batch.action["a"],
                  batch.reward.float(), batch.terminal.to(torch.uint8), batch.bootstrap.float(), batch.h0["h0"].float(),
                  batch.h0["c0"].float(), batch.seq_len.float()]
        keep = [f.to(dev).contiguous() for f in fields]
        T = self.burn_in + self.seq_len + self.multi_step
        assert keep[0].dtype == torch.uint8 and keep[0].shape[:2] == (T, B), (keep[0].shape, T, B)
        assert keep[3].dtype == torch.int64
        w = weight.detach().to(dev).float().contiguous()
        rows = (C.c_void_p * 10)(*[x.data_ptr() for x in keep])
        self._keep = (keep, w)
        capi.check(getattr(capi.lib, entry)(self.h, B, rows, C.c_void_p(w.data_ptr()),
                                            C.c_void_p(self._prio.data_ptr()), C.c_void_p(self._loss.data_ptr()),
                                            C.c_void_p(self._loss_seq.data_ptr()), self._stream()), entry)
        return self._loss, self._prio[:B], self._loss_seq[:B]

    def apply(self):
        self._capi.check(self._capi.lib.rela_r2d2_learner_apply(self.h, self._stream()), "rela_r2d2_learner_apply")

    def step(self, batch, weight, world_size=1, group=None):
        loss, prio, _ = self.backward(batch, weight)
        if world_size > 1:
            sum_grads(self, world_size, group)
        self.apply()
        return loss, prio

    def publish(self, online_handle, target_handle=None):
        """ModelLocker.update_model for device nets: repack the current weights into actor-side LSTMNetHandle
        objects (rela_lstmnet_load from device pointers, no host copy)."""
        C, capi = self._C, self._capi
        po, pt = capi.LSTMNetParams(), capi.LSTMNetParams()
        capi.check(capi.lib.rela_r2d2_learner_params(self.h, C.byref(po), C.byref(pt)), "rela_r2d2_learner_params")
        capi.check(capi.lib.rela_lstmnet_load(online_handle.h, C.byref(po), 1, self._stream()), "rela_lstmnet_load")
        if target_handle is not None:
            capi.check(capi.lib.rela_lstmnet_load(target_handle.h, C.byref(pt), 1, self._stream()), "rela_lstmnet_load")

    def close(self):
        if getattr(self, "h", None):
            self._capi.lib.rela_r2d2_learner_destroy(self.h)
            self.h = None

    def __del__(self):
        capi = getattr(self, "_capi", None)
        if capi is not None and getattr(capi, "lib", None) is not None:
            self.close()


# ---- weight publish to actor-only ranks (C3 / C4 layouts) ------------------------------
FFNET_KEYS = HipApexLearner.KEYS


def ffnet_flat_layout(num_action):
    """[(state_dict key, shape, offset)] and the total length of the flat f32 parameter buffer of
    csrc/learner.hip: rela_ffnet_params order, every segment padded to a multiple of 4 floats."""
    out, off = [], 0
    for key, shape in zip(FFNET_KEYS, HipApexLearner.SHAPES(num_action)):
        n = 1
        for d in shape:
            n *= d
        out.append((key, shape, off))
        off += (n + 3) // 4 * 4
    return out, off


def lstmnet_flat_layout(num_action):
    """As ffnet_flat_layout for the flat buffer of csrc/learner_r2d2.hip (rela_lstmnet_params order)."""
    out, off = [], 0
    for key, shape in zip(HipR2D2Learner.KEYS, HipR2D2Learner.SHAPES(num_action)):
        n = 1
        for d in shape:
            n *= d
        out.append((key, shape, off))
        off += (n + 3) // 4 * 4
    return out, off


def broadcast_weights(flat, src=0, group=None):
    """ModelLocker.update_model across processes: ONE broadcast of the flat parameter buffer
    (6.8 MB for AtariFFNet) from the learner rank to the actor ranks (RCCL over xGMI on GPUs)."""
    dist.broadcast(flat, src=src, group=group)
    return flat


def load_net_from_flat(net_handle, flat, num_action):
    """Repack an actor-side FFNetHandle from a flat parameter buffer resident on its GPU."""
    import ctypes as C

    from . import _capi as capi

    layout, total = ffnet_flat_layout(num_action)
    assert flat.is_cuda and flat.dtype == torch.float32 and flat.numel() == total and flat.is_contiguous()
    p = capi.FFNetParams()
    for (field, _), (_, _, off) in zip(capi.FFNetParams._fields_, layout):
        setattr(p, field, flat.data_ptr() + 4 * off)
    stream = C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)
    capi.check(capi.lib.rela_ffnet_load(net_handle.h, C.byref(p), 1, stream), "rela_ffnet_load")
    net_handle._keep = [flat]
