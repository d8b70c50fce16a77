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
