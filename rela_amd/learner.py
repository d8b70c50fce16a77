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


def global_is_weights(raw_w, partition_sum, partition_size, beta, group=None):
    """Importance weights of one replay PARTITION normalised over ALL partitions.

    The reference computes  w = (size * raw / sum) ** -beta;  w /= w.max()  over its single buffer
    (rela/prioritized_replay.h:320-322).  With one partition per actor GPU, `size` and `sum` become the
    totals over the partitions (one SUM all-reduce of two scalars) and the maximum is taken over every
    rank's batch (one MAX all-reduce of a scalar) -- the "priority all-reduce" of the north star.
    raw_w: f32[B] un-normalised weights of this rank's sample; partition_sum / partition_size: this
    partition's weight sum (tensor or float) and item count.  Two tiny collectives per learner step.
    """
    dev = raw_w.device
    stats = torch.stack([torch.as_tensor(partition_sum, dtype=torch.float64, device=dev).reshape(()),
                         torch.as_tensor(float(partition_size), dtype=torch.float64, device=dev).reshape(())])
    dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    total_sum, total_size = stats[0].float(), stats[1].float()
    w = (total_size * (raw_w / total_sum)).pow(-beta)
    top = w.max()
    dist.all_reduce(top, op=dist.ReduceOp.MAX, group=group)
    return w / top
