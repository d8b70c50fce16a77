"""N > 1 learner path on the CPU: two gloo ranks, replicated learners, one flat gradient
all-reduce per step (rela_amd/learner.py) -- replicas must stay bit-identical and equal to a
single learner fed the concatenated batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rela_amd.learner import allreduce_grads

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1))
    opt = torch.optim.RMSprop(model.parameters(), lr=1e-2, eps=1.5e-4)
    g = torch.Generator().manual_seed(123)
    x_all, y_all = torch.randn(3, 8, 6, generator=g), torch.randn(3, 8, 1, generator=g)
    for step in range(3):
        x, y = x_all[step].chunk(world)[rank], y_all[step].chunk(world)[rank]  # batch split B/G
        ((model(x) - y) ** 2).mean().backward()
        allreduce_grads(model.parameters(), world)
        opt.step()
        opt.zero_grad()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out.put([t.tolist() for t in gathered])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_keeps_replicas_identical():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = torch.tensor(res[0]), torch.tensor(res[1])
    assert torch.equal(a, b)
    # single learner on the full batch: mean of the two half-batch gradients == full-batch gradient
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1))
    opt = torch.optim.RMSprop(model.parameters(), lr=1e-2, eps=1.5e-4)
    g = torch.Generator().manual_seed(123)
    x_all, y_all = torch.randn(3, 8, 6, generator=g), torch.randn(3, 8, 1, generator=g)
    for step in range(3):
        ((model(x_all[step]) - y_all[step]) ** 2).mean().backward()
        opt.step()
        opt.zero_grad()
    ref = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    torch.testing.assert_close(a, ref, rtol=1e-5, atol=1e-6)


def _is_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rela_amd.learner import global_is_weights

    g = torch.Generator().manual_seed(5)
    raw = torch.rand(2, 16, generator=g) + 0.05      # two partitions' sampled raw weights
    sums = torch.tensor([123.5, 77.25])
    sizes = [1000, 640]
    w = global_is_weights(raw[rank], sums[rank], sizes[rank], 0.4)
    gathered = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    if rank == 0:
        out.put([t.tolist() for t in gathered])
    dist.barrier()
    dist.destroy_process_group()


def test_two_partition_is_weight_normalisation():
    """IS weights over two replay partitions == the single-buffer formula on the union."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_is_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(5)
    raw = torch.rand(2, 16, generator=g) + 0.05
    total_sum, total_size = torch.tensor(123.5 + 77.25), torch.tensor(1640.0)
    ref = (total_size * (raw.reshape(-1) / total_sum)).pow(-0.4)  # prioritized_replay.h:320-322 on the union
    ref = ref / ref.max()
    got = torch.tensor(res).reshape(-1)
    torch.testing.assert_close(got, ref, rtol=1e-6, atol=0)
    assert got.max() == 1.0


def _publish_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rela_amd.learner import broadcast_weights, ffnet_flat_layout

    layout, total = ffnet_flat_layout(18)
    flat = torch.arange(total, dtype=torch.float32) if rank == 0 else torch.zeros(total)
    broadcast_weights(flat, src=0)  # learner rank 0 -> actor rank 1
    key, shape, off = layout[6]  # linear.0.weight
    n = shape[0] * shape[1]
    if rank == 1:
        out.put((bool(torch.equal(flat, torch.arange(total, dtype=torch.float32))), key, list(shape), off,
                 flat[off:off + n].view(shape)[3, 5].item(), total))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_weight_publish_layout_and_broadcast():
    """The flat parameter buffer of csrc/learner.hip (rela_ffnet_params order, segments padded to 4
    floats) broadcast from the learner rank to an actor-only rank."""
    from rela_amd.learner import ffnet_flat_layout

    layout, total = ffnet_flat_layout(18)
    assert [k for k, _, _ in layout][:2] == ["net.0.weight", "net.0.bias"]
    assert all(off % 4 == 0 for _, _, off in layout) and total % 4 == 0
    # 1,693,875 parameters at A = 18, + padding of fc_v.bias (1 -> 4) and fc_a.bias (18 -> 20)
    assert total == 1693875 + 3 + 2
    offs = {k: off for k, _, off in layout}
    assert offs["net.2.weight"] == 8192 + 32 and offs["fc_a.weight"] == offs["fc_v.bias"] + 4
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_publish_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    same, key, shape, off, val, tot = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same and key == "linear.0.weight" and shape == [512, 3136] and tot == total
    assert val == float(off + 3 * 3136 + 5)
