"""N > 1 learner path on the CPU: two gloo ranks, replicated learners, one flat gradient
all-reduce per step (rela_amd/learner.py) -- replicas must stay bit-identical and equal to a
single learner fed the concatenated batch."""
import os

import pytest
import socket

import numpy as np
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
    """IS weights over two replay partitions with deliberately UNEQUAL sums (123.5 vs 77.25) and sizes: every
    partition contributes B / G draws, so item i of partition g is drawn with probability raw_i / (G * sum_g) and the
    correction is (N_total * that) ** -beta over the global maximum -- the single-buffer formula of
    prioritized_replay.h:320-322 applied to the probabilities the partitioned sampler really draws with.  With equal
    partition sums it coincides with the reference's formula on the union (second half of the test)."""
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
    sums, total_size = torch.tensor([123.5, 77.25]), torch.tensor(1640.0)
    prob = raw / (2.0 * sums[:, None])                # P(draw item i) under B / G draws per partition
    ref = (total_size * prob).pow(-0.4).reshape(-1)
    ref = ref / ref.max()
    got = torch.tensor(res).reshape(-1)
    torch.testing.assert_close(got, ref, rtol=1e-6, atol=0)
    assert got.max() == 1.0
    # the expectation of w_i**(-1/beta) * P(i) ... sanity of the probabilities themselves: they sum to one over both
    # partitions when summed over ALL items (here: per partition sum_g / (G * sum_g) = 1 / G)
    # equal sums: identical to the reference's single-buffer formula on the union
    eq = (total_size * (raw / (2.0 * 100.0))).pow(-0.4)
    single = (total_size * (raw / 200.0)).pow(-0.4)
    torch.testing.assert_close(eq, single, rtol=0, atol=0)


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


# ---- C3 / C4 layout: learner rank + actor ranks with replay partitions (rela_amd/parallel.py) ----------
_EX = dict(cap=64, batch=16, beta=0.4, alpha=1.0, rounds=3, blocks=6, block=12)


def _feed(oracle, seed):
    """the insertion stream of one partition: `blocks` blocks of `block` tagged rows"""
    import numpy as np

    rng = np.random.default_rng(seed)
    for b in range(_EX["blocks"]):
        tags = np.arange(b * _EX["block"], (b + 1) * _EX["block"]) + 1000 * seed
        assert oracle.add(tags, rng.uniform(0.05, 2.0, _EX["block"]).astype(np.float32)) == 0


class _OraclePartition:
    """A replay partition for the CPU exchange tests, built on the oracle (test infrastructure): rows are the
    int64 tag and a float payload derived from it."""

    def __init__(self, seed):
        import sys

        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from oracle_lib import OracleReplay

        self.o = OracleReplay(_EX["cap"], seed, _EX["alpha"], _EX["beta"])
        _feed(self.o, seed)

    def sample(self, n):
        import numpy as np

        st = self.o.state()
        rc, ids, tags, w = self.o.sample(n)
        assert rc == 0
        fields = {"tag": torch.from_numpy(tags.copy()),
                  "payload": torch.from_numpy(np.stack([tags * 0.5, tags * 0.25, tags + 1.0], 1).astype(np.float32)),
                  # a TIME-MAJOR field [T, n] as RNNTransition.makeBatch builds them (rela/types.cc:140-182)
                  "seq": torch.from_numpy(np.stack([tags + 10.0 * t for t in range(4)], 0).astype(np.float32))}
        return fields, torch.from_numpy(self.o.last_raw_w(n)), float(np.float32(st["sum"])), st["size"]

    def update_priority(self, p):
        assert self.o.update(p.numpy()) == 0


def _exchange_worker(rank, world, port, out, scheduled=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rela_amd.learner import ffnet_flat_layout
    from rela_amd.parallel import FieldSpec, PartitionedReplay, PartitionServer

    specs = [FieldSpec("tag", (), torch.int64), FieldSpec("payload", (3,), torch.float32),
             FieldSpec("seq", (4,), torch.float32, batch_dim=1)]
    _, total = ffnet_flat_layout(18)  # the flat buffers HipApexLearner publishes (online, target)
    if rank == 0:
        rep = PartitionedReplay(specs, _EX["batch"], _EX["beta"], "cpu", scheduled=scheduled)
        rounds = []
        if scheduled:  # first cycle: two sample / update pairs, announced with the first publish
            rep.publish(torch.zeros(total), torch.zeros(total), steps=2)
        for r in range(_EX["rounds"]):
            if scheduled and r % 2 == 1:  # a prefetching learner: the gather runs while it does something else
                pending = rep.sample(async_op=True)
                fields, w = pending.wait()
            else:
                fields, w = rep.sample()
            try:
                rep.sample()
                raise SystemExit("second sample without update was accepted")
            except AssertionError:
                pass
            prio = (fields["tag"] % 7).float() * 0.3 + 0.1 + r
            rounds.append(dict(tag=fields["tag"].tolist(), payload=fields["payload"].tolist(), w=w.tolist(),
                               prio=prio.tolist(), seq=fields["seq"].tolist()))
            fields = {k: v.clone() for k, v in fields.items()}  # (the buffers are reused two samples later)
            rep.update_priority(prio)
            if r == 1:
                rep.publish(torch.arange(total, dtype=torch.float32), torch.arange(total, dtype=torch.float32) * 2,
                            steps=_EX["rounds"] - 2)
        rep.stop()
        out.put(("learner", rounds))
    else:
        got = []
        srv = PartitionServer(_OraclePartition(rank), specs, _EX["batch"], _EX["beta"], "cpu", flat_sizes=(total, total),
                              on_weights=lambda on, tg: got.append((float(on[12345]), float(tg[12345]), on.numel())),
                              scheduled=scheduled)
        srv.serve_forever()
        out.put(("actor", rank, srv.partition.o.state()["sum"], srv.served, got))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scheduled", [False, True], ids=["command-words", "scheduled"])
def test_partitioned_replay_exchange_over_three_ranks(scheduled):
    """Learner rank + two actor ranks (gloo): B/G sampling per partition, ONE packed row gather per sample, IS weights
    normalised over both partitions, priority scatter and the flat weight publish -- with a command word per call and
    in scheduled mode (command words only at publish / stop, an asynchronous sample in between).  Each partition must behave exactly
    like a reference PrioritizedReplay(capacity/G, seed_g) fed the same stream and asked for B/G (SURVEY 8e's
    parity definition): the test replays both partitions locally on the oracle and compares everything."""
    import numpy as np

    from oracle_lib import OracleReplay

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 3, port, out, scheduled)) for r in range(3)]
    for p in procs:
        p.start()
    msgs = [out.get(timeout=180) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rounds = [m for m in msgs if m[0] == "learner"][0][1]
    actors = {m[1]: m for m in msgs if m[0] == "actor"}
    mirrors = {g: OracleReplay(_EX["cap"], g, _EX["alpha"], _EX["beta"]) for g in (1, 2)}
    for g, o in mirrors.items():
        _feed(o, g)
    bl = _EX["batch"] // 2
    for r, rec in enumerate(rounds):
        raw, sums, sizes, tags = [], [], [], []
        for g in (1, 2):
            st = mirrors[g].state()
            rc, ids, t, _ = mirrors[g].sample(bl)
            assert rc == 0
            raw.append(mirrors[g].last_raw_w(bl))
            sums.append(np.float32(st["sum"]))
            sizes.append(st["size"])
            tags.append(t)
        exp_tags = np.concatenate(tags)
        assert rec["tag"] == exp_tags.tolist(), "round %d: rows are not the partitions' own B/G samples" % r
        np.testing.assert_array_equal(np.array(rec["payload"], np.float32),
                                      np.stack([exp_tags * 0.5, exp_tags * 0.25, exp_tags + 1.0], 1).astype(np.float32))
        np.testing.assert_array_equal(np.array(rec["seq"], np.float32),  # gathered along the batch axis of [T, B]
                                      np.stack([exp_tags + 10.0 * t for t in range(4)], 0).astype(np.float32))
        # prioritized_replay.h:320-322 with the probabilities the partitioned sampler draws with: N is the total size,
        # item i of partition g has P(i) = raw_i / (G * sum_g) (B / G draws per partition), the maximum is global
        tot_n = float(sizes[0] + sizes[1])
        prob = np.concatenate([raw[i] / (np.float32(2.0) * sums[i]) for i in range(2)])
        w = (np.float32(tot_n) * prob) ** np.float32(-_EX["beta"])
        np.testing.assert_allclose(np.array(rec["w"], np.float32), w / w.max(), rtol=2e-6)
        assert max(rec["w"]) == 1.0
        prio = np.array(rec["prio"], np.float32)
        for i, g in enumerate((1, 2)):  # the scatter hands every partition ITS chunk of the priorities
            assert mirrors[g].update(prio[i * bl:(i + 1) * bl]) == 0
    for g in (1, 2):
        _, _, final_sum, served, got = actors[g]
        assert served == _EX["rounds"] and final_sum == mirrors[g].state()["sum"]
        from rela_amd.learner import ffnet_flat_layout

        expect = [(12345.0, 24690.0, ffnet_flat_layout(18)[1])]
        assert got == ([(0.0, 0.0, ffnet_flat_layout(18)[1])] + expect if scheduled else expect)


def test_lstmnet_flat_layout_matches_the_r2d2_learner_buffer():
    """The flat buffer the R2D2 learner publishes to actor-only ranks: rela_lstmnet_params order, 4-float padding."""
    from rela_amd.learner import HipR2D2Learner, lstmnet_flat_layout

    layout, total = lstmnet_flat_layout(18)
    assert [k for k, _, _ in layout] == list(HipR2D2Learner.KEYS)
    sizes = [int(np.prod(s)) if len(s) else 1 for _, s, _ in layout]
    assert sizes[6] == 2048 * 3136 and sizes[7] == 2048 * 512 and sizes[12] == 18 * 512
    offs = [o for _, _, o in layout]
    assert all(o % 4 == 0 for o in offs) and offs == [sum((n + 3) // 4 * 4 for n in sizes[:i]) for i in range(14)]
    assert total == sum((n + 3) // 4 * 4 for n in sizes)
