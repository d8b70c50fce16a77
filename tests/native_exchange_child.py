"""Child of tests/test_ipc_gpu.py::test_native_exchange_three_processes: rank 0 = learner, ranks 1..G = actor ranks, all
on cuda:0 over gloo (RCCL refuses two ranks on one device).  Exercises rela_amd.parallel's NATIVE data plane: partitions
exported through HIP IPC (rank 2's: its large fields as chunks handed over as file descriptors), the learner's own gather kernel reading the sampled rows out of the owners' memory, the
learner's flat parameter buffers mapped by the actor ranks for the weight publish."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from rela_amd.parallel import (NativePartitionedReplay, NativePartitionServer, ff_field_specs)
from rela_amd.replay import FFReplay

A, BATCH, BETA, ROUNDS = 18, 64, 0.4, 4
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
G = world - 1
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = "cuda:0"
torch.cuda.set_device(0)
specs = ff_field_specs(A)

if rank == 0:
    from rela_amd.learner import HipApexLearner
    from rela_amd.pyrela.apex import ApexAgent
    from rela_amd.pyrela.net import AtariFFNet

    torch.manual_seed(3)
    agent = ApexAgent(lambda: AtariFFNet(A), 3, 0.997).to(dev)
    learner = HipApexLearner.from_agent(agent, BATCH, lr=1e-4, eps=1e-4, grad_clip=40.0)
    flat_on, flat_tg = learner.flat()[0], learner.flat_target()
    rep = NativePartitionedReplay(specs, BATCH, BETA, dev, flats=(flat_on, flat_tg))
    b = BATCH // G
    for r in range(ROUNDS):
        flat_on.fill_(float(r + 1))
        flat_tg.fill_(float(-(r + 1)))
        rep.publish(flat_on, flat_tg)
        if r % 2 == 0:
            fields, w = rep.sample()
        else:
            fields, w = rep.sample(async_op=True).wait()
        torch.cuda.synchronize()
        a = fields["a"].cpu().numpy()
        s0 = fields["s"][:, 0, 0, :8].cpu().numpy()       # first 8 bytes of the frame = the tag, little endian
        ns0 = fields["next_s"][:, 0, 0, :8].cpu().numpy()
        for g in range(G):
            sl = slice(g * b, (g + 1) * b)
            assert ((a[sl] // 1000000) == g + 1).all(), ("rows of partition %d at the wrong place" % g, a[sl][:8])
        assert np.array_equal(s0.view(np.int64).reshape(-1), a), "frame rows do not belong to the sampled ids"
        assert np.array_equal(ns0.view(np.int64).reshape(-1), a + 500000)
        assert np.array_equal(fields["reward"].cpu().numpy(), (a % 1000).astype(np.float32))
        wc = w.cpu().numpy()
        assert wc.max() == 1.0 and (wc > 0).all() and np.isfinite(wc).all()
        print("WEIGHTS %d %s" % (r, wc.tobytes().hex()), flush=True)  # (the parent compares the two control planes bit for bit)
        rep.update_priority(torch.full((BATCH,), 0.5 + 0.1 * r, device=dev))
    rep.stop()
    print("CONTROL %s" % ("slots" if rep.slots else "collective"), flush=True)
    rep.close()
    print("LEARNER OK", flush=True)
else:
    if rank == 2:  # one partition with its frame-stack fields in 8 MB chunks (9 descriptors each), one with plain IPC handles
        from rela_amd import _capi as capi

        capi.check(capi.lib.rela_runtime_set_replay_chunk_bytes(8 << 20), "chunk bytes")
    part = FFReplay(2048, 11 + rank, 0.6, BETA, 0, A, dev)
    n = 2048
    tags = torch.arange(n, dtype=torch.int64) + rank * 1000000
    s = torch.zeros((n, 4, 84, 84), dtype=torch.uint8)
    ns = torch.zeros((n, 4, 84, 84), dtype=torch.uint8)
    s.view(n, -1)[:, :8] = torch.from_numpy(tags.numpy().view(np.uint8).reshape(n, 8))
    ns.view(n, -1)[:, :8] = torch.from_numpy((tags + 500000).numpy().view(np.uint8).reshape(n, 8))
    s, ns = s.to(dev), ns.to(dev)
    eps = torch.zeros((n, 1), device=dev)
    legal = torch.ones((n, A), device=dev)
    a = tags.to(dev)
    reward = (tags % 1000).float().to(dev)
    term = torch.zeros(n, dtype=torch.uint8, device=dev)
    boot = torch.ones(n, device=dev)
    prio = (torch.rand(n, generator=torch.Generator().manual_seed(rank)) + 0.1).to(dev)
    ptrs = [s.data_ptr(), ns.data_ptr(), eps.data_ptr(), eps.data_ptr(), legal.data_ptr(), legal.data_ptr(), a.data_ptr(),
            reward.data_ptr(), term.data_ptr(), boot.data_ptr()]
    part.add_rows(n, ptrs, prio)
    torch.cuda.synchronize()
    seen = []

    def on_weights(on_flat, tg_flat):
        seen.append((float(on_flat[::4097].mean()), float(tg_flat[::4097].mean())))

    srv = NativePartitionServer(part, specs, BATCH, BETA, dev, on_weights=on_weights)
    srv.serve_forever()
    srv.close()
    assert seen == [(float(r + 1), float(-(r + 1))) for r in range(ROUNDS)], seen  # read straight from the learner's buffers
    assert srv.served == ROUNDS
    print("ACTOR %d OK" % rank, flush=True)
dist.barrier()
dist.destroy_process_group()
