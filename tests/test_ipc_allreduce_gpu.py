"""Gradient all-reduce over IPC-mapped buffers (include/rela_amd.h: rela_ipc_allreduce_*; csrc/ipc_allreduce.hip; VERDICT r4
item 8): W processes on cuda:0, each with a bucket of 1,687,207 floats inside a library allocation; after every run all
ranks must hold, bit for bit, the f32 sum in rank order a host computes from the same seeds -- forty rounds back to back on a
side stream with no host synchronisation in between, so the write-after-read ordering between one round's peer reads and
the next round's writes is exercised; "pending": a kernel of 10-30 ms (a different length on every rank and round) runs
ahead of every round, so the peers' streams really wait for the step counters.  On a multi-GPU node the peer reads go over xGMI; addresses, descriptors, events and
the barrier are the same."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world,device_flags,busy", [(2, 1, 0), (3, 1, 0), (3, 0, 0), (2, 1, 20000000), (3, 1, 20000000),
                                                     (3, 0, 20000000)],
                         ids=["w2_flags", "w3_flags", "w3_host_sync", "w2_flags_pending", "w3_flags_pending",
                              "w3_host_sync_busy"])
def test_ipc_allreduce_equals_the_host_sum_in_rank_order(world, device_flags, busy):
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DEVICE_FLAGS=str(device_flags), BUSY_CYCLES=str(busy))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "ipc_allreduce_child.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=200)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0 and ("RANK %d OK" % r) in o, "rank %d: %s\n%s" % (r, o[-1500:], e[-3000:])
    modes = {o.strip().split()[-1] for _, o, _ in outs}
    assert len(modes) == 1, modes  # all ranks settled on the same mode
    print("world %d, device_flags %d -> mode %s" % (world, device_flags, modes.pop()))
