"""Child of tests/test_ipc_allreduce_gpu.py: one rank of W on cuda:0 (gloo carries the descriptors only).  Every round
each rank fills a library-allocated bucket with its own seeded values -- magnitudes spread over 12 binades, so the order of
the additions shows in the last bits -- runs the IPC all-reduce and compares, bit for bit, with the f32 sum in rank order
computed on the host from the same seeds.  Between rounds the bucket is rewritten WITHOUT any host synchronisation in
between: the write-after-read ordering is the library's."""
import ctypes as C
import faulthandler
import os
import sys

faulthandler.dump_traceback_later(150, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from rela_amd import _capi as capi
from rela_amd.engine import dev_view
from rela_amd.parallel import IpcAllReduce

COUNT = int(os.environ.get("COUNT", "1687207"))  # not a multiple of 4 x W: ragged last slice
ROUNDS = int(os.environ.get("ROUNDS", "40"))  # (an interprocess EVENT could be recorded 32 times: the first implementation)
DEVICE_FLAGS = os.environ.get("DEVICE_FLAGS", "1") == "1"
BUSY_CYCLES = int(os.environ.get("BUSY_CYCLES", "0"))  # > 0: a kernel of that many clock cycles runs ahead of every round, so
# that a rank's `ready` counter is still UNWRITTEN when its peers' streams wait for it (the case the hand-over exists for)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])


def values(r, round_):
    g = np.random.default_rng(1000 * round_ + r)
    return (g.standard_normal(COUNT) * np.exp2(g.integers(-6, 6, COUNT))).astype(np.float32)


dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
# a bucket INSIDE a larger library allocation, as the learner's gradients are (the handle names the allocation)
base = C.c_void_p()
capi.check(capi.lib.rela_ipc_alloc_buffer(C.byref(base), 4 * COUNT + 4096, 0), "alloc")
bucket = dev_view(base.value + 1024, (COUNT,), torch.float32, torch.device("cuda:0"))
ar = IpcAllReduce(bucket, device_flags=DEVICE_FLAGS)
assert DEVICE_FLAGS or ar.mode == 0
side = torch.cuda.Stream()
ok = True
staged = [torch.from_numpy(values(rank, k)).pin_memory() for k in range(ROUNDS)]
outs = []
with torch.cuda.stream(side):
    for k in range(ROUNDS):
        if BUSY_CYCLES:
            torch.cuda._sleep(BUSY_CYCLES * (1 + (rank + k) % world))  # ranks finish their "backward" at different times
        bucket.copy_(staged[k], non_blocking=True)  # (the "backward" of step k: overwrites what the peers read in step k - 1)
        ar.run(side)
        outs.append(bucket.clone())
side.synchronize()
for k in range(ROUNDS):
    want = values(0, k)
    for r in range(1, world):
        want = want + values(r, k)  # f32, rank order
    got = outs[k].cpu().numpy()
    if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
        bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
        print("rank %d round %d: %d of %d values differ, first at %d: %r vs %r" % (rank, k, bad.size, COUNT, bad[0], got[bad[0]],
                                                                                  want[bad[0]]), flush=True)
        ok = False
dist.barrier()
ar.close()
capi.lib.rela_ipc_free_buffer(base, 0)
dist.destroy_process_group()
print("RANK %d %s mode %d" % (rank, "OK" if ok else "FAILED", ar.mode), flush=True)
sys.exit(0 if ok else 1)
