"""Child process of tests/test_ipc_gpu.py::test_remote_gather_from_a_deduplicated_partition: OWNS a de-duplicated replay
partition on cuda:0 (frame stacks stored once in a unit ring, the transitions hold references: SURVEY 8f-3) in chunked
mode, fed by an actor shard from a seeded frame stream the parent can regenerate; exports it WITH its unit ring
(rela_replay_export_chunks) and, on request, samples without gathering and keeps its actors ticking -- units stored ahead
and blocks offered to a ring whose evicted slots are held -- before it answers.  Talks over stdin / stdout lines."""
import ctypes as C
import faulthandler
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rela_amd import _capi as capi
from rela_amd.engine import ApexActorEngine, FFNetHandle
from rela_amd.parallel import _export_desc, _FdServer
from rela_amd.replay import FFReplay
from synth import synth_params
from test_dedup_gpu import _sliding_frames

MODE = os.environ.get("DEDUP", "stack")  # stack | plane
R, K, NSTEP, TICKS, CAP, B, A = 12, 4, 2, 100, 256, 32, 6
dev = "cuda:0"


def frames(mode):
    """(ticks, R, 4, 84, 84) u8, every stack tagged (tick, row) in its newest plane; no episode ends"""
    rng = np.random.default_rng(77)
    if mode == "plane":
        stacks, _ = _sliding_frames(rng, R, TICKS, 0.0)
    else:
        stacks = rng.integers(0, 256, (TICKS, R, 4, 84, 84), dtype=np.uint8)
        stacks[:, :, :, 0, 0] = (np.arange(TICKS) % 251).astype(np.uint8)[:, None, None]
        stacks[:, :, :, 0, 1] = np.arange(R, dtype=np.uint8)[None, :, None]
    return stacks


if __name__ == "__main__":
    faulthandler.dump_traceback_later(150, exit=True)  # a blocked owner ends itself: the parent then fails instead of waiting
    capi.check(capi.lib.rela_runtime_set_replay_chunk_bytes(4 << 20), "chunk bytes")  # unit ring: 8 MB (stack) / 2 MB (plane)
    stacks = frames(MODE)
    on, tg = FFNetHandle(A, dev), FFNetHandle(A, dev)
    on.load_state_dict({k: torch.from_numpy(v) for k, v in synth_params(A, 11).items()})
    tg.load_state_dict({k: torch.from_numpy(v) for k, v in synth_params(A, 12).items()})
    replay = FFReplay(CAP, 7, 0.6, 0.4, 0, A, dev, dedup=MODE, guard_units=(NSTEP + 8) * R)
    eng = ApexActorEngine(R, K, A, NSTEP, 0.997, replay, [0.0] * R, dev)
    zeros_r, zeros_t = torch.zeros(R, device=dev), torch.zeros(R, dtype=torch.uint8, device=dev)
    tick = 0

    def run_ticks(n, nonblocking):
        global tick
        for _ in range(n):
            eng.next_obs_slot().copy_(torch.from_numpy(stacks[tick]))
            eng.act(on)
            eng.post_step(zeros_r, zeros_t, on, tg, nonblocking=nonblocking)
            tick += 1
        torch.cuda.synchronize()

    run_ticks(24, False)  # 22-24 x 12 transitions: inside the slot ring (320), so no insert waits
    run_ticks(4, True)    # ... and up to its brim: above the capacity (256), every sample evicts
    raw, fds = _export_desc(replay.h)
    server = _FdServer(fds)
    print("DESC2 %s %s" % (server.name, raw.hex()), flush=True)
    w = torch.empty(B, device=dev)
    for line in sys.stdin:
        cmd = line.strip()
        if cmd == "sample":
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            capi.check(capi.lib.rela_replay_sample(replay.h, B, None, C.c_void_p(w.data_ptr()), stream), "sample")
            torch.cuda.synchronize()
            before = replay.num_add()
            run_ticks(3, True)  # the actors go on: new units, and blocks that only fit where no held slot is
            st = capi.ReplayState()
            ids = np.zeros(B, np.int32)
            capi.check(capi.lib.rela_replay_debug_state(replay.h, C.byref(st), ids.ctypes.data_as(C.c_void_p), None, None),
                       "state")
            acts = np.zeros(B, np.int64)
            for i, s in enumerate(ids):
                one = np.zeros(1, np.int64)
                capi.check(capi.lib.rela_replay_debug_read_rows(replay.h, 6, int(s), 1, one.ctypes.data_as(C.c_void_p)), "read")
                acts[i] = one[0]
            print("SAMPLED " + json.dumps({"slots": [int(x) for x in ids], "a": acts.tolist(), "tick": tick,
                                           "added_meanwhile": replay.num_add() - before, "dev_error": st.dev_error}),
                  flush=True)
        elif cmd == "update":
            replay.update_priority(torch.linspace(0.3, 1.7, B, device=dev))
            for _ in range(8):  # room again: refill until a block is refused (never a blocking insert: nobody would evict)
                before = replay.num_add()
                run_ticks(1, True)
                if replay.num_add() == before:
                    break
            print("UPDATED", flush=True)
        elif cmd == "quit":
            break
    eng.close()
    replay.close()
