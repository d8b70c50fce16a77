"""Native partition exchange over HIP IPC (include/rela_amd.h: rela_replay_export_ipc / _import_ipc /
_remote_gather): a partition OWNED BY ANOTHER PROCESS is mapped here and the rows of its last sample are
gathered by a kernel of THIS process straight out of the owner's memory -- on a multi-GPU node those reads go over
xGMI; with both processes on the one GPU of the test box the address arithmetic, the descriptor and the protocol are
the same.  Two processes on cuda:0."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _payload_of(tags, width):
    col = np.arange(width, dtype=np.int64)[None, :]
    return ((np.asarray(tags, np.int64)[:, None] * 31 + col) % 251).astype(np.uint8)


def _remote_gather_against_owner(chunk_bytes, cap=4096, width=4096, rounds=3, timeout=60, n=None):
    """owner = a child process holding the partition; this process maps it and gathers the rows of the owner's samples"""
    import torch

    from rela_amd import _capi as capi
    from rela_amd.parallel import _import_partition

    B, SEQ = 64, 3
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CHUNK_BYTES=str(chunk_bytes), CAP=str(cap), WIDTH=str(width),
               N=str(n or cap))
    child = subprocess.Popen([sys.executable, os.path.join(HERE, "ipc_owner_child.py")], stdin=subprocess.PIPE,
                             stdout=subprocess.PIPE, text=True, env=env)
    chunks_hit = set()
    try:
        line = child.stdout.readline()
        if chunk_bytes:
            assert line.startswith("DESC2 "), line
            _, sock, hexdesc = line.split()
            raw = bytes.fromhex(hexdesc)
            d = capi.ReplayChunkDesc.from_buffer_copy(raw)
            ring = d.ipc.ring
            page = 2 << 20  # every field of a chunked partition is whole 2 MB pages in chunks of at most chunk_bytes
            want = [-(-(-(-(ring * rb) // page) * page) // chunk_bytes) for rb in (8, width, SEQ * 1024)]
            assert list(d.field_chunks[:3]) == want and d.nfds == sum(want), (list(d.field_chunks[:3]), want, d.nfds)
            import faulthandler

            faulthandler.dump_traceback_later(90, exit=False)  # (an import that does not return: show where)
            rr = _import_partition(capi, C, {"partition": raw, "fd_socket": sock}, 0)
            faulthandler.cancel_dump_traceback_later()
        else:
            assert line.startswith("DESC "), line
            desc = (C.c_ubyte * 4096).from_buffer_copy(bytes.fromhex(line[5:].strip()))
            rr = C.c_void_p()
            capi.check(capi.lib.rela_replay_import_ipc(C.byref(rr), desc, 0), "rela_replay_import_ipc")
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for round_ in range(rounds):
            child.stdin.write("sample\n")
            child.stdin.flush()
            line = child.stdout.readline()
            assert line.startswith("SAMPLED "), line
            exp = json.loads(line[8:])
            if chunk_bytes:
                chunks_hit.update(int(sl) * width // chunk_bytes for sl in exp["slots"])
            tags = torch.zeros(B, dtype=torch.int64, device="cuda")
            pay = torch.zeros((B, width), dtype=torch.uint8, device="cuda")
            seq = torch.zeros((SEQ, B, 1024), dtype=torch.uint8, device="cuda")  # time-major, as rela_replay_sample
            raw = torch.zeros(B, device="cuda")
            sm = torch.zeros(1, device="cuda")
            rows = (C.c_void_p * 3)(tags.data_ptr(), pay.data_ptr(), seq.data_ptr())
            capi.check(capi.lib.rela_replay_remote_gather(rr, B, rows, C.c_void_p(raw.data_ptr()),
                                                          C.c_void_p(sm.data_ptr()), 0, 0, stream), "rela_replay_remote_gather")
            torch.cuda.synchronize()
            got_tags = tags.cpu().numpy()
            assert got_tags.tolist() == exp["tags"], round_
            assert np.array_equal(pay.cpu().numpy(), _payload_of(got_tags, width)), round_
            want_seq = _payload_of(got_tags + 1000003, SEQ * 1024).reshape(B, SEQ, 1024).transpose(1, 0, 2)
            assert np.array_equal(seq.cpu().numpy(), want_seq), round_
            assert np.array_equal(raw.cpu().numpy(), np.float32(exp["raw_w"])), round_
            assert float(sm.item()) == float(np.float32(exp["sum"])), round_
            child.stdin.write("update\n")
            child.stdin.flush()
            assert child.stdout.readline().startswith("UPDATED")
        capi.lib.rela_replay_remote_close(rr)
        child.stdin.write("quit\n")
        child.stdin.flush()
        assert child.wait(timeout=timeout) == 0
    finally:
        if child.poll() is None:
            child.kill()
    return chunks_hit


@pytest.mark.parametrize("chunk_bytes", [0, 2 << 20, 6 << 20], ids=["ipc_handles", "chunks_2MB", "chunks_6MB"])
def test_remote_gather_reads_the_owners_partition(chunk_bytes):
    """chunk_bytes > 0 (r5): the 20 MB payload field, the 15 MB sequence field and the 40 KB tag field are chunks of physical
    memory behind one virtual range each in BOTH processes (10 + 8 + 1 chunks of 2 MB; 4 + 3 + 1 of 6 MB, the last ones
    partial), handed over as file descriptors on a Unix socket"""
    hit = _remote_gather_against_owner(chunk_bytes)
    assert not chunk_bytes or len(hit) >= 3, hit  # the samples came out of several chunks


def test_remote_gather_from_a_partition_hipipc_cannot_carry():
    """a 2^20-transition partition whose frame-stack field is ONE 37 GB array (1,310,720 slots x 28,224 B):
    hipIpcOpenMemHandle of that allocation did not return in r4 and rela_replay_export_ipc refuses it.  In 8 GB chunks it is
    exported, mapped by this process and sampled from all over: 1.2 x 2^20 rows inserted, so that live rows reach into the
    fifth chunk (32 GB up), and every chunk is read by one of the 4 x 64 draws.  (The 4 GB sequence field and the tag field
    are one chunk each: 7 descriptors.)"""
    cap = 1 << 20
    hit = _remote_gather_against_owner(8 << 30, cap=cap, width=28224, rounds=4, timeout=120, n=cap + cap // 5)
    assert hit == set(range(5)), hit


@pytest.mark.parametrize("mode", ["stack", "plane"])
def test_remote_gather_from_a_deduplicated_partition(mode):
    """a de-duplicated partition (frame stacks once in a unit ring, transitions hold references; "plane": one
    new 84x84 plane per env-step) crosses processes with its unit ring, and the learner-side gather rebuilds the stacks as
    rela_replay_sample does in the owner.  The owner is fed by its actor shard from a frame stream this process
    regenerates: every sampled transition must be obs = the env's stack at its tick (tags in the newest plane), next_obs =
    the same env's stack n ticks later, action = what the owner stored -- while the owner's actors keep ticking between its
    sample and this process's read (new units stored ahead; the slots the sample evicted are held)."""
    import torch

    from ipc_dedup_owner_child import B, NSTEP, A, frames
    from rela_amd import _capi as capi
    from rela_amd.parallel import _import_partition, ff_field_specs

    stacks = frames(mode)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DEDUP=mode)
    child = subprocess.Popen([sys.executable, os.path.join(HERE, "ipc_dedup_owner_child.py")], stdin=subprocess.PIPE,
                             stdout=subprocess.PIPE, text=True, env=env)
    try:
        line = child.stdout.readline()
        assert line.startswith("DESC2 "), line
        _, sock, hexdesc = line.split()
        raw = bytes.fromhex(hexdesc)
        d = capi.ReplayChunkDesc.from_buffer_copy(raw)
        ups = {"stack": 1, "plane": 4}[mode]
        assert d.dd_ups == ups and list(d.dd_field) == [0, 1] and d.dd_unit_bytes == 28224 // ups and d.units_chunks >= 1
        assert d.ipc.row_bytes[0] == 4 * ups  # references, not frames
        rr = _import_partition(capi, C, {"partition": raw, "fd_socket": sock}, 0)
        specs = ff_field_specs(A)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        evicting = 0
        for round_ in range(4):
            child.stdin.write("sample\n")
            child.stdin.flush()
            line = child.stdout.readline()
            assert line.startswith("SAMPLED "), line
            exp = json.loads(line[8:])
            assert exp["dev_error"] == 0
            out = {sp.name: sp.empty(2 * B, "cuda") for sp in specs}  # this partition fills the second half of a batch
            for t in out.values():
                t.fill_(0)
            rows = (C.c_void_p * len(specs))(*[out[sp.name].data_ptr() for sp in specs])
            capi.check(capi.lib.rela_replay_remote_gather(rr, B, rows, None, None, 2 * B, B, stream), "rela_replay_remote_gather")
            torch.cuda.synchronize()
            s, ns = out["s"].cpu().numpy(), out["next_s"].cpu().numpy()
            assert not s[:B].any() and not ns[:B].any()  # the other partition's half is untouched
            for b in range(B):
                t0, r0 = int(s[B + b, 3, 0, 0]), int(s[B + b, 3, 0, 1])
                assert t0 + NSTEP < exp["tick"], (round_, b, t0)
                assert np.array_equal(s[B + b], stacks[t0, r0]), (round_, b, t0, r0)
                assert np.array_equal(ns[B + b], stacks[t0 + NSTEP, r0]), (round_, b, t0, r0)
            assert out["a"][B:].cpu().numpy().tolist() == exp["a"], round_
            evicting += exp["added_meanwhile"] < 3 * 12  # blocks were refused: the ring was full of live + held slots
            child.stdin.write("update\n")
            child.stdin.flush()
            assert child.stdout.readline().startswith("UPDATED")
        assert evicting >= 1, "no round ran against a full ring"
        capi.lib.rela_replay_remote_close(rr)
        child.stdin.write("quit\n")
        child.stdin.flush()
        assert child.wait(timeout=60) == 0
    finally:
        if child.poll() is None:
            child.kill()


def _native_exchange(control):
    import socket

    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(3):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RELA_EXCHANGE_CONTROL=control)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "native_exchange_child.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=300)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, "rank %d: %s\n%s" % (r, o[-1500:], e[-3000:])
    assert "LEARNER OK" in outs[0][1] and "ACTOR 1 OK" in outs[1][1] and "ACTOR 2 OK" in outs[2][1]
    assert ("CONTROL %s" % control) in outs[0][1], outs[0][1][-300:]  # (slots: the stream operations passed every self-test)
    return [ln for ln in outs[0][1].splitlines() if ln.startswith("WEIGHTS ")]


def test_native_exchange_three_processes():
    """rela_amd.parallel's native data plane end to end with ONE learner process and TWO actor processes (all on
    cuda:0, gloo for rendezvous and command words): partitions exported (one through IPC handles, one in chunks), every
    sampled row gathered by the learner's own kernel out of the owners' memory into the right slice of the batch (tags
    checked per partition, in sync and asynchronous sampling), importance weights normalised over both partitions,
    priorities back, and the weight publish read by the actors straight from the learner's mapped flat buffers.
    Run with BOTH control planes: r5's slots (step counters in a shared page written and waited for by the streams,
    weights computed by the learner, priorities read by the owners from a mapped buffer: nothing per step on
    torch.distributed) and r4's collectives -- the importance weights of all four rounds must agree bit for bit."""
    slots = _native_exchange("slots")
    collective = _native_exchange("collective")
    assert len(slots) == 4 and slots == collective


