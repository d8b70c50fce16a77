"""The descriptor hand-over of the native exchange (rela_amd/parallel.py: _FdServer / _fetch_fds; include/rela_amd.h:
rela_replay_export_chunks).  The chunks of a large replay field cross processes as file descriptors, which only travel
as SCM_RIGHTS on a Unix socket; here ordinary files stand in for the dmabufs, so the hand-over runs without a GPU."""
import ctypes as C
import multiprocessing as mp
import os

import pytest


def _importer(name, q):
    from rela_amd.parallel import _fetch_fds

    fds = _fetch_fds(name, timeout=30)
    out = []
    for fd in fds:
        os.lseek(fd, 0, os.SEEK_SET)
        out.append(os.read(fd, 64))
        os.close(fd)
    q.put(out)


@pytest.mark.parametrize("n", [1, 9, 128])
def test_descriptors_reach_another_process_in_order(tmp_path, n):
    from rela_amd.parallel import _FdServer

    fds = []
    for i in range(n):
        fd = os.open(str(tmp_path / ("chunk%d" % i)), os.O_RDWR | os.O_CREAT)
        os.write(fd, b"chunk %d of %d" % (i, n))
        fds.append(fd)
    srv = _FdServer(fds, clients=1, timeout=30)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_importer, args=(srv.name, q))
    p.start()
    got = q.get(timeout=60)
    p.join(30)
    srv.join(30)
    assert got == [b"chunk %d of %d" % (i, n) for i in range(n)]
    assert srv.error is None and srv.fds == []
    for fd in fds:  # the owner's copies are closed once the importer has its own
        with pytest.raises(OSError):
            os.fstat(fd)


def test_server_without_a_learner_gives_up_and_closes(tmp_path):
    from rela_amd.parallel import _FdServer

    fd = os.open(str(tmp_path / "c"), os.O_RDWR | os.O_CREAT)
    srv = _FdServer([fd], timeout=0.2)
    srv.join(10)
    assert srv.error is not None and srv.fds == []
    with pytest.raises(OSError):
        os.fstat(fd)


def test_descriptor_structs_match_the_header():
    """_capi's ctypes mirrors against the C compiler's layout of include/rela_amd.h"""
    import subprocess
    import sys

    from rela_amd import _capi as capi

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "rela_amd.h"\nint main(){printf("%zu %zu %zu %zu %zu %d\\n",'
           "sizeof(rela_replay_ipc_desc),sizeof(rela_replay_chunk_desc),offsetof(rela_replay_chunk_desc,nfds),"
           "offsetof(rela_replay_chunk_desc,chunk_bytes),offsetof(rela_replay_chunk_desc,mapped_bytes),RELA_IPC_MAX_FDS);"
           'printf("%zu %zu %zu\\n",offsetof(rela_replay_chunk_desc,dd_ups),offsetof(rela_replay_chunk_desc,dd_unit_bytes),'
           "offsetof(rela_replay_chunk_desc,units_handle));}")
    exe = "/tmp/rela_desc_layout_%d" % os.getpid()
    subprocess.run(["gcc", "-x", "c", "-I", os.path.join(root, "include"), "-o", exe, "-"], input=src.encode(), check=True)
    try:
        out = subprocess.run([exe], capture_output=True, check=True).stdout.split()
    finally:
        os.unlink(exe)
    d = capi.ReplayChunkDesc
    assert [int(x) for x in out] == [C.sizeof(capi.ReplayIpcDesc), C.sizeof(d), d.nfds.offset, d.chunk_bytes.offset,
                                     d.mapped_bytes.offset, capi.IPC_MAX_FDS, d.dd_ups.offset, d.dd_unit_bytes.offset,
                                     d.units_handle.offset]
