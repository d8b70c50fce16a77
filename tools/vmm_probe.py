#!/usr/bin/env python3
"""Probe (one GPU, two processes): does HIP virtual memory management carry a replay field LARGER than hipIpcOpenMemHandle
can import (~24 GB on this pool, r4) between processes?  The owner creates the field as CHUNKS of physical memory
(hipMemCreate, <= CHUNK_GB each, exportable as POSIX file descriptors = dmabufs), maps them into ONE contiguous virtual
range (hipMemAddressReserve / hipMemMap), and hands the descriptors over a socket; the importer maps them into one
contiguous range of its own (hipMemImportFromShareableHandle), so a row is base + slot * row_bytes on both sides and no
kernel has to know about chunks.  Prints one JSON line.   CHUNKS=26 CHUNK_GB=1 python tools/vmm_probe.py"""
import ctypes as C
import json
import multiprocessing as mp
import os
from multiprocessing.reduction import recv_handle, send_handle


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class Flags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleTypes", C.c_int), ("location", Loc), ("win32", C.c_void_p), ("allocFlags", Flags)]


class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


PINNED, POSIX_FD, DEVICE, RW = 1, 1, 1, 3


def hip():
    h = C.CDLL(os.environ.get("HIPLIB", "libamdhip64.so"))  # HIPLIB=<torch>/lib/libamdhip64.so: the runtime a torch process has
    v = C.c_int()
    h.hipRuntimeGetVersion(C.byref(v))
    h.version = v.value
    h.hipMemAddressReserve.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
    h.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
    h.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Access), C.c_size_t]
    h.hipMemCreate.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(Prop), C.c_ulonglong]
    h.hipMemExportToShareableHandle.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_ulonglong]
    h.hipMemImportFromShareableHandle.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int]
    h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return h


def prop():
    p = Prop()
    p.type, p.requestedHandleTypes = PINNED, POSIX_FD
    p.location.type, p.location.id = DEVICE, 0
    return p


def map_range(h, handles, chunk):
    base = C.c_void_p()
    rc = [h.hipMemAddressReserve(C.byref(base), chunk * len(handles), 0, None, 0)]
    for i, hd in enumerate(handles):
        rc.append(h.hipMemMap(C.c_void_p(base.value + i * chunk), chunk, 0, hd, 0))
    a = Access()
    a.location.type, a.location.id, a.flags = DEVICE, 0, RW
    rc.append(h.hipMemSetAccess(base, chunk * len(handles), C.byref(a), 1))
    return base, rc


def child(conn, chunks, chunk):
    h = hip()
    out = {}
    fds = [recv_handle(conn) for _ in range(chunks)]
    handles, rcs = [], []
    for fd in fds:
        hd = C.c_void_p()
        # BYREF=1: osHandle = pointer to the descriptor (HIP 7.0 reads it that way and faults on the value form 7.2 takes)
        os_handle = C.cast(C.pointer(C.c_int(fd)), C.c_void_p) if os.environ.get("BYREF") == "1" else C.c_void_p(fd)
        rcs.append(h.hipMemImportFromShareableHandle(C.byref(hd), os_handle, POSIX_FD))
        handles.append(hd)
    out["import_rc"] = sorted(set(rcs))
    base, rc = map_range(h, handles, chunk)
    out["map_rc"] = sorted(set(rc))
    ok = True
    buf = (C.c_ubyte * 4096)()
    for i in range(chunks):  # the first and the last page of every chunk, through the CONTIGUOUS range
        for off in (0, chunk - 4096):
            r = h.hipMemcpy(buf, C.c_void_p(base.value + i * chunk + off), 4096, 2)
            ok = ok and r == 0 and all(b == (i + 1) & 0xFF for b in buf)
    out["data_ok"] = ok
    conn.send(out)


def main():
    chunks, chunk = int(os.environ.get("CHUNKS", "26")), int(float(os.environ.get("CHUNK_GB", "1")) * (1 << 30))
    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    p = ctx.Process(target=child, args=(b, chunks, chunk))
    p.start()  # before this process touches the GPU
    h = hip()
    res = {"chunks": chunks, "chunk_bytes": chunk, "total_gb": chunks * chunk / 2**30, "hip_runtime": h.version,
           "os_handle": "pointer" if os.environ.get("BYREF") == "1" else "value"}
    gran = C.c_size_t()
    pr = prop()
    res["granularity_rc"] = h.hipMemGetAllocationGranularity(C.byref(gran), C.byref(pr), 0)
    res["granularity"] = gran.value
    handles, rcs = [], []
    for _ in range(chunks):
        hd = C.c_void_p()
        rcs.append(h.hipMemCreate(C.byref(hd), chunk, C.byref(pr), 0))
        handles.append(hd)
    res["create_rc"] = sorted(set(rcs))
    base, rc = map_range(h, handles, chunk)
    res["map_rc"] = sorted(set(rc))
    for i in range(chunks):
        h.hipMemset(C.c_void_p(base.value + i * chunk), (i + 1) & 0xFF, chunk)
    res["sync"] = h.hipDeviceSynchronize()
    rcs = []
    for hd in handles:
        fd = C.c_int(-1)
        rcs.append(h.hipMemExportToShareableHandle(C.byref(fd), hd, POSIX_FD, 0))
        send_handle(a, fd.value, p.pid)
    res["export_rc"] = sorted(set(rcs))
    res["importer"] = a.recv() if a.poll(120) else "timeout"
    p.join(10)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
