#!/usr/bin/env python3
"""Probe (one GPU, two processes): do hipIpcGetMemHandle / hipIpcOpenMemHandle and INTERPROCESS EVENTS
(hipIpcGetEventHandle / hipIpcOpenEventHandle) work on this box?  Decides how rela_amd's native partition exchange
orders its two sides.  Prints one JSON line."""
import ctypes as C
import json
import multiprocessing as mp
import os

HIP = None


def hip():
    global HIP
    if HIP is None:
        HIP = C.CDLL("libamdhip64.so")
    return HIP


class Handle(C.Structure):
    _fields_ = [("b", C.c_ubyte * 64)]


def child(conn):
    h = hip()
    out = {}
    memh, evh, n = conn.recv()
    mh, eh = Handle(), Handle()
    C.memmove(C.byref(mh), memh, 64)
    C.memmove(C.byref(eh), evh, 64)
    ptr = C.c_void_p()
    out["open_mem"] = h.hipIpcOpenMemHandle(C.byref(ptr), mh, 1)  # hipIpcMemLazyEnablePeerAccess
    ev = C.c_void_p()
    out["open_event"] = h.hipIpcOpenEventHandle(C.byref(ev), eh) if evh is not None else -1
    conn.send("opened")
    conn.recv()  # parent recorded the event after its memset
    stream = C.c_void_p()
    h.hipStreamCreate(C.byref(stream))
    if out["open_event"] == 0:
        out["wait_event"] = h.hipStreamWaitEvent(stream, ev, 0)
    buf = (C.c_ubyte * n)()
    out["copy"] = h.hipMemcpyAsync(buf, ptr, n, 2, stream)  # D2H
    out["sync"] = h.hipStreamSynchronize(stream)
    out["data_ok"] = all(b == 0x5A for b in buf)
    out["close"] = h.hipIpcCloseMemHandle(ptr)
    conn.send(out)


def main():
    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    p = ctx.Process(target=child, args=(b,))
    p.start()  # before this process touches the GPU
    h = hip()
    n = 1 << 20
    ptr = C.c_void_p()
    res = {"malloc": h.hipMalloc(C.byref(ptr), n)}
    mh = Handle()
    res["get_mem"] = h.hipIpcGetMemHandle(C.byref(mh), ptr)
    ev = C.c_void_p()
    res["event_create"] = h.hipEventCreateWithFlags(C.byref(ev), 0x2 | 0x4)  # DisableTiming | Interprocess
    eh = Handle()
    res["get_event"] = h.hipIpcGetEventHandle(C.byref(eh), ev) if res["event_create"] == 0 else -1
    a.send((bytes(mh.b), bytes(eh.b) if res["get_event"] == 0 else None, n))
    a.recv()
    res["memset"] = h.hipMemsetAsync(ptr, 0x5A, n, None)
    if res["get_event"] == 0:
        res["record"] = h.hipEventRecord(ev, None)
    else:
        h.hipDeviceSynchronize()
    a.send("go")
    res["child"] = a.recv()
    p.join(30)
    print(json.dumps(res))


if __name__ == "__main__":
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    main()
