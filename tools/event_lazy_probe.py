#!/usr/bin/env python3
"""Probe (one GPU): how long do hipEventSynchronize / hipEventQuery take on an OLD, long-completed event of stream A while
stream A has fresh work queued behind a wait for stream B (which is busy for ~20 ms)?  And does hipStreamWriteValue32 accept
a word of hipHostMalloc'ed memory?  HostStage (csrc/common.h) frees its staging segments on the answer.  One JSON line."""
import ctypes as C
import json
import time

import torch

hip = C.CDLL(torch.__file__.rsplit("/", 1)[0] + "/lib/libamdhip64.so")
hip.hipEventSynchronize.argtypes = [C.c_void_p]
hip.hipEventQuery.argtypes = [C.c_void_p]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
hip.hipStreamWriteValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]
hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]

torch.cuda.set_device(0)
A, B = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1 << 20, device="cuda")
res = {}


def ev():
    e = C.c_void_p()
    assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0  # hipEventDisableTiming
    return e


def timed(f):
    t0 = time.perf_counter()
    r = f()
    return round((time.perf_counter() - t0) * 1e6, 1), r


host = C.c_void_p()
assert hip.hipHostMalloc(C.byref(host), 4096, 0) == 0
dev = C.c_void_p()
res["get_dev_ptr_rc"] = hip.hipHostGetDevicePointer(C.byref(dev), host, 0)
flag = C.cast(host, C.POINTER(C.c_uint32))
for mode in ("sync", "query", "ticket"):
    old, cross = ev(), ev()
    with torch.cuda.stream(A):
        x.add_(1)
    hip.hipEventRecord(old, C.c_void_p(A.cuda_stream))
    rc_w = hip.hipStreamWriteValue32(C.c_void_p(A.cuda_stream), dev, 7 + len(mode), 0)
    torch.cuda.synchronize()  # `old` has long fired
    with torch.cuda.stream(B):
        torch.cuda._sleep(40_000_000)  # ~20 ms
    hip.hipEventRecord(cross, C.c_void_p(B.cuda_stream))
    hip.hipStreamWaitEvent(C.c_void_p(A.cuda_stream), cross, 0)
    with torch.cuda.stream(A):
        x.add_(1)  # fresh work on A behind the wait for B
    if mode == "sync":
        res["event_synchronize_us"] = timed(lambda: hip.hipEventSynchronize(old))
    elif mode == "query":
        res["event_query_us"] = timed(lambda: hip.hipEventQuery(old))
    else:
        res["write_value32_rc"] = rc_w
        res["ticket_read_us"] = timed(lambda: int(flag[0]))
    torch.cuda.synchronize()
print(json.dumps(res))
