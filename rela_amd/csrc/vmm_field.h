// vmm_field.h -- one replay field array as CHUNKS of physical HBM behind ONE contiguous virtual range (HIP virtual memory
// management), so that a partition of any size can be mapped by another process.
//
// Why: the native exchange (include/rela_amd.h, "Native partition exchange") maps the owner's field arrays into the learner
// process.  A hipMalloc'ed array travels as one hipIpcMemHandle_t, and on this platform hipIpcOpenMemHandle of one 37 GB
// allocation (the frame-stack field of a 2^20-row partition) does not return (r4).  Physical memory created with hipMemCreate
// travels chunk by chunk instead, each chunk a POSIX file descriptor (a dmabuf) sent over a Unix socket, and both sides map
// the chunks back to back: a row stays base + slot * row_bytes and no kernel knows about chunks.  profiles/r05_vmm_probe.jsonl:
// 26 chunks of 1 GB, created / mapped / exported / imported / read back in a second process.
//
// The reference has no counterpart: its replay is host memory of one process (rela/prioritized_replay.h:156,339).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include <vector>

namespace rela_amd {

struct VmmRange {
  uint8_t* base = nullptr;
  size_t bytes = 0;  // mapped (a multiple of the granularity; >= the bytes asked for)
  size_t chunk = 0;  // every chunk but the last
  int device = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;
  std::vector<size_t> sizes;

  static hipMemAllocationProp prop_for(int device) {
    hipMemAllocationProp p{};
    p.type = hipMemAllocationTypePinned;
    p.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = device;
    return p;
  }

  hipError_t set_access(int accessing_device) {
    hipMemAccessDesc a{};
    a.location.type = hipMemLocationTypeDevice;
    a.location.id = accessing_device;
    a.flags = hipMemAccessFlagsProtReadWrite;
    return hipMemSetAccess(base, bytes, &a, 1);
  }

  // owner: `want` bytes on `dev` in chunks of (about) `chunk_bytes`
  hipError_t create(size_t want, size_t chunk_bytes, int dev) {
    device = dev;
    hipMemAllocationProp p = prop_for(dev);
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &p, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess) return e;
    if (gran < ((size_t)2 << 20)) gran = (size_t)2 << 20;  // whole 2 MB pages: the TLB reach of a hipMalloc'ed array
    auto up = [&](size_t v) { return (v + gran - 1) / gran * gran; };
    chunk = up(chunk_bytes);
    const size_t total = up(want);
    void* va = nullptr;
    e = hipMemAddressReserve(&va, total, gran, nullptr, 0);
    if (e != hipSuccess) return e;
    base = (uint8_t*)va, bytes = total;
    for (size_t off = 0; off < total; off += chunk) {
      const size_t sz = (total - off < chunk) ? total - off : chunk;
      hipMemGenericAllocationHandle_t h{};
      e = hipMemCreate(&h, sz, &p, 0);
      if (e != hipSuccess) return e;
      handles.push_back(h), sizes.push_back(sz);
      e = hipMemMap(base + off, sz, 0, h, 0);
      if (e != hipSuccess) return e;
    }
    return set_access(dev);
  }

  // owner: one descriptor per chunk, in order (the caller sends them with SCM_RIGHTS and closes its copies)
  hipError_t export_fds(int* fds_out) const {
    for (size_t i = 0; i < handles.size(); ++i) {
      int fd = -1;
      hipError_t e = hipMemExportToShareableHandle(&fd, handles[i], hipMemHandleTypePosixFileDescriptor, 0);
      if (e != hipSuccess) {
        for (size_t k = 0; k < i; ++k) (void)::close(fds_out[k]);
        return e;
      }
      fds_out[i] = fd;
    }
    return hipSuccess;
  }

  // osHandle changed meaning between HIP runtimes, and the wrong form is fatal rather than an error: the 7.0 runtime that
  // PyTorch 2.10+rocm7.0 bundles (hipRuntimeGetVersion 70051831) READS THE DESCRIPTOR THROUGH the pointer and faults on the
  // value form; ROCm 7.2's (70226015) takes the descriptor cast to a pointer, as CUDA does, and returns hipErrorInvalidValue
  // for the pointer form (profiles/r05_vmm_abi.log: all four combinations, two processes).  A process gets whichever runtime
  // was loaded first -- torch's own under Python -- so the form is chosen from the version of the runtime that is running.
  static hipError_t import_fd(hipMemGenericAllocationHandle_t* h, int fd) {
    int version = 0;
    hipError_t e = hipRuntimeGetVersion(&version);
    if (e != hipSuccess) return e;
    if (version >= 70200000) return hipMemImportFromShareableHandle(h, (void*)(intptr_t)fd, hipMemHandleTypePosixFileDescriptor);
    int by_ref = fd;  // (a runtime between the two measured ones gets the form that cannot fault)
    return hipMemImportFromShareableHandle(h, &by_ref, hipMemHandleTypePosixFileDescriptor);
  }

  // importer: the chunks of one field, `chunk_bytes` each but the last, mapped back to back for `accessing_device`
  hipError_t import(const int* fds, int n, size_t chunk_bytes, size_t total, int accessing_device) {
    device = accessing_device;
    chunk = chunk_bytes;
    void* va = nullptr;
    hipError_t e = hipMemAddressReserve(&va, total, (size_t)2 << 20, nullptr, 0);
    if (e != hipSuccess) return e;
    base = (uint8_t*)va, bytes = total;
    const bool trace = getenv("RELA_IPC_TRACE") != nullptr;
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
      const size_t sz = (total - off < chunk) ? total - off : chunk;
      hipMemGenericAllocationHandle_t h{};
      if (trace) fprintf(stderr, "[vmm] chunk %d/%d: descriptor %d, %zu bytes at +%zu: import\n", i, n, fds[i], sz, off);
      e = import_fd(&h, fds[i]);
      if (e != hipSuccess) return e;
      handles.push_back(h), sizes.push_back(sz);
      if (trace) fprintf(stderr, "[vmm] chunk %d/%d: map\n", i, n);
      e = hipMemMap(base + off, sz, 0, h, 0);
      if (e != hipSuccess) return e;
      off += sz;
    }
    if (off != total) return hipErrorInvalidValue;
    if (trace) fprintf(stderr, "[vmm] %zu bytes mapped: set access\n", total);
    e = set_access(accessing_device);
    if (trace) fprintf(stderr, "[vmm] done: %s\n", hipGetErrorString(e));
    return e;
  }

  void destroy() {
    size_t off = 0;
    for (size_t i = 0; i < handles.size(); ++i) {
      (void)hipMemUnmap(base + off, sizes[i]);  // (a chunk whose map failed: the unmap fails too, harmlessly)
      (void)hipMemRelease(handles[i]);
      off += sizes[i];
    }
    if (base) (void)hipMemAddressFree(base, bytes);
    handles.clear(), sizes.clear();
    base = nullptr, bytes = 0;
  }
};

}  // namespace rela_amd
