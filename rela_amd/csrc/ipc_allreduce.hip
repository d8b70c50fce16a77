// ipc_allreduce.hip -- gradient all-reduce of the replicated layout over IPC-mapped buffers (include/rela_amd.h:
// rela_ipc_allreduce_*; SURVEY 8e / VERDICT r4 item 8).
//
// The reference has no counterpart: it trains with ONE learner (pyrela/main.py:206-251).  The replicated layout of bench.py
// (one learner replica per GPU, one process per GPU) averages a flat f32 gradient bucket of 1.69 M values every step; this
// is that sum without a collective library: every rank maps every other rank's bucket, and
//   phase 1 (reduce-scatter): rank r sums slice r of all W buckets IN RANK ORDER ((g0 + g1) + g2) + ... into a slice
//            buffer of its own -- peer reads (xGMI between GPUs), one kernel;
//   phase 2 (all-gather):     rank r copies every reduced slice out of its owner's slice buffer into its own bucket.
// Every rank ends with bit-identical sums, equal to a host-side f32 sum in rank order (the test's definition).
//
// Ordering across processes needs no kernel that spins (a kernel waiting for another process's kernel can hang a GPU both
// share) and, in the default mode, no host synchronisation either: every rank owns two monotonic step counters in a page of
// POSIX shared memory that every process registers with HIP (hipHostRegister).  A phase boundary is
//   hipStreamWriteValue32(stream, &mine[rank], step)            -- after everything queued before it, with release semantics
//   hipStreamWaitValue32(stream, &theirs[p], step, >=)  for all p -- a wait of the stream's command processor, not of a CU
// Counters only grow, so a wait binds to a VALUE and there is nothing to re-arm: no host barrier, no ring of signals.
// (A stream that waits for a word blocks the hardware queue it shares with other streams of the process: right here, where
// the learner's stream has to wait for its peers' gradients anyway; the partition exchange, whose waits would sit in a
// prefetching side stream, waits on the host instead -- rela_amd/parallel.py.)
// (Interprocess events, hipIpcGetEventHandle, were the first implementation: on this runtime an event can be recorded 32
// times, the 33rd hipStreamWaitEvent in another process returns "invalid argument" -- profiles/r05_ipc_event_ring_limit.log.)
// device_flags = 0, or a runtime on which the stream value operations fail the self-test of connect(): every phase
// boundary is a stream synchronisation + a host barrier in the same shared page instead.  All ranks agree on the mode.
// Hazards: a rank overwrites its bucket (next backward) only after its phase 2, which waited for every peer's phase 1 --
// the only readers of that bucket; it overwrites its slice buffer in the next phase 1, which waits for every peer's next
// `ready` counter, written after that peer's phase 2 -- the only reader of that slice buffer.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

#include "common.h"
#include "prof.h"

namespace rela_amd {
namespace {

constexpr int kMaxRanks = RELA_IPC_ALLREDUCE_MAX_RANKS;
constexpr int kThreads = 256;

struct ShmBarrier {  // one page; rank 0 creates and zeroes it
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  std::atomic<uint32_t> attached;
  std::atomic<uint32_t> flags_failed;  // ranks on which the stream value operations did not pass the self-test
  uint32_t pad[12];
  uint32_t ready[kMaxRanks][16];    // step counters, one 64-byte line each: rank r's bucket holds its gradients of step n
  uint32_t reduced[kMaxRanks][16];  // rank r's slice buffer holds the sum of step n
  uint32_t probe[kMaxRanks][16];    // connect()'s self-test
};
static_assert(sizeof(ShmBarrier) <= 4096, "one page");

struct Peers {
  const float* buf[kMaxRanks];
};

// out[i - lo] = ((buf0[i] + buf1[i]) + buf2[i]) + ...   for i in [lo, hi); lo is a multiple of 4
__global__ __launch_bounds__(kThreads) void ipc_reduce_slice(Peers p, int world, int64_t lo, int64_t hi,
                                                            float* __restrict__ out) {
  const int64_t n4 = (hi - lo) >> 2;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
    float4 a = reinterpret_cast<const float4*>(p.buf[0] + lo)[i];
    for (int r = 1; r < world; ++r) {
      const float4 b = reinterpret_cast<const float4*>(p.buf[r] + lo)[i];
      a.x = __fadd_rn(a.x, b.x), a.y = __fadd_rn(a.y, b.y), a.z = __fadd_rn(a.z, b.z), a.w = __fadd_rn(a.w, b.w);
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)((hi - lo) & 3)) {  // tail of the last slice
    const int64_t i = lo + (n4 << 2) + threadIdx.x;
    float a = p.buf[0][i];
    for (int r = 1; r < world; ++r) a = __fadd_rn(a, p.buf[r][i]);
    out[i - lo] = a;
  }
}

// bucket[lo_r .. hi_r) = red_r[0 .. hi_r - lo_r) for every rank r (blockIdx.y = r)
__global__ __launch_bounds__(kThreads) void ipc_gather_slices(Peers red, int64_t chunk, int64_t count, float* __restrict__ bucket) {
  const int r = blockIdx.y;
  const int64_t lo = (int64_t)r * chunk, hi = (lo + chunk < count) ? lo + chunk : count;
  if (lo >= hi) return;
  const int64_t n4 = (hi - lo) >> 2;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride)
    reinterpret_cast<float4*>(bucket + lo)[i] = reinterpret_cast<const float4*>(red.buf[r])[i];
  if (blockIdx.x == 0 && threadIdx.x < (int)((hi - lo) & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    bucket[lo + i] = red.buf[r][i];
  }
}

}  // namespace
}  // namespace rela_amd

using namespace rela_amd;

struct rela_ipc_allreduce {
  int rank = 0, world = 1, device = 0, device_flags = 1;  // device_flags: asked for; mode: what connect() settled on
  int mode = 0;                                           // 1 = stream value operations, 0 = host synchronisation
  int64_t count = 0, chunk = 0;
  float* bucket = nullptr;  // the caller's (library-allocated) buffer, reduced in place
  float* red = nullptr;     // [chunk] the slice this rank reduces
  void* peer_base[kMaxRanks] = {};  // mapped allocations (to close)
  void* peer_red_base[kMaxRanks] = {};
  Peers buckets{}, reds{};
  ShmBarrier* shm = nullptr;
  ShmBarrier* shm_dev = nullptr;  // the same page as the device sees it (hipHostRegister)
  bool registered = false;
  char shm_name[64] = "";
  bool connected = false;
  int64_t runs = 0;
  double barrier_timeout_s = 120.0;
};

static int host_barrier(rela_ipc_allreduce* a) {
  ShmBarrier* b = a->shm;
  const uint32_t gen = b->generation.load(std::memory_order_acquire);
  if (b->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)a->world) {
    b->arrived.store(0, std::memory_order_relaxed);
    b->generation.fetch_add(1, std::memory_order_acq_rel);
    return RELA_OK;
  }
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (uint64_t spins = 0; b->generation.load(std::memory_order_acquire) == gen; ++spins) {
    if ((spins & 1023) == 1023) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
      RELA_CHECK(dt < a->barrier_timeout_s, RELA_ESTATE,
                 "rela_ipc_allreduce: rank %d waited %.0f s for the other ranks (a rank that left, or calls out of step)", a->rank, dt);
      if (dt > 0.002) usleep(50);  // a peer that is far behind: stop burning its core
    }
  }
  return RELA_OK;
}

extern "C" int rela_ipc_allreduce_create(rela_ipc_allreduce** out, int rank, int world, float* bucket_dev, int64_t count,
                                         int device, int device_flags, rela_ipc_allreduce_desc* desc_out) {
  RELA_CHECK(out && desc_out && world >= 1 && world <= kMaxRanks && rank >= 0 && rank < world && bucket_dev && count > 0 &&
                 ((uintptr_t)bucket_dev & 15) == 0,
             RELA_EINVAL, "rela_ipc_allreduce_create: bad arguments (at most %d ranks, a 16-byte aligned bucket)", kMaxRanks);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    set_last_error("rela_ipc_allreduce_create: HIP device %d not available (%d visible); there is no CPU path", device, ndev);
    return RELA_ENODEV;
  }
  DeviceGuard g(device);
  auto* a = new rela_ipc_allreduce();
  a->rank = rank, a->world = world, a->device = device, a->device_flags = device_flags ? 1 : 0;
  a->count = count;
  a->chunk = ((count + world - 1) / world + 3) / 4 * 4;
  a->bucket = bucket_dev;
  memset(desc_out, 0, sizeof(*desc_out));
  auto fail = [&](int rc) {
    rela_ipc_allreduce_destroy(a);
    return rc;
  };
  if (hipMalloc(&a->red, sizeof(float) * (size_t)a->chunk) != hipSuccess) {
    set_last_error("rela_ipc_allreduce_create: slice buffer of %lld floats", (long long)a->chunk);
    return fail(RELA_ENOMEM);
  }
  // the bucket may sit inside a larger allocation (the learner's flat buffers): the handle names the allocation
  hipDeviceptr_t base = nullptr;
  size_t span = 0;
  hipError_t e = hipMemGetAddressRange(&base, &span, bucket_dev);
  if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(desc_out->bucket_handle), base);
  if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(desc_out->red_handle), a->red);
  if (e != hipSuccess) {
    set_last_error("rela_ipc_allreduce_create: exporting the bucket: %s (the bucket must be device memory this "
                   "library allocated, e.g. rela_apex_learner_flat's gradients)", hipGetErrorString(e));
    return fail(RELA_ENODEV);
  }
  desc_out->abi = 1, desc_out->rank = rank, desc_out->world = world, desc_out->device = device, desc_out->device_flags = a->device_flags;
  desc_out->count = count;
  desc_out->bucket_offset = (int64_t)((uintptr_t)bucket_dev - (uintptr_t)base);
  if (rank == 0) {  // the host barrier lives in a shared-memory object rank 0 creates; its name travels in the descriptor
    timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(a->shm_name, sizeof(a->shm_name), "/rela-amd-ar-%d-%lld", (int)getpid(), (long long)t.tv_nsec);
    const int fd = shm_open(a->shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, 4096) != 0) {
      set_last_error("rela_ipc_allreduce_create: shm_open(%s) failed", a->shm_name);
      if (fd >= 0) ::close(fd);
      return fail(RELA_ESTATE);
    }
    void* m = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) {
      set_last_error("rela_ipc_allreduce_create: mmap of the barrier page failed");
      return fail(RELA_ESTATE);
    }
    memset(m, 0, 4096);
    a->shm = new (m) ShmBarrier();
    a->shm->attached.store(1);
    memcpy(desc_out->shm_name, a->shm_name, sizeof(desc_out->shm_name));
  }
  *out = a;
  return RELA_OK;
}

extern "C" int rela_ipc_allreduce_connect(rela_ipc_allreduce* a, const rela_ipc_allreduce_desc* descs) {
  RELA_CHECK(a && descs && !a->connected, RELA_EINVAL, "rela_ipc_allreduce_connect: bad arguments (or connected already)");
  DeviceGuard g(a->device);
  for (int r = 0; r < a->world; ++r)
    RELA_CHECK(descs[r].abi == 1 && descs[r].rank == r && descs[r].world == a->world && descs[r].count == a->count &&
                   descs[r].device_flags == a->device_flags,
               RELA_EINVAL, "rela_ipc_allreduce_connect: descriptor %d does not belong to this group (rank %d of %d, %lld floats)", r,
               descs[r].rank, descs[r].world, (long long)descs[r].count);
  if (a->rank != 0) {
    memcpy(a->shm_name, descs[0].shm_name, sizeof(a->shm_name));
    a->shm_name[sizeof(a->shm_name) - 1] = 0;
    const int fd = shm_open(a->shm_name, O_RDWR, 0600);
    RELA_CHECK(fd >= 0, RELA_ESTATE, "rela_ipc_allreduce_connect: shm_open(%s) failed: same host?", a->shm_name);
    void* m = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);
    RELA_CHECK(m != MAP_FAILED, RELA_ESTATE, "rela_ipc_allreduce_connect: mmap of the barrier page failed");
    a->shm = reinterpret_cast<ShmBarrier*>(m);
    a->shm->attached.fetch_add(1);
  }
  for (int r = 0; r < a->world; ++r) {
    if (r == a->rank) {
      a->buckets.buf[r] = a->bucket, a->reds.buf[r] = a->red;
      continue;
    }
    hipIpcMemHandle_t mh;
    memcpy(&mh, descs[r].bucket_handle, sizeof(mh));
    RELA_HIP(hipIpcOpenMemHandle(&a->peer_base[r], mh, hipIpcMemLazyEnablePeerAccess));
    a->buckets.buf[r] = reinterpret_cast<const float*>((const char*)a->peer_base[r] + descs[r].bucket_offset);
    memcpy(&mh, descs[r].red_handle, sizeof(mh));
    RELA_HIP(hipIpcOpenMemHandle(&a->peer_red_base[r], mh, hipIpcMemLazyEnablePeerAccess));
    a->reds.buf[r] = reinterpret_cast<const float*>(a->peer_red_base[r]);
  }
  a->connected = true;
  int rc = host_barrier(a);  // everyone has the page open: its name can go
  if (rc != RELA_OK) return rc;
  if (a->rank == 0) (void)shm_unlink(a->shm_name);
  if (a->device_flags && a->world > 1) {
    // self-test of the stream value operations on the shared page: this rank writes its probe counter through a stream and
    // waits for it, then (after a host barrier) waits for every peer's -- all of which must return at once.  Any failure on
    // any rank and ALL ranks fall back to host synchronisation.
    bool ok = false;
    hipStream_t st = nullptr;
    int can = 0;
    void* dev = nullptr;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, a->device) == hipSuccess && can &&
        hipHostRegister(a->shm, 4096, hipHostRegisterMapped) == hipSuccess) {
      a->registered = true;
      if (hipHostGetDevicePointer(&dev, a->shm, 0) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
        a->shm_dev = reinterpret_cast<ShmBarrier*>(dev);
        ok = hipStreamWriteValue32(st, &a->shm_dev->probe[a->rank][0], 1u, 0) == hipSuccess &&
             hipStreamWaitValue32(st, &a->shm_dev->probe[a->rank][0], 1u, hipStreamWaitValueGte, 0xFFFFFFFFu) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess &&
             reinterpret_cast<volatile uint32_t*>(&a->shm->probe[a->rank][0])[0] == 1u;
      }
    }
    (void)hipGetLastError();
    if (!ok) a->shm->flags_failed.fetch_add(1);
    rc = host_barrier(a);  // every probe counter that will ever be written is written
    if (rc != RELA_OK) return rc;
    if (ok && a->shm->flags_failed.load() == 0) {
      for (int r = 0; r < a->world && ok; ++r)
        ok = hipStreamWaitValue32(st, &a->shm_dev->probe[r][0], 1u, hipStreamWaitValueGte, 0xFFFFFFFFu) == hipSuccess;
      ok = ok && hipStreamSynchronize(st) == hipSuccess;
      (void)hipGetLastError();
      if (!ok) a->shm->flags_failed.fetch_add(1);
    }
    rc = host_barrier(a);
    if (st) (void)hipStreamDestroy(st);
    if (rc != RELA_OK) return rc;
    a->mode = a->shm->flags_failed.load() == 0 ? 1 : 0;
  }
  return RELA_OK;
}

extern "C" int rela_ipc_allreduce_mode(const rela_ipc_allreduce* a) { return a ? a->mode : -1; }

extern "C" int rela_ipc_allreduce_run(rela_ipc_allreduce* a, void* stream_) {
  RELA_CHECK(a && a->connected, RELA_ESTATE, "rela_ipc_allreduce_run: connect first");
  if (a->world == 1) return RELA_OK;
  hipStream_t s = (hipStream_t)stream_;
  DeviceGuard g(a->device);
  a->runs += 1;
  const uint32_t step = (uint32_t)a->runs;
  auto phase_boundary = [&](uint32_t (*counters)[16], const char* phase) -> int {
    if (a->mode == 1) {
      ShmBarrier* d = a->shm_dev;
      uint32_t(*dev_counters)[16] = reinterpret_cast<uint32_t(*)[16]>((char*)d + ((char*)counters - (char*)a->shm));
      hipError_t e = hipStreamWriteValue32(s, &dev_counters[a->rank][0], step, 0);
      for (int r = 0; r < a->world && e == hipSuccess; ++r)
        if (r != a->rank) e = hipStreamWaitValue32(s, &dev_counters[r][0], step, hipStreamWaitValueGte, 0xFFFFFFFFu);
      RELA_CHECK(e == hipSuccess, RELA_ENODEV, "rela_ipc_allreduce_run: call %lld, rank %d, `%s` counters on stream %p: %s",
                 (long long)a->runs, a->rank, phase, (void*)s, hipGetErrorString(e));
      return RELA_OK;
    }
    RELA_HIP(hipStreamSynchronize(s));
    return host_barrier(a);  // every rank's stream is idle
  };
  int rc = phase_boundary(a->shm->ready, "ready");
  if (rc != RELA_OK) return rc;
  const int64_t lo = (int64_t)a->rank * a->chunk, hi = std::min(a->count, lo + a->chunk);
  if (lo < hi) {
    ProfScope prof("ipc_reduce_slice", s);
    const int gx = (int)std::min<int64_t>(std::max<int64_t>(1, (((hi - lo) >> 2) + kThreads - 1) / kThreads), 1024);
    hipLaunchKernelGGL(ipc_reduce_slice, dim3(gx), dim3(kThreads), 0, s, a->buckets, a->world, lo, hi, a->red);
  }
  rc = phase_boundary(a->shm->reduced, "reduced");
  if (rc != RELA_OK) return rc;
  {
    ProfScope prof("ipc_gather_slices", s);
    const int gx = (int)std::min<int64_t>(std::max<int64_t>(1, ((a->chunk >> 2) + kThreads - 1) / kThreads), 256);
    hipLaunchKernelGGL(ipc_gather_slices, dim3(gx, a->world), dim3(kThreads), 0, s, a->reds, a->chunk, a->count, a->bucket);
  }
  RELA_LAUNCH_CHECK();
  return RELA_OK;
}

// exportable device memory for callers that have none of the library's own (tests; a Python learner's gradient bucket):
// torch's caching allocator hands out pieces of larger blocks whose handles other processes cannot open reliably
extern "C" int rela_ipc_alloc_buffer(void** dev_ptr_out, int64_t bytes, int device) {
  RELA_CHECK(dev_ptr_out && bytes > 0, RELA_EINVAL, "rela_ipc_alloc_buffer: bad arguments");
  DeviceGuard g(device);
  RELA_CHECK(g.ok, RELA_ENODEV, "rela_ipc_alloc_buffer: HIP device %d not available; there is no CPU path", device);
  RELA_HIP(hipMalloc(dev_ptr_out, (size_t)bytes));
  return RELA_OK;
}
extern "C" int rela_ipc_free_buffer(void* dev_ptr, int device) {
  if (!dev_ptr) return RELA_OK;
  DeviceGuard g(device);
  RELA_HIP(hipFree(dev_ptr));
  return RELA_OK;
}

extern "C" void rela_ipc_allreduce_destroy(rela_ipc_allreduce* a) {
  if (!a) return;
  DeviceGuard g(a->device);
  (void)hipDeviceSynchronize();  // this rank's reads of its peers' memory are done ...
  if (a->connected && a->world > 1) {  // ... and nobody unmaps while a peer may still be reading (a peer that died: give up)
    a->barrier_timeout_s = 10.0;
    (void)host_barrier(a);
  }
  for (int r = 0; r < a->world; ++r) {
    if (a->peer_base[r]) (void)hipIpcCloseMemHandle(a->peer_base[r]);
    if (a->peer_red_base[r]) (void)hipIpcCloseMemHandle(a->peer_red_base[r]);
  }
  (void)hipFree(a->red);
  if (a->shm) {
    if (a->registered) (void)hipHostUnregister(a->shm);
    if (a->rank == 0 && !a->connected && a->shm_name[0]) (void)shm_unlink(a->shm_name);
    (void)munmap(a->shm, 4096);
  }
  delete a;
}
