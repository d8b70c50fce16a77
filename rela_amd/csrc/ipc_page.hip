// ipc_page.hip -- a page of 32-bit words shared by the processes of one host AND by their GPUs' command processors
// (include/rela_amd.h: rela_ipc_page_*): the control plane of the native partition exchange (rela_amd/parallel.py).
//
// The page is POSIX shared memory that every process registers with HIP (hipHostRegister), so a word can be
//   written by a stream   (hipStreamWriteValue32: after everything queued before it, with release semantics),
//   waited for by a stream (hipStreamWaitValue32, >=: a wait of the command processor -- no CU spins, the host goes on),
//   read by a kernel       (the device view of the page), and read, written or waited for by a host thread.
// Words used as step counters only grow, so a wait binds to a value and nothing has to be re-armed or acknowledged.
// csrc/ipc_allreduce.hip orders its two phases the same way (and records why interprocess events were given up).
// The reference has no counterpart: its actors, replay and learner share one address space (pyrela/main.py:131-251).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

#include "common.h"

using namespace rela_amd;

struct rela_ipc_page {
  int device = 0;
  uint32_t* host = nullptr;
  uint32_t* dev = nullptr;
  bool registered = false, creator = false;
  char name[64] = "";
};

static int map_page(rela_ipc_page* p, int fd) {
  void* m = mmap(nullptr, RELA_IPC_PAGE_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  ::close(fd);
  RELA_CHECK(m != MAP_FAILED, RELA_ESTATE, "rela_ipc_page: mmap failed");
  p->host = reinterpret_cast<uint32_t*>(m);
  DeviceGuard g(p->device);
  RELA_CHECK(g.ok, RELA_ENODEV, "rela_ipc_page: HIP device %d not available; there is no CPU path", p->device);
  RELA_HIP(hipHostRegister(p->host, RELA_IPC_PAGE_BYTES, hipHostRegisterMapped));
  p->registered = true;
  void* d = nullptr;
  RELA_HIP(hipHostGetDevicePointer(&d, p->host, 0));
  p->dev = reinterpret_cast<uint32_t*>(d);
  return RELA_OK;
}

extern "C" int rela_ipc_page_create(rela_ipc_page** out, char name_out[64], int device) {
  RELA_CHECK(out && name_out, RELA_EINVAL, "rela_ipc_page_create: bad arguments");
  auto* p = new rela_ipc_page();
  p->device = device, p->creator = true;
  static std::atomic<int> serial{0};
  timespec t;
  clock_gettime(CLOCK_REALTIME, &t);
  snprintf(p->name, sizeof(p->name), "/rela-amd-page-%d-%d-%lld", (int)getpid(), serial.fetch_add(1), (long long)t.tv_nsec);
  const int fd = shm_open(p->name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, RELA_IPC_PAGE_BYTES) != 0) {
    set_last_error("rela_ipc_page_create: shm_open(%s) failed", p->name);
    if (fd >= 0) ::close(fd), (void)shm_unlink(p->name);
    delete p;
    return RELA_ESTATE;
  }
  int rc = map_page(p, fd);
  if (rc != RELA_OK) {
    (void)shm_unlink(p->name);
    rela_ipc_page_close(p);
    return rc;
  }
  memset(p->host, 0, RELA_IPC_PAGE_BYTES);
  memcpy(name_out, p->name, sizeof(p->name));
  *out = p;
  return RELA_OK;
}

extern "C" int rela_ipc_page_open(rela_ipc_page** out, const char* name, int device) {
  RELA_CHECK(out && name && name[0] == '/', RELA_EINVAL, "rela_ipc_page_open: bad arguments");
  auto* p = new rela_ipc_page();
  p->device = device;
  snprintf(p->name, sizeof(p->name), "%s", name);
  const int fd = shm_open(p->name, O_RDWR, 0600);
  if (fd < 0) {
    set_last_error("rela_ipc_page_open: shm_open(%s) failed: another host, or the creator unlinked it already", p->name);
    delete p;
    return RELA_ESTATE;
  }
  int rc = map_page(p, fd);
  if (rc != RELA_OK) {
    rela_ipc_page_close(p);
    return rc;
  }
  *out = p;
  return RELA_OK;
}

// the creator, once every other process has opened the page: the name goes, the mappings stay
extern "C" int rela_ipc_page_unlink(rela_ipc_page* p) {
  RELA_CHECK(p && p->creator, RELA_EINVAL, "rela_ipc_page_unlink: not the creator");
  (void)shm_unlink(p->name);
  return RELA_OK;
}

extern "C" void rela_ipc_page_close(rela_ipc_page* p) {
  if (!p) return;
  if (p->host) {
    if (p->registered) {
      DeviceGuard g(p->device);
      (void)hipDeviceSynchronize();  // no stream operation on the page is still queued
      (void)hipHostUnregister(p->host);
    }
    (void)munmap(p->host, RELA_IPC_PAGE_BYTES);
  }
  delete p;
}

extern "C" void* rela_ipc_page_host_ptr(rela_ipc_page* p) { return p ? p->host : nullptr; }
extern "C" void* rela_ipc_page_dev_ptr(rela_ipc_page* p) { return p ? p->dev : nullptr; }

#define PAGE_WORD_CHECK(fn) \
  RELA_CHECK(p && word >= 0 && word < RELA_IPC_PAGE_BYTES / 4, RELA_EINVAL, fn ": word %d outside the page", word)

extern "C" int rela_ipc_page_write32(rela_ipc_page* p, int word, uint32_t value, void* stream) {
  PAGE_WORD_CHECK("rela_ipc_page_write32");
  DeviceGuard g(p->device);
  RELA_HIP(hipStreamWriteValue32((hipStream_t)stream, p->dev + word, value, 0));
  return RELA_OK;
}

extern "C" int rela_ipc_page_wait32(rela_ipc_page* p, int word, uint32_t value, void* stream) {
  PAGE_WORD_CHECK("rela_ipc_page_wait32");
  DeviceGuard g(p->device);
  RELA_HIP(hipStreamWaitValue32((hipStream_t)stream, p->dev + word, value, hipStreamWaitValueGte, 0xFFFFFFFFu));
  return RELA_OK;
}

extern "C" int rela_ipc_page_host_store32(rela_ipc_page* p, int word, uint32_t value) {
  PAGE_WORD_CHECK("rela_ipc_page_host_store32");
  reinterpret_cast<std::atomic<uint32_t>*>(p->host + word)->store(value, std::memory_order_release);
  return RELA_OK;
}

extern "C" int rela_ipc_page_host_load32(rela_ipc_page* p, int word, uint32_t* value_out) {
  PAGE_WORD_CHECK("rela_ipc_page_host_load32");
  RELA_CHECK(value_out, RELA_EINVAL, "rela_ipc_page_host_load32: bad arguments");
  *value_out = reinterpret_cast<std::atomic<uint32_t>*>(p->host + word)->load(std::memory_order_acquire);
  return RELA_OK;
}

// a host thread waits until the word has reached `value` (written by a stream of any process, or by a host thread)
extern "C" int rela_ipc_page_host_wait32(rela_ipc_page* p, int word, uint32_t value, double timeout_s) {
  PAGE_WORD_CHECK("rela_ipc_page_host_wait32");
  auto* w = reinterpret_cast<std::atomic<uint32_t>*>(p->host + word);
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (uint64_t spins = 0; (int32_t)(w->load(std::memory_order_acquire) - value) < 0; ++spins) {
    if ((spins & 255) == 255) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
      RELA_CHECK(dt < timeout_s, RELA_EWOULDBLOCK, "rela_ipc_page_host_wait32: word %d still %u after %.1f s (waiting for %u)", word,
                 w->load(), dt, value);
      if (dt > 0.0005) usleep(20);  // the writer is a learner step away: leave the core to the actor threads
    }
  }
  return RELA_OK;
}

// Do the stream operations work on this page, in this process, on this runtime?  Writes `value` to `word` through a
// private stream, waits for it through the same stream and checks the host's view.  -> 1 / 0 in *ok_out.
extern "C" int rela_ipc_page_selftest(rela_ipc_page* p, int word, uint32_t value, int* ok_out) {
  PAGE_WORD_CHECK("rela_ipc_page_selftest");
  RELA_CHECK(ok_out, RELA_EINVAL, "rela_ipc_page_selftest: bad arguments");
  DeviceGuard g(p->device);
  *ok_out = 0;
  int can = 0;
  hipStream_t st = nullptr;
  if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, p->device) == hipSuccess && can &&
      hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
    const bool ok = hipStreamWriteValue32(st, p->dev + word, value, 0) == hipSuccess &&
                    hipStreamWaitValue32(st, p->dev + word, value, hipStreamWaitValueGte, 0xFFFFFFFFu) == hipSuccess &&
                    hipStreamSynchronize(st) == hipSuccess &&
                    reinterpret_cast<std::atomic<uint32_t>*>(p->host + word)->load(std::memory_order_acquire) == value;
    *ok_out = ok ? 1 : 0;
  }
  if (st) (void)hipStreamDestroy(st);
  (void)hipGetLastError();
  return RELA_OK;
}
