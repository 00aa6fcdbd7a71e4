// ABI plumbing of libkeep_hip.so: version, thread-local error string, device check.
#include <stdarg.h>

#include "keep_common.h"

static thread_local char g_err[512] = "";

void keep_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------------------------------------ work-queue tickets
// The persistent convolution kernels take their work items from per-XCD ticket counters (dynamic order: a block that runs on a
// slow CU simply takes fewer items; with static striding the slowest block set the kernel time, ~10 % above the mean).  A slot
// is 16 u32 -- 8 tickets + a count of finished blocks -- and is ZERO between launches: the last block to finish clears it.
// Kernels of one stream never overlap, so a slot belongs to a (device, stream) pair; the pool is allocated and zeroed once per
// device (keep_device_ok at load time, or the first launch outside a stream capture) and never freed.  Launches recorded into a
// stream capture keep the strided order.
#include <mutex>
#include <vector>
#define KEEP_SCHED_SLOTS 256
struct SchedPool {
  unsigned* base = nullptr;
  std::vector<hipStream_t> owners;
};
static std::mutex g_sched_mu;
static SchedPool g_sched[16];

static bool sched_pool_init(int dev) {
  SchedPool& sp = g_sched[dev];
  if (sp.base) return true;
  unsigned* ptr = nullptr;
  if (hipMalloc(&ptr, KEEP_SCHED_SLOTS * 16 * sizeof(unsigned)) != hipSuccess) return false;
  if (hipMemset(ptr, 0, KEEP_SCHED_SLOTS * 16 * sizeof(unsigned)) != hipSuccess) {
    (void)hipFree(ptr);
    return false;
  }
  sp.base = ptr;
  return true;
}

// Ticket slot of this stream (nullptr: none available -- the kernel falls back to static striding, same results).
unsigned* keep_sched_slot(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  // Not inside a stream capture: a captured kernel node keeps the slot of the CAPTURE stream, and two graphs captured on one stream
  // but replayed concurrently on two others would share its counters (and the pool must not be allocated inside a capture either).
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
  std::lock_guard<std::mutex> lock(g_sched_mu);
  SchedPool& sp = g_sched[dev];
  if (!sp.base && !sched_pool_init(dev)) return nullptr;
  for (size_t i = 0; i < sp.owners.size(); ++i)
    if (sp.owners[i] == st) return sp.base + i * 16;
  if (sp.owners.size() >= KEEP_SCHED_SLOTS) return nullptr;
  sp.owners.push_back(st);
  return sp.base + (sp.owners.size() - 1) * 16;
}

extern "C" int32_t keep_abi_version(void) { return KEEP_ABI_VERSION; }

extern "C" const char* keep_last_error(void) { return g_err; }

extern "C" int32_t keep_device_ok(int32_t dev) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    keep_set_error("keep_device_ok: hipGetDeviceProperties(%d) failed: %s", dev, hipGetErrorString(e));
    return KEEP_EHIP;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    keep_set_error("keep_device_ok: device %d is %s; this library is built for gfx950 (MI355X) only", dev,
                   prop.gcnArchName);
    return KEEP_EUNSUP;
  }
  int cur = 0;
  if (hipGetDevice(&cur) == hipSuccess && cur == dev && dev < 16) {      // ticket pool of the work queues: allocate outside any capture
    std::lock_guard<std::mutex> lock(g_sched_mu);
    (void)sched_pool_init(dev);
  }
  return KEEP_OK;
}
