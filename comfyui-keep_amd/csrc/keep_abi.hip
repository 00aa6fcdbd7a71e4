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
  return KEEP_OK;
}
