// Error channel + misc host entry points of libserl_hip (C ABI, see include/serl_hip.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void serl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* serl_last_error(void) { return g_err; }

extern "C" int serl_version(void) { return 1; }

// Device query used by the Python loader to fail loudly on a non-gfx950 device.
extern "C" int serl_device_arch(int device, char* out, int out_len) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    serl_set_error("hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
    return SERL_ERR_HIP;
  }
  snprintf(out, out_len, "%s", prop.gcnArchName);
  return SERL_OK;
}
