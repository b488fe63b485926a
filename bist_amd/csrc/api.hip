// Library-wide plumbing: version, error string, device probe.
#include "common.hpp"
#include <string.h>

namespace {
thread_local char g_err[512] = "";
}

void bist_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int bist_version(void) { return 100; }   // 0.1.0

extern "C" const char* bist_last_error(void) { return g_err; }

extern "C" int bist_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    bist_set_error("no HIP device visible");
    return 0;
  }
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    bist_set_error("cannot query HIP device");
    return 0;
  }
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    bist_set_error("device is %s, this library is built for gfx950 only", p.gcnArchName);
    return 0;
  }
  return 1;
}
