// Library-wide plumbing: version, error string, device probe.
#include "common.hpp"
#include <string.h>
#include <atomic>

namespace {
thread_local char g_err[512] = "";
}

void bist_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Host-side launch counters per kernel family (BIST_K_*): lets tests and the bench state WHICH implementation ran for a shape
// (e.g. the matrix-core stage-1 kernel rather than the fp32 fallback) without a profiler.
namespace { std::atomic<long long> g_launches[BIST_K_COUNT]; }
void bist_count_launch(int family) { if (family >= 0 && family < BIST_K_COUNT) g_launches[family].fetch_add(1, std::memory_order_relaxed); }
extern "C" int64_t bist_launch_count(int32_t family) { return family >= 0 && family < BIST_K_COUNT ? (int64_t)g_launches[family].load() : -1; }
extern "C" void bist_launch_count_reset(void) { for (auto& c : g_launches) c.store(0); }

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
