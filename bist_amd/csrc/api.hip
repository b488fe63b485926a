// Library-wide plumbing: version, error string, device probe.
#include "common.hpp"
#include <string.h>
#include <stdlib.h>
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

// ---- development hooks ---------------------------------------------------------------------------------------------------
// In-kernel s_memtime stamp buffers of the fused stage-1 kernel (which = 0) and the persistent decoder kernel (which = 1): set by
// an explicit call that hands over a device buffer the CALLER owns (scripts/stamp_*.py); null in production.  The ablation bits
// (BIST_ST1F_DBG / BIST_DECSTACK_DBG) are read from the environment once per process.
namespace { std::atomic<unsigned long long*> g_stamps[2]; }
unsigned long long* bist_dev_stamps(int which) { return (which == 0 || which == 1) ? g_stamps[which].load(std::memory_order_relaxed) : nullptr; }
int bist_dev_dbg(int which) {
  static const int st1f = [] { const char* e = getenv("BIST_ST1F_DBG"); return e ? atoi(e) : 0; }();
  static const int dec = [] { const char* e = getenv("BIST_DECSTACK_DBG"); return e ? atoi(e) : 0; }();
  return which == 0 ? st1f : which == 1 ? dec : 0;
}
extern "C" int bist_dev_set_stamps(int32_t which, void* device_buffer) {
  BIST_REQUIRE(which == 0 || which == 1, "bist_dev_set_stamps: which = 0 (fused stage 1) or 1 (decoder stack)");
  g_stamps[which].store(reinterpret_cast<unsigned long long*>(device_buffer), std::memory_order_relaxed);
  return BIST_OK;
}

// Development hook: one thread writes the device's constant-rate wall clock (100 MHz) to *slot, as a launch of its own on `stream` --
// captured into a hipGraph it timestamps that point of that stream in every replay (bist_amd/stamps.py, scripts/stamp_step.py).
namespace { __global__ void timestamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); } }
extern "C" int bist_dev_timestamp(void* slot, void* stream) {
  BIST_REQUIRE(slot != nullptr, "bist_dev_timestamp: null slot");
  timestamp_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(reinterpret_cast<unsigned long long*>(slot));
  return BIST_OK;
}

extern "C" int bist_version(void) { return 101; }   // 0.1.1

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
