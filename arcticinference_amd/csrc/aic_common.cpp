// Error channel and device probing shared by every entry point of libarctic_hip.so.
#include "aic_common.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <utility>
#include <vector>

namespace aic {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

bool has_device() {
  static int cached = -1;
  if (cached < 0) {
    int n = 0;
    cached = (hipGetDeviceCount(&n) == hipSuccess && n > 0) ? 1 : 0;
    (void)hipGetLastError();
  }
  return cached == 1;
}

static bool g_prof = false;
static int g_stride = 1;      // every g_stride-th launch is bracketed (event pairs cost the stream a few us each)
static unsigned g_calls = 0;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pairs;
static size_t g_used = 0;
static bool g_open = false;

bool profile_enabled() { return g_prof; }
void profile_begin(hipStream_t stream) {
  if (!g_prof) return;
  if (g_calls++ % static_cast<unsigned>(g_stride) != 0) return;
  if (g_used == g_pairs.size()) {
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    g_pairs.emplace_back(a, b);
  }
  g_open = hipEventRecord(g_pairs[g_used].first, stream) == hipSuccess;
}
void profile_end(hipStream_t stream) {
  if (!g_prof || !g_open) return;
  if (hipEventRecord(g_pairs[g_used].second, stream) == hipSuccess) ++g_used;
  g_open = false;
}

}  // namespace aic

extern "C" {
// profiling of the dominant kernel (verify_attn_kernel): enable / reset, then read {sum of device
// time in microseconds, number of launches}.  Reading synchronises the recorded events.
int aic_profile_enable(int on) {
  aic::g_prof = on != 0;
  aic::g_stride = on > 1 ? on : 1;
  aic::g_calls = 0;
  aic::g_open = false;
  aic::g_used = 0;
  return AIC_OK;
}
int aic_profile_read(double* total_us, int* launches) {
  double us = 0.0;
  int n = 0;
  for (size_t i = 0; i < aic::g_used; ++i) {
    float ms = 0.0f;
    if (hipEventSynchronize(aic::g_pairs[i].second) != hipSuccess) continue;
    if (hipEventElapsedTime(&ms, aic::g_pairs[i].first, aic::g_pairs[i].second) == hipSuccess) {
      us += static_cast<double>(ms) * 1000.0;
      ++n;
    }
  }
  if (total_us) *total_us = us;
  if (launches) *launches = n;
  aic::g_used = 0;
  return AIC_OK;
}
// What the instrument itself reads: `pairs` event pairs recorded back to back on `stream` with NOTHING between the two
// events of a pair (the distance of two consecutive marker packets).  bench.py reports it next to the raw average and
// subtracts it, and checks the result against rocprofv3's kernel durations of the same command.
int aic_profile_event_overhead(void* stream, int pairs, double* mean_us, double* min_us) {
  AIC_REQUIRE(pairs > 0 && pairs <= 4096 && mean_us, "pairs in [1, 4096] and an output pointer");
  AIC_NEED_DEVICE();
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev(static_cast<size_t>(pairs), {nullptr, nullptr});
  int rc = AIC_OK;
  for (auto& e : ev)
    if (hipEventCreate(&e.first) != hipSuccess || hipEventCreate(&e.second) != hipSuccess) rc = AIC_ERR_HIP;
  if (rc == AIC_OK) {
    for (auto& e : ev)
      if (hipEventRecord(e.first, s) != hipSuccess || hipEventRecord(e.second, s) != hipSuccess) rc = AIC_ERR_HIP;
  }
  double sum = 0.0, lo = 1e30;
  int n = 0;
  if (rc == AIC_OK) {
    for (auto& e : ev) {
      float ms = 0.0f;
      if (hipEventSynchronize(e.second) != hipSuccess || hipEventElapsedTime(&ms, e.first, e.second) != hipSuccess) {
        rc = AIC_ERR_HIP;
        break;
      }
      sum += ms * 1000.0;
      lo = std::min(lo, static_cast<double>(ms) * 1000.0);
      ++n;
    }
  }
  for (auto& e : ev) {
    if (e.first) (void)hipEventDestroy(e.first);
    if (e.second) (void)hipEventDestroy(e.second);
  }
  if (rc != AIC_OK) {
    aic::set_error("HIP events failed while measuring the event-pair overhead: %s", hipGetErrorString(hipGetLastError()));
    return rc;
  }
  *mean_us = sum / n;
  if (min_us) *min_us = lo;
  return AIC_OK;
}
const char* aic_last_error(void) { return aic::g_err; }
int aic_version(void) { return 100; }
int aic_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
}
