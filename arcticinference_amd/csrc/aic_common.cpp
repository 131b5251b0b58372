// Error channel and device probing shared by every entry point of libarctic_hip.so.
#include "aic_common.h"

#include <cstdarg>
#include <cstdio>

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

}  // namespace aic

extern "C" {
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
