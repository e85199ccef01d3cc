// Error reporting + version for the C ABI (include/fastvim_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/fastvim_hip.h"

static thread_local char g_err[512] = "";

void fv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* fv_last_error(void) { return g_err; }
extern "C" int fv_version(void) { return FV_ABI_VERSION; }
