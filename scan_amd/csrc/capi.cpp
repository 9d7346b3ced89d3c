// Error channel + ABI version of libscan_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/scan_hip.h"

static thread_local char g_err[512] = "";

extern "C" void scan_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* scan_last_error(void) { return g_err; }
extern "C" int scan_abi_version(void) { return 1; }
