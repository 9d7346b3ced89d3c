// Error channel + ABI version of libscan_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

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

// launch-selection knobs (scan_tune): defined next to the launch code that reads them
extern int g_scan_conv_bn256;
extern int g_scan_conv_v2;
extern int g_scan_wgrad_v2;
extern int g_scan_conv_wg1024;
extern int g_scan_conv_w8;
extern int g_scan_dbscan_bf16x3;
extern int g_scan_conv_tpb3;
extern int g_scan_wgrad_wg1024;
extern int g_scan_wgrad_v3;
extern int g_scan_conv_bn64_th16;
extern int g_scan_gconv_mfma;
extern int g_scan_wgrad_wgs;
extern int g_scan_conv_glds;
extern int g_scan_wgrad_exp;
extern int g_scan_conv_exp;
extern int g_scan_wgrad_v4;
extern int g_scan_wgrad_v5;
extern int g_scan_wgrad_il;
extern int g_scan_wgrad_v6;
extern int g_scan_wgrad_prio;

static int* tune_slot(const char* key) {
  int* slot = nullptr;
  if (strcmp(key, "conv_bn256") == 0) slot = &g_scan_conv_bn256;
  if (strcmp(key, "conv_v2") == 0) slot = &g_scan_conv_v2;
  if (strcmp(key, "wgrad_v2") == 0) slot = &g_scan_wgrad_v2;
  if (strcmp(key, "conv_wg1024") == 0) slot = &g_scan_conv_wg1024;
  if (strcmp(key, "conv_w8") == 0) slot = &g_scan_conv_w8;
  if (strcmp(key, "dbscan_bf16x3") == 0) slot = &g_scan_dbscan_bf16x3;
  if (strcmp(key, "conv_tpb3") == 0) slot = &g_scan_conv_tpb3;
  if (strcmp(key, "wgrad_wg1024") == 0) slot = &g_scan_wgrad_wg1024;
  if (strcmp(key, "wgrad_v3") == 0) slot = &g_scan_wgrad_v3;
  if (strcmp(key, "conv_bn64_th16") == 0) slot = &g_scan_conv_bn64_th16;
  if (strcmp(key, "gconv_mfma") == 0) slot = &g_scan_gconv_mfma;
  if (strcmp(key, "wgrad_wgs") == 0) slot = &g_scan_wgrad_wgs;
  if (strcmp(key, "conv_glds") == 0) slot = &g_scan_conv_glds;
  if (strcmp(key, "wgrad_exp") == 0) slot = &g_scan_wgrad_exp;
  if (strcmp(key, "conv_exp") == 0) slot = &g_scan_conv_exp;
  if (strcmp(key, "wgrad_v4") == 0) slot = &g_scan_wgrad_v4;
  if (strcmp(key, "wgrad_v5") == 0) slot = &g_scan_wgrad_v5;
  if (strcmp(key, "wgrad_il") == 0) slot = &g_scan_wgrad_il;
  if (strcmp(key, "wgrad_v6") == 0) slot = &g_scan_wgrad_v6;
  if (strcmp(key, "wgrad_prio") == 0) slot = &g_scan_wgrad_prio;
  return slot;
}

extern "C" int scan_tune(const char* key, int value) {
  if (key == nullptr) return -1;
  int* slot = tune_slot(key);
  if (slot == nullptr) return -1;
  const int old = *slot;
  *slot = value;
  return old;
}

// read-only: the current value of a knob, -1 for an unknown key (nothing is written)
extern "C" int scan_tune_get(const char* key) {
  if (key == nullptr) return -1;
  const int* slot = tune_slot(key);
  return slot == nullptr ? -1 : *slot;
}
