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
extern int g_scan_conv_wg1024;
extern int g_scan_conv_w8;
extern int g_scan_conv_tpb3;
extern int g_scan_conv_bn64_th16;
extern int g_scan_conv_glds;
extern int g_scan_wgrad_v6;
extern int g_scan_wgrad_prio;
extern int g_scan_wgrad_tile;
extern int g_scan_wgrad_wgs;
extern int g_scan_gconv_mfma;
extern int g_scan_dbscan_bf16x3;

static int* tune_slot(const char* key) {
  static const struct {
    const char* key;
    int* slot;
  } knobs[] = {
      {"conv_bn256", &g_scan_conv_bn256},   {"conv_v2", &g_scan_conv_v2},         {"conv_wg1024", &g_scan_conv_wg1024},
      {"conv_w8", &g_scan_conv_w8},         {"conv_tpb3", &g_scan_conv_tpb3},     {"conv_bn64_th16", &g_scan_conv_bn64_th16},
      {"conv_glds", &g_scan_conv_glds},     {"wgrad_v6", &g_scan_wgrad_v6},       {"wgrad_prio", &g_scan_wgrad_prio},   {"wgrad_tile", &g_scan_wgrad_tile},
      {"wgrad_wgs", &g_scan_wgrad_wgs},     {"gconv_mfma", &g_scan_gconv_mfma},   {"dbscan_bf16x3", &g_scan_dbscan_bf16x3},
  };
  for (const auto& k : knobs)
    if (strcmp(key, k.key) == 0) return k.slot;
  return nullptr;
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
