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
extern int g_scan_conv1x1;
extern int g_scan_wgrad_v6;
extern int g_scan_wgrad_prio;
extern int g_scan_wgrad_tile;
extern int g_scan_wgrad_wgs;
extern int g_scan_gconv_mfma;
extern int g_scan_dbscan_bf16x3;
extern int g_scan_reduce_blocks;

struct Knob {
  const char* key;
  int* slot;
  int dflt;
};
static Knob* knob_table(int* count) {
  static Knob knobs[] = {
      {"conv_bn256", &g_scan_conv_bn256, 0},   {"conv_v2", &g_scan_conv_v2, 0},         {"conv_wg1024", &g_scan_conv_wg1024, 0},
      {"conv_w8", &g_scan_conv_w8, 0},         {"conv_tpb3", &g_scan_conv_tpb3, 0},     {"conv_bn64_th16", &g_scan_conv_bn64_th16, 0},
      {"conv_glds", &g_scan_conv_glds, 0},     {"wgrad_v6", &g_scan_wgrad_v6, 0},       {"wgrad_prio", &g_scan_wgrad_prio, 0},
      {"wgrad_tile", &g_scan_wgrad_tile, 0},   {"wgrad_wgs", &g_scan_wgrad_wgs, 0},     {"gconv_mfma", &g_scan_gconv_mfma, 0},
      {"dbscan_bf16x3", &g_scan_dbscan_bf16x3, 0}, {"reduce_blocks", &g_scan_reduce_blocks, 0}, {"conv1x1", &g_scan_conv1x1, 0},
  };
  static bool init = false;
  if (!init) {  // the values the library was built with: knobs are only ever written through scan_tune below
    for (auto& k : knobs) k.dflt = *k.slot;
    init = true;
  }
  *count = (int)(sizeof(knobs) / sizeof(knobs[0]));
  return knobs;
}
static Knob* find_knob(const char* key) {
  int n = 0;
  Knob* t = knob_table(&n);
  if (key == nullptr) return nullptr;
  for (int i = 0; i < n; ++i)
    if (strcmp(key, t[i].key) == 0) return &t[i];
  return nullptr;
}

// SCAN_TUNE_UNKNOWN (INT_MIN) for an unknown key: every other int, negative ones included, is a legal knob value
extern "C" int scan_tune(const char* key, int value) {
  Knob* k = find_knob(key);
  if (k == nullptr) return SCAN_TUNE_UNKNOWN;
  const int old = *k->slot;
  *k->slot = value;
  return old;
}

// read-only: the current value of a knob (nothing is written)
extern "C" int scan_tune_get(const char* key) {
  const Knob* k = find_knob(key);
  return k == nullptr ? SCAN_TUNE_UNKNOWN : *k->slot;
}

// the value the library was built with
extern "C" int scan_tune_default(const char* key) {
  const Knob* k = find_knob(key);
  return k == nullptr ? SCAN_TUNE_UNKNOWN : k->dflt;
}

// enumeration: the index-th knob's name, NULL past the end
extern "C" const char* scan_tune_key(int index) {
  int n = 0;
  const Knob* t = knob_table(&n);
  return (index >= 0 && index < n) ? t[index].key : nullptr;
}
