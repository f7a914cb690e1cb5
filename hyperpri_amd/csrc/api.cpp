// Error reporting and version of the hyperpri_amd C ABI (see include/hyperpri_hip.h).
#include "common.h"
#include <string.h>

static thread_local char g_err[256] = "";

int hpri_set_error(int code, const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
  return code;
}

extern "C" const char* hpri_last_error(void) { return g_err; }
extern "C" int hpri_version(void) { return 100; }  // 0.1.0

// ---- launch-plan options (process-wide; defaults can also come from the environment) -----------------------------
//   conv_nbx_min          output-channel blocks from which conv kernels use the XCD-aware 1-D grid      (HPRI_NBX_MIN, 9)
//   wgrad_xcd_min_tiles   (C, N) tiles from which the weight-gradient kernels use the XCD-aware grid;
//                         0 = never                                                          (HPRI_WGRAD_XCD_MIN, 128)
//   wgrad_xcd_min_strips  ... and only with at least this many 64-pixel strips                (HPRI_WGRAD_XCD_STRIPS, 2048)
#include <stdlib.h>
static int g_opt[3] = {-1, -1, -1};
static const char* const g_opt_name[3] = {"conv_nbx_min", "wgrad_xcd_min_tiles", "wgrad_xcd_min_strips"};
static const char* const g_opt_env[3] = {"HPRI_NBX_MIN", "HPRI_WGRAD_XCD_MIN", "HPRI_WGRAD_XCD_STRIPS"};
static const int g_opt_default[3] = {9, 128, 2048};

int hpri_option(int idx) {
  if (g_opt[idx] < 0) {
    const char* e = getenv(g_opt_env[idx]);
    g_opt[idx] = e ? atoi(e) : g_opt_default[idx];
    if (g_opt[idx] < 0) g_opt[idx] = g_opt_default[idx];
  }
  return g_opt[idx];
}

extern "C" int hpri_set_option(const char* name, int value) {
  for (int i = 0; i < 3; ++i)
    if (name && strcmp(name, g_opt_name[i]) == 0) {
      if (value < 0) return hpri_set_error(HPRI_ERR_ARG, "set_option: value must be >= 0");
      g_opt[i] = value;
      return HPRI_OK;
    }
  return hpri_set_error(HPRI_ERR_ARG, "set_option: unknown option");
}

extern "C" int hpri_get_option(const char* name) {
  for (int i = 0; i < 3; ++i)
    if (name && strcmp(name, g_opt_name[i]) == 0) return hpri_option(i);
  return hpri_set_error(HPRI_ERR_ARG, "get_option: unknown option");
}
