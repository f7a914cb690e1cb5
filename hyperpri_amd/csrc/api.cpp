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
