// Error channel + version of libdm4d.so.
#include <stdio.h>
#include <string.h>

#include "dm4d.h"
#include "errors.h"

static thread_local char g_err[512] = "";

int dm4d_set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int dm4d_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return DM4D_OK;
  snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", what, hipGetErrorString(e));
  return DM4D_ERR_LAUNCH;
}

extern "C" int dm4d_version(void) { return 100; }
extern "C" const char* dm4d_last_error(void) { return g_err; }
