// Thread-local error reporting shared by all entry points of libdm4d.so.
#pragma once
#include <hip/hip_runtime.h>

int dm4d_set_error(int code, const char* msg);
// hipGetLastError() after a launch -> DM4D_ERR_LAUNCH with the HIP error string
int dm4d_check_launch(const char* what);
