// ABI version + thread-local error text for libnnue_hip.so.
#include <cstdarg>
#include <cstdio>
#include "common.h"

static thread_local char g_err[512] = "";

void nnue_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int nnue_hip_abi_version(void) { return NNUE_HIP_ABI_VERSION; }
extern "C" const char* nnue_hip_last_error(void) { return g_err; }
