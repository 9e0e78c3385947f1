// Error plumbing and ABI self-description for libusdm_hip.so.
#include "common.h"
#include "../../include/usdm_hip.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void usdm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* usdm_last_error(void) { return g_err; }
extern "C" int usdm_abi_version(void) { return 1; }
extern "C" int usdm_sizeof_gemm_args(void) { return (int)sizeof(usdm_gemm_args); }
