// fmhr_amd/csrc/abi.cpp — error plumbing of the C ABI (include/fmhr_ham.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/fmhr_ham.h"

namespace fmhr {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what)
{
    set_error("HIP error in %s: %s", what, hipGetErrorString(e));
    return FMHR_E_HIP;
}
}  // namespace fmhr

extern "C" int fmhr_abi_version(void) { return FMHR_ABI_VERSION; }
extern "C" const char* fmhr_last_error(void) { return fmhr::g_err; }
