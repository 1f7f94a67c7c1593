// fmhr_amd/csrc/abi.cpp — error plumbing of the C ABI (include/fmhr_ham.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
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

bool debug_sync()
{
    static int v = -1;
    if (v < 0) { const char* e = getenv("FMHR_DEBUG_SYNC"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

int hip_fail(hipError_t e, const char* what)
{
    set_error("HIP error in %s: %s", what, hipGetErrorString(e));
    return FMHR_E_HIP;
}
}  // namespace fmhr

namespace fmhr {
static thread_local void** g_ev = nullptr;
static thread_local int g_only = -1;

void prof_mark(int slot, int end, hipStream_t st)
{
    if (g_ev == nullptr || (g_only >= 0 && g_only != slot)) return;
    (void)hipEventRecord((hipEvent_t)g_ev[2 * slot + end], st);
}
}  // namespace fmhr

extern "C" int fmhr_timing_attach(void** events, int n_events, int only_slot)
{
    if (events == nullptr || n_events < 2 * FMHR_SLOT_COUNT || only_slot >= FMHR_SLOT_COUNT) {
        fmhr::set_error("timing_attach: need %d events", 2 * FMHR_SLOT_COUNT);
        return FMHR_E_ARG;
    }
    fmhr::g_ev = events;
    fmhr::g_only = only_slot;
    return FMHR_OK;
}

extern "C" int fmhr_timing_detach(void)
{
    fmhr::g_ev = nullptr;
    fmhr::g_only = -1;
    return FMHR_OK;
}

extern "C" int fmhr_abi_version(void) { return FMHR_ABI_VERSION; }
extern "C" const char* fmhr_last_error(void) { return fmhr::g_err; }
