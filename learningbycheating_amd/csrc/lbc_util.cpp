// Error plumbing for the C-ABI: every entry point returns an int status and
// leaves a human-readable message retrievable through lbc_last_error().
#include "lbc_common.hpp"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void lbc_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int lbc_check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        lbc_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return LBC_ELAUNCH;
    }
    return LBC_OK;
}

extern "C" const char* lbc_last_error(void) { return g_err; }
