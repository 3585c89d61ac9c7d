// api.hip -- error reporting and ABI version of libcdetr_hip.so
#include <stdarg.h>
#include <stdio.h>

#include "../../include/cdetr_hip.h"
#include "common.h"

static thread_local char g_err[512] = "";

void cdetr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cdetr_last_error(void) { return g_err; }
extern "C" int cdetr_abi_version(void) { return CDETR_ABI_VERSION; }
