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

// The per-call A/B knobs are read only under CDETR_TUNING (common.h: cdetr_tune_env): say so once, at load time, when one of them is
// set without it -- otherwise the default kernel runs and nothing tells the user why the knob did nothing.
#include <stdlib.h>
#include <string.h>
extern char** environ;
namespace {
struct TuneEnvCheck {
    TuneEnvCheck() {
        if (getenv("CDETR_TUNING") != nullptr || environ == nullptr) return;
        static const char* const knobs[] = {"CDETR_GEMM_FEWROW_SPLIT", "CDETR_GEMM_SPLITK", "CDETR_GEMM_VARIANT", "CDETR_LSAP_GENERIC",
                                            "CDETR_RCDA_HS", "CDETR_RCDA_NW", "CDETR_RCDA_NW5", "CDETR_WGRAD_VARIANT"};
        for (const char* k : knobs)
            if (getenv(k) != nullptr)
                fprintf(stderr, "libcdetr_hip: %s is set but CDETR_TUNING is not -- per-call tuning knobs are ignored without it\n", k);
    }
} g_tune_env_check;
}  // namespace
