// error.cpp — thread-local last-error string behind uia_last_error() (include/uia_hip.h).
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void uia_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* uia_last_error(void) { return g_err; }
