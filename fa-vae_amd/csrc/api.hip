// Error reporting and version for libfavae_hip.so.
#include <stdarg.h>
#include "common.h"
#include "favae_hip.h"

static thread_local char g_err[512] = "";

extern "C" void favae_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* favae_last_error(void) { return g_err; }
extern "C" int favae_version(void) { return 1; }
