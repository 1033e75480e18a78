// Error reporting and version for libfavae_hip.so.
#include <stdarg.h>
#include <stdlib.h>
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

// FAVAE_SYNC=1 in the environment makes every launch synchronous (debugging aid; never set in production).
extern "C" int favae_debug_sync_enabled(void) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FAVAE_SYNC"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
}
