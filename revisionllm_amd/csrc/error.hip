// Thread-local error string behind rv_last_error().
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void rv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int rv_abi_version(void) { return RV_ABI_VERSION; }

extern "C" int rv_last_error(char* buf, size_t n) {
    if (!buf || n == 0) return RV_ERR_ARG;
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
    return RV_OK;
}
