// Thread-local error string behind rv_last_error().
#include <stdarg.h>

#include "kernels.h"

static thread_local char g_err[512] = "";

void rv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int rv_abi_version(void) { return RV_ABI_VERSION; }
extern "C" int rv_operand_dtype(void) { return RV_OP16; }

extern "C" int rv_last_error(char* buf, size_t n) {
    if (!buf || n == 0) return RV_ERR_ARG;
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
    return RV_OK;
}

// ---- per-call option scope (see kernels.h) ----
const RvOpts g_default_opts{};
static thread_local const RvOpts* t_opts = nullptr;
const RvOpts& rv_cur_opts() { return t_opts ? *t_opts : g_default_opts; }
RvOptScope::RvOptScope(const RvOpts* o) : prev(t_opts) { t_opts = o; }
RvOptScope::~RvOptScope() { t_opts = prev; }
