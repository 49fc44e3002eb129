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

// ---- numeric status (common.h): every translation unit registers itself at load time; rv_numeric_status_bind points all of them at one
// caller-owned device buffer.  The list is built by static constructors (no HIP call), walked under the caller's current device.
static RvTuNode* g_numeric_head = nullptr;
static unsigned int* g_numeric_ptr[64] = {};      // per device ordinal (one process per GPU is the design; several devices in one process each get their own buffer)
void rv_numeric_register(RvTuNode* n) {
    n->next = g_numeric_head;
    g_numeric_head = n;
}
static int numeric_device() {
    int d = 0;
    return hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64 ? d : -1;
}
unsigned int* rv_numeric_bound() {
    const int d = numeric_device();
    return d < 0 ? nullptr : g_numeric_ptr[d];
}

extern "C" int rv_numeric_status_bind(uint32_t* status_dev) {
    RV_CHECK_ARG(((uintptr_t)status_dev & 15) == 0, "rv_numeric_status_bind: the buffer must be 16-byte aligned");
    for (RvTuNode* n = g_numeric_head; n; n = n->next)
        if (n->bind((unsigned int*)status_dev) != 0) {
            rv_set_error("rv_numeric_status_bind: hipMemcpyToSymbol failed: %s", hipGetErrorString(hipGetLastError()));
            return RV_ERR_HIP;
        }
    const int d = numeric_device();
    RV_CHECK_ARG(d >= 0, "rv_numeric_status_bind: no current device");
    g_numeric_ptr[d] = (unsigned int*)status_dev;
    return RV_OK;
}

// ---- per-call option scope (see kernels.h) ----
const RvOpts g_default_opts{};
static thread_local const RvOpts* t_opts = nullptr;
const RvOpts& rv_cur_opts() { return t_opts ? *t_opts : g_default_opts; }
RvOptScope::RvOptScope(const RvOpts* o) : prev(t_opts) { t_opts = o; }
RvOptScope::~RvOptScope() { t_opts = prev; }
