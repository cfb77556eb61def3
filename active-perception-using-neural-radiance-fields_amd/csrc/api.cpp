// Error reporting and library-level queries of libmi355nerf.so.
#include "common.h"

namespace mnf {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mnf

extern "C" const char *mnf_last_error(void) { return mnf::g_err; }
extern "C" int mnf_version(void) { return 1; }
extern "C" int mnf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
