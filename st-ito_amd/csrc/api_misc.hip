// api_misc.hip -- error reporting and version for libstito_hip.so
#include "common.h"

namespace stito {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace stito

extern "C" const char *stito_last_error(void) { return stito::g_err; }
extern "C" int stito_version(void) { return 9; }
