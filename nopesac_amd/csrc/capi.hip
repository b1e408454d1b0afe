// Library bookkeeping: version + thread-local last-error string.
#include <stdarg.h>

#include "common.h"

namespace nps {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace nps

extern "C" int nopesac_version(void) { return 100; }
extern "C" const char* nopesac_last_error(void) { return nps::g_err; }
