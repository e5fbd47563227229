// Error text and version of libprl_hip.so.
#include <stdarg.h>
#include <string.h>

#include "prl_common.h"

namespace prl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

void clear_error() { g_err[0] = '\0'; }

}  // namespace prl

extern "C" const char* prl_last_error(void) { return prl::g_err; }

extern "C" int prl_abi_version(void) { return 1; }
