// api.hip -- error reporting and version entry points of the C ABI.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace gsr {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace gsr

extern "C" int gsr_version(void) { return 2; }   // 2: pair words + tight lists (round 3)
extern "C" const char *gsr_last_error(void) { return gsr::g_err; }
extern "C" const char *gsr_arch(void) { return "gfx950"; }
