// api.hip — error plumbing and version of libscorp_gs (see include/scorp_gs.h).
#include <stdarg.h>

#include "common.hpp"

namespace scorp {
namespace {
thread_local char g_error[512] = "";
}
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
}  // namespace scorp

extern "C" int scorp_version(void) { return 100; /* 0.1.0 */ }
extern "C" const char *scorp_last_error(void) { return scorp::g_error; }
