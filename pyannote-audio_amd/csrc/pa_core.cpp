// Error reporting + version for libpyannote_amd.so (C ABI declared in include/pyannote_amd.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

namespace pa {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace pa

extern "C" {
int pa_version(void) { return 100; }
const char* pa_last_error(void) { return pa::g_err; }
}
