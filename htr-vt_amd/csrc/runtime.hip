// runtime.hip -- error reporting shared by every entry point of libhtrvt_hip.so.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace htrvt {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local char g_kernel[160] = "";

void set_last_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return -3;
  }
  return 0;
}

}  // namespace htrvt

extern "C" int htrvt_version(void) { return 100; }
extern "C" const char* htrvt_last_error(void) { return htrvt::g_err; }
extern "C" const char* htrvt_last_kernel(void) { return htrvt::g_kernel; }
